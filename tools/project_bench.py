"""GPU box: timing of the diagnostic projections (wave_projection var 0/1/2) on the
resident rays of the bench workload, and of msgw_saturation on caller arrays."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lprop, grid, grids, uu, vv = bench.column(101)
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
p = _capi.Propagator(101, n)
p.set_config(0.01, 0.0, 1.0, False)
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
p.step(120.0, 50); p.sync()
for var, G, name in ((0, grids, "var 0 pseudo-momentum flux on grids"), (1, grids, "var 1 wave-action flux on grids"), (2, grid, "var 2 wave action on grid")):
    p.project(var, G)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        out = p.project(var, G)
    dt = (time.perf_counter() - t0) / reps
    # resident projection reads rr, drr, mm, dmm(or vol), kk, ll, dens, (pvf): 8 words per ray
    print(f"{name:40s}: {dt * 1e6:8.1f} us per call incl. download of the profile  ({n * 64 / dt / 1e9:7.1f} GB/s of 8 words/ray)  sum {np.sum(out):.6e}")
