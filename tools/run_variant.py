"""GPU box: run ONE kernel path for a number of steps (target of rocprofv3; bench.py covers the persistent and the
fused fixed-background kernels).  usage: run_variant.py hprop|nz|nz_sat|tall301|tall201 [rays] [steps]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import time
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
kind = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ngrid = int(kind[4:]) if kind.startswith("tall") else 101
lprop, grid, grids, uu, vv = bench.column(ngrid)
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
rng = np.random.default_rng(0)
p = _capi.Propagator(ngrid, n)
phi = sp["phi"]
if kind == "hprop":
    phi = rng.uniform(-0.5, 0.5, n)
    p.set_config(0.01, 0.4, 1.0, False, hprop=True)
elif kind.startswith("tall"):
    p.set_config(0.01, 0.0, 1.0, False)
else:
    p.set_config(0.01, 0.0, 1.0, kind == "nz_sat")
    p.set_bvf_column(0.01 * (1 + 0.2 * grids / grids[-1]))
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], phi, sp["dkk"], sp["dll"], sp["area"])
if kind == "hprop":
    p.upload_hprop(np.zeros(n), phi)
p.step(120.0, 10); p.sync()
t0 = time.perf_counter(); p.step(120.0, steps); p.sync(); dt = time.perf_counter() - t0
words = {"hprop": 71, "nz": 63, "nz_sat": 75}.get(kind, 35)
print(f"{kind}: {dt / steps * 1e6:.1f} us/step  {n * steps / dt:.3e} ray-steps/s  {n * words * 8 / (dt / steps) / 1e9:.0f} GB/s of {words} words per ray-step")
p.close()
