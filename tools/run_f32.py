"""GPU box: float32 rays through the persistent kernel for 100 steps (target of rocprofv3).
usage: run_f32.py alpha [sat] [relaunch] [rays]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
alpha = float(sys.argv[1]); sat = len(sys.argv) > 2 and sys.argv[2] == "1"; rl = len(sys.argv) > 3 and sys.argv[3] == "1"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1_250_000
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
lprop, grid, grids, uu, vv = bench.column(101)
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=alpha)
p = _capi.Propagator(101, n, dtype=dtype)
p.set_config(0.01, 0.0, 1.0, sat)
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
fl = _capi.RELAUNCH if rl else 0
p.step(120.0, 20, fl); p.sync()
t0 = time.perf_counter(); p.step(120.0, 100, fl); p.sync(); dt = time.perf_counter() - t0
u, v = p.download_column()
print(f"alpha {alpha} sat {sat} rl {rl} n {n} {dtype}: {dt / 100 * 1e6:.1f} us/step  max|u| {np.abs(u).max():.3g}  finite {np.isfinite(u).all()}")
p.close()
