"""GPU box: where does the time of ONE msgw_step call go beyond its kernel?  (bench.py's driver arguments time 20-step
launches: the per-call overhead is part of the headline number.)  usage: launch_overhead.py [rays] [steps]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lprop, grid, grids, uu, vv = bench.column(101)
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
p = _capi.Propagator(101, n)
p.set_config(0.01, 0.0, 1.0, False)
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
p.step(120.0, 5); p.sync()
for flags, name in ((0, "plain"), (_capi.TIME_KERNELS, "TIME_KERNELS")):
    walls, calls, kern = [], [], []
    for _ in range(60):
        c0 = p.counters()
        t0 = time.perf_counter()
        p.step(120.0, steps, flags)
        t1 = time.perf_counter()
        p.sync()
        t2 = time.perf_counter()
        c1 = p.counters()
        walls.append(t2 - t0); calls.append(t1 - t0)
        if flags:
            kern.append(c1["ray_kernel_ms_sum"] - c0["ray_kernel_ms_sum"])
    w, c = np.median(walls) * 1e6, np.median(calls) * 1e6
    k = np.median(kern) * 1e3 if kern else float("nan")
    print(f"{name:13s}: step+sync {w:7.1f} us per call of {steps} steps ({w / steps:.2f} us/step); the msgw_step call itself {c:6.1f} us; "
          f"kernel by events {k:7.1f} us ({k / steps:.2f} us/step); outside the kernel {w - k:6.1f} us")
# back-to-back calls without a sync in between (what a driver loop that feeds the state back does)
p.sync()
t0 = time.perf_counter()
for _ in range(50):
    p.step(120.0, steps)
p.sync()
w = (time.perf_counter() - t0) / 50 * 1e6
print(f"50 calls back to back, one sync: {w:7.1f} us per call ({w / steps:.2f} us/step)")
p.close()
