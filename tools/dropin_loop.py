"""GPU box: cost of the drop-in loop `state = lprop.RK3(dt, state)` (lazy device-backed results fed back) per step."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
import msgwam_amd.libprop as lprop
from msgwam_amd.spectrum import gaussian_spectrum
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
_, grid, grids, uu, vv = bench.column(101)
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
lprop.set_statics(dkk=sp["dkk"], dll=sp["dll"], rr_mm_area=sp["area"])
state = np.empty(11, dtype=object)
for i, k in enumerate(["dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm"]):
    state[i] = sp[k]
state[9], state[10] = uu, vv
for mode, freeze in (("safe", False), ("safe", True), ("fast", False)):
    lprop.set_residency(mode)
    for k in ("dkk", "dll", "rr_mm_area"):                  # read-only statics are recognised by identity
        lprop.statics[k].setflags(write=not freeze)
    s = state
    for _ in range(5):
        s = lprop.RK3(120.0, s)
    np.asarray(s[3])
    t0 = time.perf_counter()
    for _ in range(100):
        s = lprop.RK3(120.0, s)
    rr = np.asarray(s[3])                       # one download at the end
    dt = time.perf_counter() - t0
    print(f"residency {mode}{' + read-only statics' if freeze else ''}: {dt / 100 * 1e6:.0f} us per RK3 call at {n} rays (state fed back, one download at the end)")
lprop.release_device()
