"""GPU box: step time of the kernel variants (online saturation, direct saturation, relaunch, per-ray latitude) for
both ray types through the persistent kernel (register-resident tiles / all rays streamed) and through the per-stage
launch chain, bench workload.  usage: variant_bench.py [rays] [f64|f32|both] [alpha]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dtypes = ("f64", "f32") if len(sys.argv) < 3 or sys.argv[2] == "both" else (sys.argv[2],)
alpha = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
lprop, grid, grids, uu, vv = bench.column(101)
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=alpha)
rng = np.random.default_rng(0)
RL = _capi.RELAUNCH
for dtype in dtypes:
    for name, sat, flags, vec in (("plain", False, 0, False), ("online sat", True, 0, False), ("direct sat", False, _capi.DIRECT_SAT_QUIRK, False),
                                  ("relaunch", False, RL, False), ("sat+relaunch", True, RL, False),
                                  ("latitude", False, 0, True), ("sat+latitude", True, 0, True)):
        res = []
        for env in ({"MSGW_REGTILES": "2"}, {"MSGW_REGTILES": "0"}, {"MSGW_PERSIST": "0"}, {"MSGW_REGTILES": "4"}, {"MSGW_REGTILES": "3"}):
            for k in ("MSGW_REGTILES", "MSGW_PERSIST"):
                os.environ.pop(k, None)
            os.environ.update(env)
            p = _capi.Propagator(101, n, dtype=dtype)
            p.set_config(0.01, 0.0, 1.0, sat)
            p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
            phi = rng.uniform(-0.5, 0.5, n) if vec else sp["phi"]
            p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], phi, sp["dkk"], sp["dll"], sp["area"])
            p.set_tuning(4, 4)
            p.step(120.0, 20, flags); p.sync()
            t0 = time.perf_counter(); p.step(120.0, 100, flags); p.sync(); dt = time.perf_counter() - t0
            c = p.counters()
            res.append((dt / 100 * 1e6, c["persist_steps"], c["persist_resident_tiles"]))
            p.close()
        print(f"{dtype} {name:14s}: resident {res[0][0]:6.1f} (res {res[0][2]})   streamed {res[1][0]:6.1f}   chain {res[2][0]:6.1f}   four resident {res[3][0]:6.1f} (res {res[3][2]})   three {res[4][0]:6.1f} (res {res[4][2]}) us/step", flush=True)
if "f64" in dtypes:
    # HPROP_GLOBAL = True: its own per-stage kernel (7 evolving slots per ray)
    for k in ("MSGW_REGTILES", "MSGW_PERSIST"):
        os.environ.pop(k, None)
    p = _capi.Propagator(101, n)
    p.set_config(0.01, 0.4, 1.0, False, hprop=True)
    p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
    phi = rng.uniform(-0.5, 0.5, n)
    p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], phi, sp["dkk"], sp["dll"], sp["area"])
    p.upload_hprop(np.zeros(n), phi)
    p.step(120.0, 10); p.sync()
    t0 = time.perf_counter(); p.step(120.0, 50); p.sync(); dt = time.perf_counter() - t0
    # per stage: read dens lam phi rr drr kk ll mm vol (+6 q, stages 1-2), write 6 (+6 q, stages 0-1): 27+12+18+12 words per step
    print(f"{'HPROP on':16s}: {dt / 50 * 1e6:7.1f} us/step  ({n * 50 / dt:.3e} ray-steps/s, {n * 69 * 8 / (dt / 50) / 1e9:.0f} GB/s of 69 words per ray-step)")
    p.close()
