"""GPU box: step time of the coupled path on tall columns (ngrid > 130), persistent kernel (register-resident tiles /
all rays streamed) vs the per-stage launch chain.  usage: tall_probe.py [rays] [ngrid ...]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for ngrid in [int(a) for a in sys.argv[2:]] or [101, 301, 601]:
    lprop, grid, grids, uu, vv = bench.column(ngrid)
    sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
    res = []
    for env in ({}, {"MSGW_REGTILES": "0"}, {"MSGW_PERSIST": "0"}):
        for k in ("MSGW_REGTILES", "MSGW_PERSIST"):
            os.environ.pop(k, None)
        os.environ.update(env)
        p = _capi.Propagator(ngrid, n)
        p.set_config(0.01, 0.0, 1.0, False)
        p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
        p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
        p.step(120.0, 20); p.sync()
        t0 = time.perf_counter(); p.step(120.0, 100); p.sync(); dt = time.perf_counter() - t0
        c = p.counters()
        res.append((dt / 100 * 1e6, c["persist_steps"], c["persist_resident_tiles"]))
        p.close()
    frac = [n * 280.0 / (r[0] * 1e-6) / 8e12 for r in res]
    print(f"ngrid {ngrid:4d}: resident {res[0][0]:6.1f} us/step (persist {res[0][1]}, res {res[0][2]}, {frac[0]:.2f} of 8 TB/s)   "
          f"streamed {res[1][0]:6.1f} ({frac[1]:.2f})   chain {res[2][0]:6.1f} ({frac[2]:.2f})", flush=True)
