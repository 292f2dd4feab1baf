"""Diagnostic: error statistics of the float32 ray state (MSGW_DTYPE_F32) against the float64 oracle / goldens,
used to choose the tolerances written in tests/test_gpu_f32.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-msgwam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import load, setup_from, state_from   # noqa: E402
from msgwam_amd import _capi                       # noqa: E402


def make(setup, state, dtype):
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = state
    p = _capi.Propagator(len(setup.grid), len(dens), dtype=dtype)
    p.set_config(setup.bvf, setup.phi0, setup.kappa, setup.saturate_online)
    p.set_column(setup.grid, setup.grids, setup.rhobar, setup.pressure_gradient, uu, vv)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, setup.dkk, setup.dll, setup.rr_mm_area)
    return p


def stats(tag, got, want):
    for k, i in (("dens", 0), ("rr", 3), ("mm", 7)):
        e = np.abs(got[i] - want[i]) / np.maximum(np.abs(want[i]), 1e-300)
        print(f"  {tag} {k}: max {e.max():.3e}  p99.9 {np.quantile(e, .999):.3e}  median {np.median(e):.3e}  >1e-4: {(e > 1e-4).sum()}/{e.size}")
    scale = max(np.max(np.abs(want[9])), np.max(np.abs(want[10])), 1e-300)
    print(f"  {tag} uu: {np.max(np.abs(got[9] - want[9])) / scale:.3e}  vv: {np.max(np.abs(got[10] - want[10])) / scale:.3e}")


for name, marks, flags in (("g5_spectrum_coupled", (1, 3), 0), ("g4_saturation_online", (1, 5), 0),
                           ("g3_rk3_coupled_driver", (1, 10, 100), 0),
                           ("g3_rk3_fixedbg_config1", (1, 10, 100, 1000), _capi.FIXED_BACKGROUND)):
    d = load(name)
    st = state_from(d, "in")
    for dtype in ("f32",):
        p = make(setup_from(d), st, dtype)
        done = 0
        print(name, dtype)
        for n in marks:
            p.step(float(d["dt"]), n - done, flags)
            done = n
            dens, rr, mm = p.download_rays()
            uu, vv = p.download_column()
            got = list(st)
            got[0], got[3], got[7], got[9], got[10] = dens, rr, mm, uu, vv
            stats(f"s{n}", got, state_from(d, f"s{n}"))
        print("  counters", {k: v for k, v in p.counters().items() if k in ("persist_steps", "persist_resident_tiles", "elem_bytes", "blocks")})
        p.close()
