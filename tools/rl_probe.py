"""Diagnostic: online saturation + relaunch, float64 / float32, persistent kernel / chain, against the oracle loop."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-msgwam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import msgwam_oracle as orc
from oracle.c_oracle import COracle
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
from helpers import STATE_KEYS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
kappa = float(sys.argv[3]) if len(sys.argv) > 3 else 0.008
frac = 0.5
grid = np.linspace(0, 100e3, 101)
s0 = orc.Setup(grid)
sp = gaussian_spectrum(n, s0.grids, s0.rhobar, alpha=alpha)
s = orc.Setup(grid, kappa=kappa, saturate_online=True, dkk=sp["dkk"], dll=sp["dll"], rr_mm_area=sp["area"])
uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 40e3, 10e3); vv = np.zeros_like(uu)
s.set_pressure_gradient(uu, vv)
st = [sp[k] for k in STATE_KEYS[:9]] + [uu, vv]
src = (st[0].copy(), st[3].copy(), st[7].copy())
co = COracle(s)
for nsteps in (1, 2):
    want = st
    masks = []
    for _ in range(nsteps):
        want = co.step(120.0, 1, want)
        want, mask = orc.relaunch(s, want, src, frac)
        masks.append(mask)
    for dtype in ("f64", "f32"):
        for env in ({}, {"MSGW_REGTILES": "0"}, {"MSGW_PERSIST": "0"}):
            for k in ("MSGW_REGTILES", "MSGW_PERSIST"):
                os.environ.pop(k, None)
            os.environ.update(env)
            p = _capi.Propagator(len(grid), n, dtype=dtype)
            p.set_config(s.bvf, s.phi0, s.kappa, True)
            p.set_column(s.grid, s.grids, s.rhobar, s.pressure_gradient, uu, vv)
            p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
            p.set_relaunch(frac)
            p.step(120.0, nsteps, _capi.RELAUNCH)
            dens, rr, mm = p.download_rays()
            gu, gv = p.download_column()
            c = p.counters()
            p.close()
            e = {k: np.abs(a - b) / np.abs(b) for k, a, b in (("dens", dens, want[0]), ("rr", rr, want[3]), ("mm", mm, want[7]))}
            bad = e["mm"] > 1e-4
            print(f"steps {nsteps} {dtype} {env} persist={c['persist_steps']} res={c['persist_resident_tiles']}: "
                  + " ".join(f"{k} max {v.max():.2e} bad {np.mean(v > 1e-4):.4f}" for k, v in e.items())
                  + f" uu {np.abs(gu - want[9]).max() / 4:.2e}; recycled(last) {int(masks[-1].sum())}; bad&recycled {int((bad & masks[-1]).sum())} bad&~recycled {int((bad & ~masks[-1]).sum())}")
            if bad.any() and dtype == "f32" and not env:
                i = np.flatnonzero(bad)[:5]
                print("   idx", i, "got mm", mm[i], "want", want[7][i], "src", src[2][i], "dens got/want/src", dens[i], want[0][i], src[0][i])
