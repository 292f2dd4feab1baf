import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
from oracle import msgwam_oracle as orc
from oracle.c_oracle import COracle
from gpu_helpers import make_prop, gpu_state
for ngrid, gmax in ((301, 150e3), (101, 150e3)):
    rng = np.random.default_rng(77)
    n = 30_011
    grid = np.linspace(0, gmax, ngrid)
    area = rng.uniform(1e-3, 1e-1, n)
    s = orc.Setup(grid, phi0=0.2, kappa=0.9, saturate_online=False, dkk=np.full(n, 1e-4), dll=np.full(n, 1e-4), rr_mm_area=area)
    uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 40e3, 10e3); vv = 0.1 * uu[::-1].copy()
    s.set_pressure_gradient(uu, vv)
    rr = np.sort(rng.uniform(-1e3, 155e3, n)); drr = rng.uniform(50, 1500, n)
    st = [rng.uniform(0, 1e9, n), np.zeros(n), np.full(n, 0.2), rr, drr, rng.normal(0, 1e-4, n), rng.normal(0, 1e-4, n), rng.normal(0, 2e-3, n), area / drr, uu, vv]
    for nsteps in (1, 2, 3):
        want = COracle(s).step(60.0, nsteps, st)
        p = make_prop(s, st); p.step(60.0, nsteps); got = gpu_state(p, st); p.close()
        e = np.abs(got[7] - want[7]); r = e / np.abs(want[7]); i = np.argmax(r)
        print(f"ngrid {ngrid} steps {nsteps}: mm max rel {r.max():.2e} at i={i} mm={want[7][i]:.3e} abs err {e[i]:.2e}; max abs err {e.max():.2e}; uu err {np.max(np.abs(got[9]-want[9])):.2e} (max|uu| {np.max(np.abs(want[9])):.2f})")
