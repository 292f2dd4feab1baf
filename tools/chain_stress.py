"""GPU box: randomized cross-check of the GENERAL per-stage kernel (csrc/chain_kernels.h) against the numpy oracle over
random combinations of HPROP, N(z) column, online / direct saturation, relaunch, ray type, ray count, column height and
call pattern.  usage: python tools/chain_stress.py [seconds] [seed]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
from msgwam_amd import _capi
from oracle import msgwam_oracle as orc
from test_gpu_parity import _random_case, _tall_case
from test_gpu_chain import make_chain_prop, chain_state, oracle_loop
from helpers import STATE_KEYS

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases, worst64, worst32 = 0, 0.0, 0.0
while time.time() < t_end:
    hprop, nz = [(True, False), (False, True), (True, True)][int(rng.integers(3))]
    f32 = bool(rng.random() < 0.35)
    sat = bool(rng.random() < 0.3)
    direct = 0 if sat else int(rng.choice([0, 0, 1, 2]))
    rl = bool(rng.random() < 0.3)
    tall = bool(rng.random() < 0.2)
    n = int(rng.choice([1, 2, 63, 64, 65, 511, 512, 513, 1025])) if rng.random() < 0.3 else int(10 ** rng.uniform(0, 4.4))
    seed = int(rng.integers(1 << 30))
    if tall:
        s, st = _tall_case(int(rng.choice([131, 201, 301, 451])), n, seed=seed)
        s.saturate_online = sat
    else:
        s, st = _random_case(n, seed, sat, "vector" if rng.random() < 0.5 else "uniform", bool(rng.random() < 0.7))
        st[0] = st[0] * 1e-3
    r2 = np.random.default_rng(seed + 1)
    st[1] = r2.uniform(0, 2 * np.pi, n)
    if hprop:
        st[2] = r2.uniform(-1.2, 1.2, n)
    col = None
    if nz:
        col = 0.01 * (1 + 0.3 * np.sin(s.grids / r2.uniform(10e3, 40e3) + r2.uniform(0, 6)) + 0.1 * s.grids / s.grids[-1])
        s.bvf = col
    s.hprop = hprop
    if direct:                                             # densities around the cap, weak forcing (tests/test_gpu_chain.py)
        s.kappa = 1e-4
        z = np.zeros(n)
        cap = orc.saturation(s, 60.0, np.full(n, np.inf), st[3], z, st[4], z, st[5], st[6], st[7], z, direct=True)
        pv = s.dkk * s.dll * s.rr_mm_area / st[4]
        ok = np.isfinite(cap) & (cap > 0)
        st[0] = np.where(ok, cap / pv * r2.uniform(0.2, 3.0, n), st[0])
    calls = [int(x) for x in rng.integers(1, 4, size=int(rng.integers(1, 3)))]
    flags = (_capi.DIRECT_SAT_QUIRK if direct == 1 else _capi.DIRECT_SAT if direct == 2 else 0) | (_capi.RELAUNCH if rl else 0)
    want, _, _ = oracle_loop(s, st, 60.0, sum(calls), direct=direct, relaunch=1e-6 if rl else None)
    p = make_chain_prop(s, st, hprop, col, dtype="f32" if f32 else "f64")
    p.set_relaunch(1e-6)
    for k in calls:
        p.step(60.0, k, flags)
    got = chain_state(p, st, hprop, nz)
    p.close()
    # float64: one ray in 1e8 amplifies the device sin / cos ulps to 1e-9; the tall column (rays up to 155 km, where the
    # density is 1e-9 of the ground's) amplifies summation-order noise by orders of magnitude per step (tests/test_gpu_parity.py,
    # _tall_case): 1.8e-8 seen after five steps
    tol = 1e-4 if f32 else (1e-6 if tall else 5e-9)
    for i, k in enumerate(STATE_KEYS[:9]):
        w = np.asarray(want[i], dtype=np.float64)
        fin = np.isfinite(w)
        scale = np.max(np.abs(w[fin])) if fin.any() else 0.0
        err = np.abs(got[i] - w)
        e = np.where(np.isnan(got[i]) & np.isnan(w), 0.0, err / (np.abs(w) + 1e-2 * scale + 1e-300))
        bad = np.mean(e > tol)
        # float32: a ray can take the other side of a saturation / relaunch threshold (it then differs by orders of magnitude
        # in dens, or sits at its source instead of its evolved position): those rays are counted, not hidden
        lim = (3e-2 if (direct or rl) else 1e-2 if sat else 3e-3) if f32 else 0.0
        if bad > lim and not (f32 and n < 200 and np.sum(e > tol) <= 1):
            print(f"FAIL n={n} seed={seed} hprop={hprop} nz={nz} f32={f32} sat={sat} direct={direct} rl={rl} tall={tall} calls={calls} "
                  f"slot={k} worst={float(np.nanmax(e)):.3e} bad={bad:.4f}", flush=True)
            sys.exit(1)
        m = float(np.nanmax(np.where(e > tol, 0.0, e))) if e.size else 0.0
        if f32: worst32 = max(worst32, m)
        else: worst64 = max(worst64, float(np.nanmax(e)) if e.size else 0.0)
    cases += 1
    if cases % 25 == 0:
        print(f"{cases} cases ok, worst float64 {worst64:.2e}, float32 (within tolerance) {worst32:.2e}; last n={n} hprop={hprop} nz={nz} "
              f"f32={f32} sat={sat} direct={direct} rl={rl} tall={tall} calls={calls}", flush=True)
print(f"done: {cases} cases, worst float64 {worst64:.2e}, float32 {worst32:.2e}")
