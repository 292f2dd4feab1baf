"""GPU box: randomized cross-check of the persistent kernel (all variants, resident tiles, reducer/column
workgroups) against the per-stage launch chain (MSGW_PERSIST=0) over many sizes and call patterns.
usage: python tools/stress.py [seconds] [seed]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
from msgwam_amd import _capi
from test_gpu_parity import _random_case
from gpu_helpers import make_prop, gpu_state
from test_gpu_f32 import make_prop32



def run(budget=120.0, seed=1, max_exp=6.4, verbose=True):
    """Random cases for `budget` seconds; returns (number of cases, worst relative error of the float64 cases); the
    float32 cases are held to 2e-5 of the slot's scale and reported separately."""
    rng = np.random.default_rng(seed)
    TILE = 512
    special = [1, 2, 511, 512, 513, 1023, 1024, 1025, 494 * 512, 494 * 512 + 1, 494 * 1024 - 1, 494 * 1024, 494 * 1024 + 2,
               977 * 1024, 977 * 1024 + 3, 1006 * 1024 + 1, 1_000_000, 1_250_000]
    special = [x for x in special if x <= 10 ** max_exp]
    t_end = time.time() + budget
    n_cases = 0
    worst = 0.0
    worst32 = 0.0
    outliers32 = 0
    saved = {k: os.environ.get(k) for k in ("MSGW_PERSIST", "MSGW_REGTILES", "MSGW_SERVICE", "MSGW_FORCE_COLLECTIVE")}
    try:
        while time.time() < t_end:
            n = int(rng.choice(special)) if rng.random() < 0.4 else int(10 ** rng.uniform(0, max_exp))
            sat = bool(rng.random() < 0.25)
            vec = "vector" if rng.random() < 0.3 else "uniform"
            direct = (not sat) and rng.random() < 0.25
            flags = _capi.DIRECT_SAT if direct else 0
            if rng.random() < 0.2:
                flags |= _capi.RELAUNCH
            f32 = bool(rng.random() < 0.3)                     # float32 ray state: same per-ray arithmetic on both paths,
            tol = 2e-5 if f32 else 1e-9                        # the float64 flux sums differ in order
            calls = [int(x) for x in rng.integers(1, 5, size=int(rng.integers(1, 4)))]
            case_seed = int(rng.integers(1 << 30))
            sorted_z = bool(rng.random() < 0.7)
            s, st = _random_case(n, case_seed, sat, vec, sorted_z)
            st[0] = st[0] * 1e-3
            res = {}
            mode = dict(MSGW_REGTILES=str(int(rng.choice([0, 2, 3, 4, 4]))), MSGW_SERVICE=str(int(rng.random() < 0.7)))
            exchange = rng.random() < 0.25                     # the multi-rank path with a 1-rank communicator
            for persist in ("1", "0"):
                os.environ["MSGW_PERSIST"] = persist
                for k_, v_ in mode.items():
                    os.environ[k_] = v_ if persist == "1" else "1"
                os.environ.pop("MSGW_FORCE_COLLECTIVE", None)
                if exchange and persist == "1":
                    os.environ["MSGW_FORCE_COLLECTIVE"] = "1"
                p = make_prop32(s, st) if f32 else make_prop(s, st)
                if exchange and persist == "1":
                    p.comm_init(_capi.comm_unique_id(), 0, 1)
                    assert p.counters()["exchange"] == 1
                used = []
                for k in calls:
                    p.step(60.0, k, flags)
                    used.append(p.counters()["persist_steps"])
                res[persist] = (gpu_state(p, st), used, p.counters()["persist_resident_tiles"])
                p.close()
            a, b = res["1"][0], res["0"][0]
            scale = max(np.max(np.abs(b[9])), np.max(np.abs(b[10])), 1e-300)
            errs = {}
            for i, k in ((0, "dens"), (3, "rr"), (7, "mm")):
                m = np.isfinite(b[i])
                assert np.array_equal(np.isfinite(a[i]), m), (n, k, "finiteness differs")
                if not m.any():
                    errs[k] = 0.0
                elif f32:      # float32: relative to the slot's scale (a ray at rr = 10 m carries the rounding of one at 1e5 m);
                    # an ill-conditioned ray (kh and m both tiny) may amplify the 1-ulp differences of the two code paths:
                    # at most 1e-4 of the rays may miss the tolerance (seen once in ~4600 cases: one ray, 3e-3)
                    e32 = np.abs(a[i][m] - b[i][m]) / max(np.max(np.abs(b[i][m])), 1e-300)
                    nout = int(np.sum(e32 > tol))
                    assert nout <= max(1, int(1e-4 * e32.size)), (n, k, nout, float(e32.max()), case_seed)
                    errs[k] = float(np.max(np.where(e32 > tol, 0.0, e32)))
                    outliers32 += nout
                    if nout and verbose:
                        j = int(np.argmax(e32))
                        print(f"  float32 outlier in {k}: case seed {case_seed} n={n} ray {j}: chain {b[i][m][j]:.9g} persistent "
                              f"{a[i][m][j]:.9g}; kk {st[5][j]:.3g} ll {st[6][j]:.3g} mm0 {st[7][j]:.3g} rr0 {st[3][j]:.6g} "
                              f"mm now {b[7][j]:.3g}", flush=True)
                else:
                    errs[k] = float(np.max(np.abs(a[i][m] - b[i][m]) / np.maximum(np.abs(b[i][m]), 1e-300)))
            for i, k in ((9, "uu"), (10, "vv")):
                errs[k] = float(np.max(np.abs(a[i] - b[i])) / scale)
            e = max(errs.values())
            if f32:
                worst32 = max(worst32, e)
            else:
                worst = max(worst, e)
            n_cases += 1
            tag = (f"n={n} seed={case_seed} sorted={sorted_z} {'f32' if f32 else 'f64'} sat={sat} {vec} direct={direct} relaunch={bool(flags & _capi.RELAUNCH)} calls={calls} persist_steps={res['1'][1]} "
                   f"resident_tiles={res['1'][2]} {mode} exchange={exchange}")
            if e > tol or res["1"][1] != calls:
                raise AssertionError(f"persistent kernel and launch chain disagree: {tag} {errs}")
            if verbose and n_cases % 10 == 0:
                print(f"{n_cases} cases ok, worst rel err {worst:.2e} (float32 cases: {worst32:.2e} of scale, {outliers32} outlier rays); last: {tag}", flush=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return n_cases, worst


if __name__ == "__main__":
    n_cases, worst = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"done: {n_cases} cases, worst rel err {worst:.2e}")
