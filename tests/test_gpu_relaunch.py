"""-m gpu: the removal + source-relaunch EXTENSION (MSGW_RELAUNCH; BASELINE config 5 / SURVEY 8f rank 2).
The reference has nothing like it, so parity is UNPINNED: oracle/msgwam_oracle.py:relaunch is the
definition and the GPU path must reproduce the oracle's step -> relaunch loop."""
import numpy as np
import pytest

from oracle import msgwam_oracle as orc
from gpu_helpers import make_prop, gpu_state
from msgwam_amd import _capi
from test_gpu_parity import _random_case, check_state

pytestmark = pytest.mark.gpu


def _oracle_loop(s, st, dt, nsteps, frac, fixed=False, direct=False):
    src = (st[0].copy(), st[3].copy(), st[7].copy())
    cur = [np.asarray(a, dtype=np.float64).copy() for a in st]
    recycled = 0
    for _ in range(nsteps):
        if direct:
            cur, _ = orc.driver_step(s, dt, cur, fixed_background=fixed, ref_quirks=False)
        else:
            cur = orc.rk3(s, dt, cur, fixed_background=fixed)
        if fixed:
            cur[9], cur[10] = st[9], st[10]
        cur, mask = orc.relaunch(s, cur, src, frac)
        recycled += int(mask.sum())
    return cur, recycled


def test_relaunch_coupled_rays_leaving_the_column():
    """Coupled run in which rays leave through the top and the bottom: every step the GPU must recycle
    exactly the rays the oracle recycles, to their source values."""
    s, st = _random_case(20_000, 81, False, "uniform", True)
    st[0] = st[0] * 1e-3
    want, n = _oracle_loop(s, st, 60.0, 25, 1e-6)
    assert n > 100                                     # the case really exercises the extension
    p = make_prop(s, st)
    p.step(60.0, 10, _capi.RELAUNCH)
    p.step(60.0, 15, _capi.RELAUNCH)
    assert p.counters()["persist_steps"] == 15         # a compile-time variant of the persistent kernel
    got = gpu_state(p, st)
    p.close()
    check_state(got, want, 1e-10, 1e-10, "relaunch coupled")
    # without the flag the same rays are simply gone: the results must differ
    p = make_prop(s, st)
    p.step(60.0, 25)
    plain = gpu_state(p, st)
    p.close()
    assert not np.array_equal(plain[3], got[3])


def test_relaunch_broken_rays_with_direct_saturation():
    """Direct saturation collapses a breaking ray's dens by ~12 orders of magnitude (the reference's
    one-shot removal); with the extension such a slot is relaunched at its source."""
    s, st = _random_case(8_000, 82, False, "vector", True)
    want, n = _oracle_loop(s, st, 60.0, 6, 1e-6, direct=True)
    assert n > 50
    p = make_prop(s, st)
    p.set_relaunch(1e-6)
    p.step(60.0, 6, _capi.RELAUNCH | _capi.DIRECT_SAT)
    got = gpu_state(p, st)
    p.close()
    check_state(got, want, 1e-10, 1e-10, "relaunch direct saturation")


def test_relaunch_fixed_background_all_steps_in_one_launch():
    s, st = _random_case(30_001, 83, False, "uniform", False)
    want, n = _oracle_loop(s, st, 120.0, 40, 1e-6, fixed=True)
    assert n > 100
    p = make_prop(s, st)
    p.step(120.0, 40, _capi.RELAUNCH | _capi.FIXED_BACKGROUND)
    assert p.counters()["persist_steps"] == 40
    got = gpu_state(p, st)
    p.close()
    for i, k in ((0, "dens"), (3, "rr"), (7, "mm")):
        a, b = got[i], want[i]
        m = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), m), k
        assert np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300)) <= 1e-10, k


def test_relaunch_fraction_argument_is_checked():
    s, st = _random_case(100, 84, False)
    p = make_prop(s, st)
    with pytest.raises(_capi.MsgwError):
        p.set_relaunch(1.5)
    p.set_relaunch(0.0)                                # 0 disables the "broken" criterion
    p.close()
