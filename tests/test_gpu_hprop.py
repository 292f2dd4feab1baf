"""-m gpu: HPROP_GLOBAL = True (lib/libprop.py:5, libprop's own default): horizontal
propagation on the sphere -- lam, phi, kk, ll evolve too.  Goldens come from the real reference
(oracle/gen_golden.py g6).  sin/cos/tan are the device library's, so per-ray results agree with numpy to a
few ulp, not bit for bit: rtol 1e-10 (north_star), with an absolute floor of 1e-12 x the slot's largest
magnitude for values that are cancellation residues."""
import numpy as np
import pytest

from oracle import msgwam_oracle as orc
from helpers import STATE_KEYS, load, setup_from, state_from
from msgwam_amd import _capi
from test_gpu_parity import _random_case

pytestmark = pytest.mark.gpu


def _close(got, want, rtol=1e-10, what=""):
    scale = np.max(np.abs(want[np.isfinite(want)])) if np.any(np.isfinite(want)) else 0.0
    err = np.abs(got - want)
    ok = (err <= rtol * np.abs(want) + 1e-12 * scale) | (np.isnan(got) & np.isnan(want))
    assert np.all(ok), (what, float(np.nanmax(err / np.maximum(np.abs(want), 1e-300))))


def _prop(s, st):
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    p = _capi.Propagator(len(s.grid), len(dens))
    p.set_config(s.bvf, s.phi0, s.kappa, s.saturate_online, hprop=True)
    p.set_column(s.grid, s.grids, s.rhobar, s.pressure_gradient, uu, vv)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, s.dkk, s.dll, s.rr_mm_area)
    p.upload_hprop(lam, phi)
    return p


def _state(p, st):
    dens, rr, mm = p.download_rays()
    lam, phi, kk, ll = p.download_hprop()
    uu, vv = p.download_column()
    out = [np.asarray(a, dtype=np.float64).copy() for a in st]
    out[0], out[1], out[2], out[3], out[5], out[6], out[7], out[9], out[10] = dens, lam, phi, rr, kk, ll, mm, uu, vv
    return out


@pytest.mark.parametrize("name", ["g6_hprop_rhs_sat0", "g6_hprop_rhs_sat1"])
def test_hprop_single_rhs_vs_reference_golden(name):
    d = load(name)
    s = setup_from(d)
    st = state_from(d, "in")
    p = _prop(s, st)
    t = p.rhs(float(d["dt"]))
    lam, phi, kk, ll = p.download_hprop(tendencies=True)
    p.close()
    got = dict(dens=t["dens"], lam=lam, phi=phi, rr=t["rr"], kk=kk, ll=ll, mm=t["mm"], uu=t["uu"], vv=t["vv"])
    for k, v in got.items():
        _close(v, d[f"out_{k}"], what=k)


def test_hprop_rk3_vs_reference_golden():
    d = load("g6_hprop_rk3_coupled")
    s = setup_from(d)
    st = state_from(d, "in")
    p = _prop(s, st)
    done = 0
    for n in (1, 5, 20, 100):
        p.step(float(d["dt"]), n - done)
        done = n
        got = _state(p, st)
        for i, k in enumerate(STATE_KEYS):
            _close(got[i], d[f"s{n}_{k}"], what=(n, k))
    assert p.counters()["persist_steps"] == 0
    p.close()


def test_hprop_random_rays_vs_oracle():
    """Ragged size, rays at scattered latitudes, coupled, 3 steps against the numpy oracle's spherical branch."""
    s, st = _random_case(30_001, 91, False, "vector", True)
    st[0] = st[0] * 1e-3
    st[1] = np.random.default_rng(5).uniform(0, 2 * np.pi, len(st[0]))
    s.hprop = True
    want = st
    for _ in range(3):
        want = orc.rk3(s, 60.0, want)
    p = _prop(s, st)
    p.step(60.0, 3)
    got = _state(p, st)
    p.close()
    for i, k in enumerate(STATE_KEYS):
        _close(got[i], want[i], what=k)


def test_hprop_call_order_and_scope_errors():
    s, st = _random_case(100, 92, False)
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    p = _capi.Propagator(len(s.grid), 100)
    p.set_config(s.bvf, s.phi0, s.kappa, False, hprop=True)
    p.set_column(s.grid, s.grids, s.rhobar, s.pressure_gradient, uu, vv)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, s.dkk, s.dll, s.rr_mm_area)
    with pytest.raises(_capi.MsgwError, match="msgw_upload_hprop"):
        p.step(60.0, 1)
    p.upload_hprop(lam, phi)
    p.step(60.0, 1)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, s.dkk, s.dll, s.rr_mm_area)   # lam, phi belonged to the old rays
    with pytest.raises(_capi.MsgwError, match="msgw_upload_hprop"):
        p.step(60.0, 1)
    p.close()


def test_hprop_through_the_allreduce_column_path(monkeypatch):
    """Several ranks: the HPROP chain reduces its flux rows, all-reduces them (RCCL) and updates the column.
    With a 1-rank communicator the all-reduce is the identity: results must be BITWISE the plain chain's."""
    s, st = _random_case(20_000, 93, False, "vector", True)
    st[0] = st[0] * 1e-3
    p = _prop(s, st)
    p.step(60.0, 4)
    want = _state(p, st)
    p.close()
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    p = _prop(s, st)
    p.comm_init(_capi.comm_unique_id(), 0, 1)
    p.step(60.0, 4)
    got = _state(p, st)
    p.close()
    for k, a, b in zip(STATE_KEYS, got, want):
        assert np.array_equal(a, b, equal_nan=True), k


def test_hprop_on_a_tall_column():
    """More than 130 levels: the stage kernel leaves sparse rows and the separate reduce kernel sums them."""
    from test_gpu_parity import _tall_case
    rng = np.random.default_rng(3)
    s, st = _tall_case(301, 6_007, seed=12)
    st[2] = rng.uniform(-1.0, 1.0, len(st[0]))                 # scattered latitudes
    s.hprop = True
    want = st
    for _ in range(2):
        want = orc.rk3(s, 60.0, want)
    p = _prop(s, st)
    p.step(60.0, 2)
    got = _state(p, st)
    p.close()
    for i, k in enumerate(STATE_KEYS):
        _close(got[i], want[i], what=("tall", k))
