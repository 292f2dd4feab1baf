"""-m gpu: EXTENSION -- buoyancy frequency as a column N(z) on grids (msgw_set_bvf_column; SURVEY 8f rank 4, north_star
"U(z)/N2(z) column").  The reference has a scalar bvf only, so the height-dependent case is PARITY UNPINNED: the
numpy restatement's definition (bvf_at / bvf_gradient_at) is what the GPU is held to.  The new code path IS pinned to
the reference where it can be: with a constant column every expression reduces to lib/libprop.py's, and the kernel
must reproduce the reference's goldens."""
import numpy as np
import pytest

from oracle import msgwam_oracle as orc
from oracle.c_oracle import COracle
from helpers import STATE_KEYS, load, setup_from, state_from, relerr
from gpu_helpers import gpu_state
from msgwam_amd import _capi
from test_gpu_parity import _random_case, check_state, prof_err
from test_oracle_golden import nz_kat_case

pytestmark = pytest.mark.gpu


def make_prop_nz(setup, state, bvf_column, dtype="f64"):
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = state
    p = _capi.Propagator(len(setup.grid), len(dens), dtype=dtype)
    p.set_config(float(np.mean(bvf_column)), setup.phi0, setup.kappa, setup.saturate_online)
    p.set_bvf_column(bvf_column)
    p.set_column(setup.grid, setup.grids, setup.rhobar, setup.pressure_gradient, uu, vv)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, setup.dkk, setup.dll, setup.rr_mm_area)
    return p


def gpu_state_nz(p, st):
    out = gpu_state(p, st)
    out[4], out[8] = p.download_extents()
    return out


@pytest.mark.parametrize("name", ["g1_rhs_f0_sat0", "g1_rhs_f0_sat1", "g1_rhs_f45_sat0", "g1_rhs_f45_sat1"])
def test_constant_column_single_rhs_vs_reference_golden(name):
    d = load(name)
    s = setup_from(d)
    p = make_prop_nz(s, state_from(d, "in"), np.full(len(s.grids), float(d["bvf"])))
    out = p.rhs(float(d["dt"]))
    for k in ("dens", "rr", "mm"):
        assert relerr(out[k], d[f"out_{k}"]) <= 1e-12, k
    ddrr, ddmm = p.download_extents(tendencies=True)
    assert not np.any(ddrr) and not np.any(ddmm)             # cgr_up == cgr_down when N is constant (:641, :645)
    assert prof_err(out["pm_flux"][:, 1:-1], d["pm_flux_inner"]) <= 1e-12
    for k in ("uu", "vv"):
        assert prof_err(out[k], d[f"out_{k}"]) <= 1e-12, k
    p.close()


@pytest.mark.parametrize("name,marks,flags", [("g3_rk3_coupled_f45", (1, 10, 100), 0), ("g4_saturation_online", (1, 5), 0),
                                              ("g3_rk3_fixedbg_config1", (1, 100), _capi.FIXED_BACKGROUND)])
def test_constant_column_rk3_vs_reference_golden(name, marks, flags):
    d = load(name)
    s = setup_from(d)
    st = state_from(d, "in")
    p = make_prop_nz(s, st, np.full(len(s.grids), float(d["bvf"])))
    done = 0
    for n in marks:
        p.step(float(d["dt"]), n - done, flags)
        done = n
        got = gpu_state_nz(p, st)
        check_state(got, state_from(d, f"s{n}"), 1e-10, 1e-10, (name, n))
        assert np.array_equal(got[4], st[4]) and np.array_equal(got[8], st[8])   # drr, dmm do not move
    assert p.counters()["persist_steps"] == 0                # a per-stage kernel of its own
    p.close()


def _column(grids, seed):
    rng = np.random.default_rng(seed)
    return 0.01 * (1 + 0.3 * np.sin(grids / 17e3 + rng.uniform(0, 6)) + 0.1 * grids / grids[-1])


@pytest.mark.parametrize("n,seed,sat,phi_mode,sorted_z", [(3, 1, False, "uniform", False), (4097, 2, True, "vector", True),
                                                         (60_001, 3, True, "uniform", False)])
def test_height_dependent_column_vs_the_definition(n, seed, sat, phi_mode, sorted_z):
    """Random N(z), coupled, online saturation, rays incl. below-ground / above-top (np.interp clamps): one RHS
    (all 11 tendencies, drr and dmm included) at rtol 1e-12 and three steps at rtol 1e-10 against the numpy
    restatement that defines the extension."""
    s, st = _random_case(n, 40 + seed, sat, phi_mode, sorted_z)
    col = _column(s.grids, seed)
    s.bvf = col
    want_t = orc.rhs(s, 60.0, st)
    p = make_prop_nz(s, st, col)
    t = p.rhs(60.0)
    ddrr, ddmm = p.download_extents(tendencies=True)
    got_t = {0: t["dens"], 3: t["rr"], 4: ddrr, 7: t["mm"], 8: ddmm}
    for i, g in got_t.items():
        scale = np.max(np.abs(want_t[i])) or 1.0
        assert np.max(np.abs(g - want_t[i])) <= 1e-12 * scale, STATE_KEYS[i]
    assert np.any(ddrr != 0.0)
    for k, i in (("uu", 9), ("vv", 10)):
        assert prof_err(t[k], want_t[i]) <= 1e-12, k
    want = st
    for _ in range(3):
        want = orc.rk3(s, 60.0, want)
    p.step(60.0, 1)
    p.step(60.0, 2)
    got = gpu_state_nz(p, st)
    check_state(got, want, 1e-10, 1e-11, ("N(z)", n))
    for i in (4, 8):
        assert relerr(got[i], want[i]) <= 1e-10, STATE_KEYS[i]
    p.close()


def test_height_dependent_column_at_size_vs_c_oracle():
    """3e5 rays, random N(z), online saturation, two calls: against the C restatement of the extension (which agrees
    bit for bit with the numpy definition, tests/test_oracle_c.py)."""
    s, st = _random_case(300_007, 47, True, "uniform", True)
    col = _column(s.grids, 9)
    s.bvf = col
    want = COracle(s).step(60.0, 3, st)
    p = make_prop_nz(s, st, col)
    p.step(60.0, 1)
    p.step(60.0, 2)
    got = gpu_state_nz(p, st)
    check_state(got, want, 1e-10, 1e-11, "N(z) 3e5")
    for i in (4, 8):
        assert relerr(got[i], want[i]) <= 1e-10, STATE_KEYS[i]
    p.close()


def test_nz_chain_through_the_allreduce_column_path_and_on_a_tall_column(monkeypatch):
    """Several ranks: the N(z) chain reduces its flux rows inside the stage kernel, all-reduces the row (RCCL) and
    updates the column; with a 1-rank communicator the all-reduce is the identity, so the results must be BITWISE the
    plain chain's.  On a column with more than 130 levels the rows are reduced by the separate kernel instead."""
    s, st = _random_case(20_000, 48, True, "uniform", True)
    col = _column(s.grids, 4)
    s.bvf = col
    p = make_prop_nz(s, st, col)
    p.step(60.0, 4)
    want = gpu_state_nz(p, st)
    p.close()
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    p = make_prop_nz(s, st, col)
    p.comm_init(_capi.comm_unique_id(), 0, 1)
    p.step(60.0, 4)
    got = gpu_state_nz(p, st)
    p.close()
    for k, a, b in zip(STATE_KEYS, got, want):
        assert np.array_equal(a, b, equal_nan=True), k
    monkeypatch.delenv("MSGW_FORCE_COLLECTIVE")
    # tall column, against the numpy definition
    from test_gpu_parity import _tall_case
    s, st = _tall_case(301, 5_003, seed=8)
    col = 0.01 * (1 + 0.2 * s.grids / s.grids[-1])
    s.bvf = col
    want = st
    for _ in range(2):
        want = orc.rk3(s, 60.0, want)
    p = make_prop_nz(s, st, col)
    p.step(60.0, 2)
    got = gpu_state_nz(p, st)
    p.close()
    check_state(got, want, 1e-10, 1e-11, "N(z) tall")


def test_kat_frequency_conservation_on_the_gpu():
    """The known-answer test of the extension (tests/test_oracle_golden.py) on the GPU: omega is conserved along the
    rays to the time-stepping error while m changes by several per cent."""
    s, st = nz_kat_case(n=5000, seed=7)
    om0 = orc.omega(st[5], st[6], st[7], st[2], orc.bvf_at(s, st[3]))
    p = make_prop_nz(s, st, s.bvf)
    p.step(60.0, 120, _capi.FIXED_BACKGROUND)
    got = gpu_state_nz(p, st)
    p.close()
    om = orc.omega(got[5], got[6], got[7], got[2], orc.bvf_at(s, got[3]))
    assert np.max(np.abs(om / om0 - 1)) <= 1e-5
    assert np.max(np.abs(got[7] / st[7] - 1)) >= 0.03 and np.max(np.abs(got[4] / st[4] - 1)) >= 0.01
    assert np.allclose(got[8] / got[4], st[8] / st[4], rtol=1e-12)


def test_bvf_column_through_the_module_surface_and_scope():
    """`model_config['bvf']` as an array on lprop.grids: RK3 and rhs_default return drr, dmm as evolving slots
    (the combinations with HPROP, direct saturation, relaunch and a float32 state: tests/test_gpu_chain.py)."""
    import msgwam_amd.libprop as lprop
    s, st = _random_case(2000, 55, True, "uniform", True)
    col = _column(s.grids, 5)
    s.bvf = col
    lprop.HPROP_GLOBAL = False
    lprop.set_model_setup(bvf=col, rhs=lprop.rhs_default, phi0=s.phi0, kappa=s.kappa, saturate_online=True)
    lprop.grid, lprop.grids, lprop.rhobar, lprop.pressure_gradient = s.grid, s.grids, s.rhobar, s.pressure_gradient
    lprop.set_statics(dkk=s.dkk, dll=s.dll, rr_mm_area=s.rr_mm_area)
    try:
        var = np.empty(11, dtype=object)
        for i, a in enumerate(st):
            var[i] = a
        t = lprop.rhs_default(60.0, var)
        want_t = orc.rhs(s, 60.0, st)
        for i in (3, 4, 7, 8):
            assert np.max(np.abs(np.asarray(t[i]) - want_t[i])) <= 1e-12 * (np.max(np.abs(want_t[i])) or 1.0), i
        out = lprop.RK3(60.0, var)
        out = lprop.RK3(60.0, out)
        assert isinstance(out[4], lprop.DeviceArray) and isinstance(out[8], lprop.DeviceArray)
        want = orc.rk3(s, 60.0, orc.rk3(s, 60.0, st))
        for i in (0, 3, 4, 7, 8):
            assert relerr(np.asarray(out[i]), want[i]) <= 1e-10, i
    finally:
        lprop.set_model_setup(bvf=0.01, saturate_online=True)
        lprop.release_device()


def test_standalone_saturation_and_projection_use_the_column():
    """`lprop.saturation` (the driver's post-step call, raytracer.py:182-188) and `lprop.wave_projection`
    with `model_config['bvf']` as a column must follow the extension's definition (N at rr_center for omega, at the
    projected height for the cap; N at the ray centre in the projections) -- not a scalar mean of the column."""
    import msgwam_amd.libprop as lprop
    s, st = _random_case(5003, 61, True, "uniform", False)
    col = _column(s.grids, 7)
    s.bvf = col
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    rng = np.random.default_rng(8)
    rr_st = rng.normal(0, 20.0, len(rr))
    drr_st = rng.normal(0, 1e-3, len(rr))
    mm_st = mm * rng.normal(0, 1e-4, len(rr))
    big = dens * 10.0 ** rng.uniform(0, 14, len(rr))                 # so that a good share of the rays saturates
    lprop.HPROP_GLOBAL = False
    lprop.set_model_setup(bvf=col, rhs=lprop.rhs_default, phi0=s.phi0, kappa=s.kappa, saturate_online=False)
    lprop.grid, lprop.grids, lprop.rhobar, lprop.pressure_gradient = s.grid, s.grids, s.rhobar, s.pressure_gradient
    lprop.set_statics(dkk=s.dkk, dll=s.dll, rr_mm_area=s.rr_mm_area)
    try:
        for direct in (True, False):
            want = orc.saturation(s, 60.0, big, rr, rr_st, drr, drr_st, kk, ll, mm, mm_st, direct=direct)
            got = lprop.saturation(60.0, big, rr, rr_st, drr, drr_st, kk, ll, mm, mm_st, direct=direct)
            hit = want != (big if direct else 0.0)
            assert 0.05 * len(rr) < np.count_nonzero(hit) < 0.95 * len(rr)
            assert relerr(got, want) <= 1e-12, direct
            # and it is NOT what a scalar mean of the column gives
            s_mean = orc.Setup(s.grid, bvf=float(np.mean(col)), phi0=s.phi0, kappa=s.kappa, saturate_online=True,
                               dkk=s.dkk, dll=s.dll, rr_mm_area=s.rr_mm_area)
            wrong = orc.saturation(s_mean, 60.0, big, rr, rr_st, drr, drr_st, kk, ll, mm, mm_st, direct=direct)
            assert relerr(wrong, want) > 1e-3
        lo, up = rr - .5 * drr, rr + .5 * drr
        for G in (s.grid, s.grids):
            for var in (0, 1, 2, 3, 4):
                want = orc.wave_projection(dens, lo, up, kk, ll, mm - .5 * dmm, mm + .5 * dmm, phi, s.dkk, s.dll, dmm, G,
                                           orc.bvf_at(s, .5 * (lo + up)), var=var)
                got = lprop.wave_projection(dens, lam, phi, lo, up, kk, ll, mm - .5 * dmm, mm + .5 * dmm, s.dkk, s.dll,
                                            dmm, G, var=var)
                assert got.shape == want.shape
                assert prof_err(np.atleast_2d(got), np.atleast_2d(want)) <= 1e-12, (var, len(G))
    finally:
        lprop.set_model_setup(bvf=0.01, saturate_online=True)
        lprop.release_device()
    # the resident projection (msgw_project): N at the resident ray centre rr
    p = make_prop_nz(s, st, col)
    for var in (0, 1, 2):
        want = orc.wave_projection(dens, lo, up, kk, ll, mm - .5 * dmm, mm + .5 * dmm, phi, s.dkk, s.dll, dmm, s.grids,
                                   orc.bvf_at(s, rr), var=var)
        assert prof_err(np.atleast_2d(p.project(var, s.grids)), np.atleast_2d(want)) <= 1e-12, var
    # ... and of the state two steps later: drr, dmm have evolved, and with them the rays' extents and phase-space volumes
    p.step(60.0, 2)
    cur = gpu_state_nz(p, st)
    d2, r2, dr2, m2, dm2 = cur[0], cur[3], cur[4], cur[7], cur[8]
    assert relerr(dm2, dmm) > 1e-4
    for var in (0, 1, 2):
        want = orc.wave_projection(d2, r2 - .5 * dr2, r2 + .5 * dr2, kk, ll, m2 - .5 * dm2, m2 + .5 * dm2, phi, s.dkk, s.dll, dm2,
                                   s.grids, orc.bvf_at(s, r2), var=var)
        assert prof_err(np.atleast_2d(p.project(var, s.grids)), np.atleast_2d(want)) <= 1e-12, ("after two steps", var)
    p.close()
    # a scalar context asked for "the context's column" says so
    q = _capi.Propagator(len(s.grid), 16)
    with pytest.raises(_capi.MsgwError, match="N\\(z\\) column"):
        q.project_arrays(0, np.nan, dens[:8], phi[:8], lo[:8], up[:8], kk[:8], ll[:8], mm[:8], mm[:8], s.dkk[:8] if np.ndim(s.dkk) else s.dkk,
                         s.dll[:8] if np.ndim(s.dll) else s.dll, dmm[:8], s.grids)
    q.close()
