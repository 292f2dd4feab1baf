"""-m gpu: the GENERAL per-stage kernel (csrc/chain_kernels.h) in the combinations round 2 rejected: HPROP_GLOBAL = True
(lib/libprop.py:5) and / or the N(z) column extension together with the driver's direct saturation
(raytracer.py:182-188), the relaunch extension, both at once, and a float32 ray state.
HPROP + direct saturation is a combination of the REFERENCE (its libprop default with its driver's post-step call):
held to the numpy restatement of exactly that composition (oracle.driver_step with setup.hprop), which is pinned to the
reference by goldens g6 (HPROP rhs / RK3) and g4 / g3 (saturation, driver step).  Everything with N(z) or relaunch is
an extension: the oracle is the definition (parity unpinned by nature, as in test_gpu_nz.py / test_gpu_relaunch.py)."""
import numpy as np
import pytest

from oracle import msgwam_oracle as orc
from helpers import STATE_KEYS, relerr
from msgwam_amd import _capi
from test_gpu_parity import _random_case, prof_err
from test_gpu_nz import _column

pytestmark = pytest.mark.gpu


def make_chain_prop(s, st, hprop, column=None, dtype="f64"):
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    p = _capi.Propagator(len(s.grid), len(dens), dtype=dtype)
    p.set_config(float(np.mean(column)) if column is not None else s.bvf, s.phi0, s.kappa, s.saturate_online, hprop=hprop)
    if column is not None:
        p.set_bvf_column(column)
    p.set_column(s.grid, s.grids, s.rhobar, s.pressure_gradient, uu, vv)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, s.dkk, s.dll, s.rr_mm_area)
    if hprop:
        p.upload_hprop(lam, phi)
    return p


def chain_state(p, st, hprop, nz):
    out = [np.asarray(a, dtype=np.float64).copy() for a in st]
    out[0], out[3], out[7] = p.download_rays()
    out[9], out[10] = p.download_column()
    if hprop:
        out[1], out[2], out[5], out[6] = p.download_hprop()
    if nz:
        out[4], out[8] = p.download_extents()
    return out


def close(got, want, rtol, what, outliers=0.0):
    """Per-ray slots: |got - want| <= rtol |want| + 1e-12 x the slot's largest magnitude (cancellation residues of the
    spherical terms); at most `outliers` of the rays may miss it (threshold decisions of a float32 state)."""
    for i, k in enumerate(STATE_KEYS[:9]):
        w = np.asarray(want[i], dtype=np.float64)
        fin = np.isfinite(w)
        scale = np.max(np.abs(w[fin])) if fin.any() else 0.0
        err = np.abs(got[i] - w)
        ok = (err <= rtol * np.abs(w) + max(rtol * 1e-2, 1e-12) * scale) | (np.isnan(got[i]) & np.isnan(w))
        assert np.mean(~ok) <= outliers, (what, k, float(np.nanmax(err / np.maximum(np.abs(w), 1e-300))), float(np.mean(~ok)))
    for k, i in (("uu", 9), ("vv", 10)):
        assert prof_err(got[i], want[i]) <= max(rtol, 1e-10) * 10, (what, k)


def oracle_loop(s, st, dt, nsteps, direct=0, relaunch=None):
    """rk3 (or the driver's step: direct = 1 with its quirk, 2 with '/ dt') followed by the relaunch rule."""
    src = (st[0].copy(), st[3].copy(), st[7].copy())
    cur = [np.asarray(a, dtype=np.float64).copy() for a in st]
    hits = recycled = 0
    for _ in range(nsteps):
        if direct:
            cur, prop = orc.driver_step(s, dt, cur, ref_quirks=direct == 1)
            hits += int(np.sum(cur[0] != prop))
        else:
            cur = orc.rk3(s, dt, cur)
        if relaunch is not None:
            cur, mask = orc.relaunch(s, cur, src, relaunch)
            recycled += int(mask.sum())
    return cur, hits, recycled


def _case(n, seed, sat, hprop, nz, near_cap=False):
    s, st = _random_case(n, seed, sat, "vector", True)
    rng = np.random.default_rng(seed + 1000)
    st[1] = rng.uniform(0, 2 * np.pi, n)
    col = None
    if nz:
        col = _column(s.grids, seed)
        s.bvf = col
    s.hprop = hprop
    if near_cap:                                             # wave action densities around the saturation cap (:601-604)
        # (a small kappa: the cap, hence the flux of a packet that sits at it, scales with kappa**2 -- with the default the
        # mean flow moves so fast that one ray's threshold decision is felt by every ray near it within a step, which
        # tests the chaos of the system, not the kernel)
        s.kappa = 1e-4
        z = np.zeros(n)
        cap = orc.saturation(s, 60.0, np.full(n, np.inf), st[3], z, st[4], z, st[5], st[6], st[7], z, direct=True)
        pv = s.dkk * s.dll * s.rr_mm_area / st[4]
        ok = np.isfinite(cap) & (cap > 0)
        st[0] = np.where(ok, cap / pv * rng.uniform(0.2, 3.0, n), st[0] * 1e-3)
    return s, st, col


@pytest.mark.parametrize("hprop,nz", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("direct", [1, 2])
def test_direct_saturation_in_the_chain(hprop, nz, direct):
    """The driver's step (RK3, then the direct saturation on the NEW kk, ll, the OLD rr, mm, drr and the tendencies the
    driver forms, raytracer.py:182-188) with HPROP on and / or an N(z) column, four steps in two calls."""
    s, st, col = _case(20_011, 300 + 10 * direct + 2 * hprop + nz, False, hprop, nz, near_cap=True)
    want, hits, _ = oracle_loop(s, st, 60.0, 4, direct=direct)
    assert hits > 100                                        # the case really saturates
    p = make_chain_prop(s, st, hprop, col)
    flag = _capi.DIRECT_SAT_QUIRK if direct == 1 else _capi.DIRECT_SAT
    p.step(60.0, 1, flag)
    p.step(60.0, 3, flag)
    assert p.counters()["persist_steps"] == 0
    got = chain_state(p, st, hprop, nz)
    p.close()
    close(got, want, 1e-10, ("direct", hprop, nz, direct))
    # without the flag dens is untouched
    p = make_chain_prop(s, st, hprop, col)
    p.step(60.0, 4)
    plain = chain_state(p, st, hprop, nz)
    p.close()
    assert np.array_equal(plain[0], st[0]) and not np.array_equal(got[0], st[0])


@pytest.mark.parametrize("hprop,nz,sat", [(True, False, False), (False, True, True), (True, True, False)])
def test_relaunch_in_the_chain(hprop, nz, sat):
    """Rays leave through the top and the bottom, or (online saturation) break: after every step the slot returns to the
    (dens, rr, mm) it was uploaded with; lam, phi, kk, ll, drr, dmm keep what they have evolved to."""
    s, st, col = _case(20_003, 340 + 2 * hprop + nz, sat, hprop, nz)
    if not sat:
        st[0] = st[0] * 1e-3
    want, _, n = oracle_loop(s, st, 60.0, 12, relaunch=1e-6)
    assert n > 100
    p = make_chain_prop(s, st, hprop, col)
    p.set_relaunch(1e-6)
    p.step(60.0, 5, _capi.RELAUNCH)
    p.step(60.0, 7, _capi.RELAUNCH)
    got = chain_state(p, st, hprop, nz)
    p.close()
    close(got, want, 1e-10, ("relaunch", hprop, nz))
    p = make_chain_prop(s, st, hprop, col)
    p.step(60.0, 12)
    plain = chain_state(p, st, hprop, nz)
    p.close()
    assert not np.array_equal(plain[3], got[3])


def test_direct_saturation_and_relaunch_together_with_everything_on():
    s, st, col = _case(10_007, 371, False, True, True, near_cap=True)
    want, hits, n = oracle_loop(s, st, 60.0, 6, direct=2, relaunch=1e-6)
    assert hits > 50 and n > 50
    p = make_chain_prop(s, st, True, col)
    p.set_relaunch(1e-6)
    p.step(60.0, 6, _capi.DIRECT_SAT | _capi.RELAUNCH)
    got = chain_state(p, st, True, True)
    p.close()
    close(got, want, 1e-10, "direct + relaunch, HPROP + N(z)")


@pytest.mark.parametrize("n,seed,sat", [(5, 1, False), (4097, 2, True), (60_001, 3, True)])
def test_hprop_with_a_height_dependent_column_vs_the_definition(n, seed, sat):
    """HPROP_GLOBAL = True AND N(z): all nine per-ray slots evolve.  One RHS (all 11 tendencies) at rtol 1e-10 of each
    slot's scale and three steps against the numpy restatement that composes the two."""
    s, st, col = _case(n, 380 + seed, sat, True, True)
    want_t = orc.rhs(s, 60.0, st)
    p = make_chain_prop(s, st, True, col)
    t = p.rhs(60.0)
    lam, phi, kk, ll = p.download_hprop(tendencies=True)
    ddrr, ddmm = p.download_extents(tendencies=True)
    got_t = {0: t["dens"], 1: lam, 2: phi, 3: t["rr"], 4: ddrr, 5: kk, 6: ll, 7: t["mm"], 8: ddmm}
    for i, g in got_t.items():
        scale = np.max(np.abs(want_t[i])) or 1.0
        assert np.max(np.abs(g - want_t[i])) <= 1e-10 * scale, STATE_KEYS[i]
    assert np.any(ddrr != 0.0) and np.any(kk != 0.0)
    for k, i in (("uu", 9), ("vv", 10)):
        assert prof_err(t[k], want_t[i]) <= 1e-12, k
    want = st
    for _ in range(3):
        want = orc.rk3(s, 60.0, want)
    p.step(60.0, 1)
    p.step(60.0, 2)
    got = chain_state(p, st, True, True)
    p.close()
    close(got, want, 1e-10, ("HPROP + N(z)", n))


def test_constant_column_with_hprop_is_the_reference_hprop():
    """N(z) = const through the combined kernel reproduces the reference's own HPROP golden (g6): the new path is
    pinned where the reference has an answer; drr, dmm do not move."""
    from helpers import load, setup_from, state_from
    d = load("g6_hprop_rk3_coupled")
    s = setup_from(d)
    st = state_from(d, "in")
    col = np.full(len(s.grids), s.bvf)
    p = make_chain_prop(s, st, True, col)
    done = 0
    for n in (1, 5, 20):
        p.step(float(d["dt"]), n - done)
        done = n
        got = chain_state(p, st, True, True)
        want = [d[f"s{n}_{k}"] for k in STATE_KEYS]
        close(got, want, 1e-10, ("g6 through HPROP + N", n))
        assert np.array_equal(got[4], st[4]) and np.array_equal(got[8], st[8])
    p.close()


@pytest.mark.parametrize("hprop,nz,sat", [(True, False, False), (True, False, True), (False, True, True), (True, True, False)])
def test_float32_state_in_the_chain(hprop, nz, sat):
    """float32 ray state (MSGW_DTYPE_F32) through the general kernel, held to the float64 oracle at float32 tolerance:
    one RHS within 2e-5 of each tendency's scale, three steps within 5e-5 per ray (online saturation: up to 0.1 % of
    the rays may decide a threshold differently)."""
    s, st, col = _case(30_001, 400 + 4 * hprop + 2 * nz + sat, sat, hprop, nz)
    st[0] = st[0] * (1.0 if sat else 1e-3)
    want_t = orc.rhs(s, 60.0, st)
    p = make_chain_prop(s, st, hprop, col, dtype="f32")
    t = p.rhs(60.0)
    got_t = {3: t["rr"], 7: t["mm"]}
    if hprop:
        got_t[1], got_t[2], got_t[5], got_t[6] = p.download_hprop(tendencies=True)
    if nz:
        got_t[4], got_t[8] = p.download_extents(tendencies=True)
    for i, g in got_t.items():
        scale = np.max(np.abs(want_t[i])) or 1.0
        # ddrr_st = cg_rr(N at the upper edge) - cg_rr(N at the lower edge) (:641) cancels to 1e-2 .. 1e-1 of either term:
        # float32 rounding of the terms is 1e-4 of the difference's scale; ddmm_st is proportional to it (:645)
        tol = 5e-4 if i in (4, 8) else 2e-5
        bad = np.mean(np.abs(g - want_t[i]) > tol * scale)
        assert bad <= 1e-4, (STATE_KEYS[i], float(np.max(np.abs(g - want_t[i])) / scale), float(bad))
    want = st
    for _ in range(3):
        want = orc.rk3(s, 60.0, want)
    p.step(60.0, 3)
    got = chain_state(p, st, hprop, nz)
    p.close()
    close(got, want, 5e-5, ("f32", hprop, nz, sat), outliers=1e-3 if sat else 1e-4)


def test_float32_direct_saturation_and_relaunch_with_hprop():
    s, st, col = _case(20_011, 431, False, True, False, near_cap=True)
    want, hits, n = oracle_loop(s, st, 60.0, 3, direct=2, relaunch=1e-6)
    assert hits > 100
    p = make_chain_prop(s, st, True, None, dtype="f32")
    p.set_relaunch(1e-6)
    p.step(60.0, 3, _capi.DIRECT_SAT | _capi.RELAUNCH)
    got = chain_state(p, st, True, False)
    p.close()
    close(got, want, 5e-5, "f32 HPROP direct + relaunch", outliers=2e-3)


def test_module_surface_with_hprop_and_a_bvf_column():
    """`lprop.HPROP_GLOBAL = True` (the reference's default) with `model_config['bvf']` as an array: RK3 returns lam, phi,
    kk, ll, drr, dmm as evolving slots, rhs_default all eleven tendencies."""
    import msgwam_amd.libprop as lprop
    s, st, col = _case(2003, 441, True, True, True)
    lprop.HPROP_GLOBAL = True
    lprop.set_model_setup(bvf=col, rhs=lprop.rhs_default, phi0=s.phi0, kappa=s.kappa, saturate_online=True)
    lprop.grid, lprop.grids, lprop.rhobar, lprop.pressure_gradient = s.grid, s.grids, s.rhobar, s.pressure_gradient
    lprop.set_statics(dkk=s.dkk, dll=s.dll, rr_mm_area=s.rr_mm_area)
    try:
        var = np.empty(11, dtype=object)
        for i, a in enumerate(st):
            var[i] = a
        t = lprop.rhs_default(60.0, var)
        want_t = orc.rhs(s, 60.0, st)
        for i in range(9):
            assert np.max(np.abs(np.asarray(t[i]) - want_t[i])) <= 1e-10 * (np.max(np.abs(want_t[i])) or 1.0), i
        out = lprop.RK3(60.0, lprop.RK3(60.0, var))
        for i in (1, 2, 4, 5, 6, 8):
            assert isinstance(out[i], lprop.DeviceArray), i
        want = orc.rk3(s, 60.0, orc.rk3(s, 60.0, st))
        close([np.asarray(a) for a in out], want, 1e-10, "module surface")
    finally:
        lprop.HPROP_GLOBAL = True
        lprop.set_model_setup(bvf=0.01, saturate_online=True)
        lprop.release_device()


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-11), ("f32", 5e-5)])
def test_resident_projection_with_a_column_and_hprop(dtype, tol):
    """`msgw_project` on a state that carries an N(z) column and evolving phi, kk, ll (HPROP): N at the resident ray
    centre, the CURRENT phi, kk, ll, drr, dmm -- against the numpy projection of the downloaded state."""
    s, st, col = _case(20_003, 451, False, True, True)
    st[0] = st[0] * 1e-3
    p = make_chain_prop(s, st, True, col, dtype=dtype)
    p.step(60.0, 2)
    cur = chain_state(p, st, True, True)
    dens, lam, phi, rr, drr, kk, ll, mm, dmm = cur[:9]
    assert not np.array_equal(phi, st[2])
    lo, up = rr - .5 * drr, rr + .5 * drr
    for var in (0, 1, 2):
        want = orc.wave_projection(dens, lo, up, kk, ll, mm - .5 * dmm, mm + .5 * dmm, phi, s.dkk, s.dll, dmm, s.grids,
                                   orc.bvf_at(s, rr), var=var)
        got = p.project(var, s.grids)
        assert prof_err(np.atleast_2d(got), np.atleast_2d(want)) <= tol, var
    p.close()


@pytest.mark.parametrize("dtype,ngrid,n", [("f32", 301, 6_007), ("f64", 600, 3_001), ("f32", 101, 1), ("f64", 101, 2)])
def test_chain_on_tall_columns_and_tiny_ray_counts(dtype, ngrid, n):
    """Everything on (HPROP, N(z), direct saturation, relaunch) where the launch geometry is unusual: columns with more
    than 130 levels (sparse per-workgroup rows, the separate reduce kernel) and one or two rays (a single, mostly inert
    tile)."""
    from test_gpu_parity import _tall_case
    rng = np.random.default_rng(ngrid + n)
    if ngrid > 101:
        s, st = _tall_case(ngrid, n, seed=21)
    else:
        s, st = _random_case(n, 460 + n, False, "vector", True)
        st[0] = st[0] * 1e-3
    st[1] = rng.uniform(0, 2 * np.pi, n)
    st[2] = rng.uniform(-1.0, 1.0, n)
    col = 0.01 * (1 + 0.25 * np.sin(s.grids / 23e3))
    s.bvf, s.hprop = col, True
    want, _, _ = oracle_loop(s, st, 60.0, 3, direct=2, relaunch=1e-6)
    p = make_chain_prop(s, st, True, col, dtype=dtype)
    p.set_relaunch(1e-6)
    p.step(60.0, 1, _capi.DIRECT_SAT | _capi.RELAUNCH)
    p.step(60.0, 2, _capi.DIRECT_SAT | _capi.RELAUNCH)
    got = chain_state(p, st, True, True)
    p.close()
    close(got, want, 1e-10 if dtype == "f64" else 5e-5, (dtype, ngrid, n), outliers=0.0 if dtype == "f64" else 2e-3)
