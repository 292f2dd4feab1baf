"""-m gpu: the HIP path (through the C ABI) against (1) golden vectors from the
real reference, (2) the CPU oracles on seeded inputs, (3) size-independent
properties at BASELINE sizes.

Tolerances (fp64; north_star: per-ray state within rtol 1e-10):
  * single RHS / single step, per ray: rtol 1e-12  
  * flux profile / mean-flow tendencies: |err| <= 1e-12 * max|profile| (reduction order differs)
  * multi-step well-posed horizons: rtol 1e-10     (P2/P3)
"""
import os

import numpy as np
import pytest

from oracle import msgwam_oracle as orc
from oracle.c_oracle import COracle
from helpers import STATE_KEYS, load, setup_from, state_from, relerr
from gpu_helpers import make_prop, gpu_state
from msgwam_amd import _capi

pytestmark = pytest.mark.gpu

EVOLVING = ("dens", "rr", "mm", "uu", "vv")


def prof_err(a, b):
    s = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (s if s > 0 else 1.0))


def check_state(got, want, tol_ray, tol_col, what=""):
    """Per-ray slots: element-wise relative error.  The wind columns uu, vv are the two
    components of one vector and a component can be a pure cancellation residue (vv ~ 1e-20
    when all rays share an azimuth pair), so their error is measured against the common
    velocity scale max(|uu|, |vv|)."""
    for k, a, b in zip(STATE_KEYS, got, want):
        if k in ("dens", "rr", "mm"):
            e = relerr(a, b)
            assert e <= tol_ray, (what, k, e)
    scale = max(np.max(np.abs(want[9])), np.max(np.abs(want[10])), 1e-300)
    for k, i in (("uu", 9), ("vv", 10)):
        e = float(np.max(np.abs(got[i] - want[i])) / scale)
        assert e <= tol_col, (what, k, e)


@pytest.mark.parametrize("name", ["g1_rhs_f0_sat0", "g1_rhs_f0_sat1", "g1_rhs_f45_sat0", "g1_rhs_f45_sat1"])
def test_single_rhs_vs_reference_golden(name):
    d = load(name)
    p = make_prop(setup_from(d), state_from(d, "in"))
    out = p.rhs(float(d["dt"]))
    for k in ("dens", "rr", "mm"):
        assert relerr(out[k], d[f"out_{k}"]) <= 1e-12, k
    assert prof_err(out["pm_flux"][:, 1:-1], d["pm_flux_inner"]) <= 1e-12
    for k in ("uu", "vv"):
        assert prof_err(out[k], d[f"out_{k}"]) <= 1e-12, k
    p.close()


@pytest.mark.parametrize("name,marks,flags,tol", [
    ("g3_rk3_coupled_driver", (1, 10, 100), 0, 1e-10),
    ("g3_rk3_coupled_f45", (1, 10, 100), 0, 1e-10),
    ("g3_rk3_fixedbg_config1", (1, 10, 100, 1000), _capi.FIXED_BACKGROUND, 1e-10),
    # strong-amplitude coupled case: chaotic (summation-order noise grows x10 per step, also between
    # two CPU runs that differ only in ray order), so parity is asserted on steps 1 and 5 only
    ("g4_saturation_online", (1, 5), 0, 1e-10),
    ("g4_saturation_online_fixedbg", (1, 20, 60), _capi.FIXED_BACKGROUND, 1e-10),
    ("g5_spectrum_coupled", (1, 3), 0, 1e-10),
])
def test_rk3_vs_reference_golden(name, marks, flags, tol):
    d = load(name)
    st = state_from(d, "in")
    p = make_prop(setup_from(d), st)
    done = 0
    for n in marks:
        p.step(float(d["dt"]), n - done, flags)
        done = n
        check_state(gpu_state(p, st), state_from(d, f"s{n}"), tol, tol, (name, n))
    p.close()


def test_driver_loop_direct_saturation_quirk():
    """raytracer.py:157-188: 1440 steps, post-step saturation with the `/1` quirk."""
    d = load("g4_saturation_direct_driver")
    st = state_from(d, "in")
    p = make_prop(setup_from(d), st)
    done = 0
    for n in (1, 10, 100, 709, 710, 711, 1000):
        p.step(float(d["dt"]), n - done, _capi.DIRECT_SAT_QUIRK)
        done = n
        check_state(gpu_state(p, st), state_from(d, f"s{n}"), 1e-10, 1e-10, n)
    # the first saturation event is at step 710
    assert np.array_equal(d["s709_dens"], d["in_dens"]) and not np.array_equal(d["s710_dens"], d["in_dens"])
    p.close()


def _random_case(n, seed, sat, phi_mode="uniform", sorted_z=False):
    rng = np.random.default_rng(seed)
    grid = np.linspace(0, 100e3, 101)
    area = rng.uniform(1e-3, 1e-1, n)
    s = orc.Setup(grid, phi0=0.4, kappa=0.95, saturate_online=sat,
                  dkk=np.full(n, 1e-4), dll=np.full(n, 1e-4), rr_mm_area=area)
    uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 40e3, 10e3)
    vv = 0.3 * uu[::-1].copy()
    s.set_pressure_gradient(uu, vv)
    rr = rng.uniform(-1e3, 105e3, n)
    if sorted_z:
        rr = np.sort(rr)
    drr = rng.uniform(50, 4000, n)
    kk, ll = rng.normal(0, 1e-4, n), rng.normal(0, 1e-4, n)
    mm = rng.normal(0, 2e-3, n)
    phi = np.full(n, 0.4) if phi_mode == "uniform" else rng.uniform(-1.2, 1.2, n)
    st = [rng.uniform(0, 1e9, n), np.zeros(n), phi, rr, drr, kk, ll, mm, area / drr, uu, vv]
    return s, st


@pytest.mark.parametrize("n,seed,sat,phi_mode,sorted_z", [
    (1, 1, False, "uniform", False), (2, 2, True, "uniform", False), (513, 3, False, "vector", False),
    (4097, 4, True, "vector", True), (100_000, 5, False, "uniform", False), (100_001, 6, True, "uniform", True),
])
def test_step_vs_c_oracle_random(n, seed, sat, phi_mode, sorted_z):
    """Ragged sizes, unsorted rays (LDS-atomic deposit path) and sorted rays (DPP path),
    per-ray latitude, online saturation; 2 coupled steps against the C oracle."""
    s, st = _random_case(n, seed, sat, phi_mode, sorted_z)
    want = COracle(s).step(60.0, 2, st)
    p = make_prop(s, st)
    p.step(60.0, 2)
    check_state(gpu_state(p, st), want, 1e-10, 1e-11, n)
    p.close()


def test_fixed_background_and_direct_sat_vs_c_oracle():
    s, st = _random_case(50_001, 11, False, "vector", False)
    want = COracle(s, fixed_background=True).step(60.0, 3, st, direct_sat=2)
    p = make_prop(s, st)
    p.step(60.0, 3, _capi.FIXED_BACKGROUND | _capi.DIRECT_SAT)
    got = gpu_state(p, st)
    for k, a, b in zip(STATE_KEYS, got, want):
        if k in EVOLVING:
            assert relerr(a, b) <= 1e-11, (k, relerr(a, b))
    assert np.array_equal(got[9], st[9])            # the column is frozen
    p.close()


def test_edge_rays_outside_domain_and_nan():
    s, st = _random_case(300, 12, False)
    st[3][:100] = -5e4            # wholly below the ground: no deposit
    st[3][100:200] = 5e5          # far above the top
    st[3][250] = np.nan           # NaN propagates for that ray only
    want = COracle(s).step(60.0, 1, st)
    p = make_prop(s, st)
    p.step(60.0, 1)
    got = gpu_state(p, st)
    for k, a, b in zip(STATE_KEYS, got, want):
        if k in ("rr", "mm"):
            assert np.array_equal(np.isnan(a), np.isnan(b)), k
            m = ~np.isnan(b)
            assert relerr(a[m], b[m]) <= 1e-12, k
        elif k in ("uu", "vv"):
            assert prof_err(a, b) <= 1e-11, k
    p.close()


def test_projection_vs_reference_golden():
    d = load("g2_projection")
    g101 = d["grid101"]
    grids = .5 * (g101[:-1] + g101[1:])
    r = {k: d["rand_" + k] for k in STATE_KEYS[:9] + ["dkk", "dll", "area"]}
    s = orc.Setup(g101, dkk=r["dkk"], dll=r["dll"], rr_mm_area=r["area"])
    uu = np.zeros(100)
    st = [r[k] for k in STATE_KEYS[:9]] + [uu, uu]
    p = make_prop(s, st)
    for gname, G in (("grid", g101), ("grids", grids)):
        for var in (0, 1, 2, 3, 4):                    # 3, 4: the interface variants (lib/libprop.py:199-219)
            want = d[f"rand_{gname}_var{var}"]
            got_res = p.project(var, G)
            got_arr = p.project_arrays(var, 0.01, r["dens"], r["phi"], r["rr"] - .5 * r["drr"],
                                       r["rr"] + .5 * r["drr"], r["kk"], r["ll"], r["mm"] - .5 * r["dmm"],
                                       r["mm"] + .5 * r["dmm"], r["dkk"], r["dll"], r["dmm"], G)
            assert prof_err(got_res, want) <= 1e-12, (gname, var)
            assert prof_err(got_arr, want) <= 1e-12, (gname, var)
    # edge-case table, one ray at a time: weights must match exactly (var = 2 payload is dens = 1)
    e = d["edges"]
    one = np.ones(1)
    for gname, G in (("G10", d["G10"]), ("grid", d["grid11"]), ("grids", d["grids11"])):
        for i in range(len(e)):
            got = p.project_arrays(2, 0.01, one, 0 * one, e[i:i + 1, 0], e[i:i + 1, 1], 2e-4 * one, 1e-4 * one,
                                   -1e-3 * one, -1e-3 * one, one, one, one, G)
            np.testing.assert_array_equal(got, d[f"edgerows_{gname}_var2"][i], err_msg=f"{gname} ray {i}")
        # the whole edge table at the interfaces (payload x volume of the straddling rays, unit weights)
        n = len(e)
        o = np.ones(n)
        for var in (3, 4):
            got = p.project_arrays(var, 0.01, o, 0 * o, e[:, 0], e[:, 1], 2e-4 * o, 1e-4 * o, -1e-3 * o, -1e-3 * o,
                                   o, o, o, G)
            want = d[f"edge_{gname}_var{var}"]
            assert got.shape == want.shape and prof_err(got, want) <= 1e-13, (gname, var)
    p.close()


def test_saturation_arrays_vs_oracle():
    s, st = _random_case(10_000, 21, True)
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    rng = np.random.default_rng(5)
    rr_st, mm_st, drr_st = rng.normal(0, 5, len(rr)), rng.normal(0, 1e-6, len(rr)), rng.normal(0, 0.1, len(rr))
    # put dens around the trigger threshold (max_dens / phase volume) so that about half the rays trigger
    cap = orc.saturation(s, 60.0, np.full_like(dens, np.inf), rr, rr_st, drr, drr_st, kk, ll, mm, mm_st, direct=True)
    pv = s.dkk * s.dll * (s.rr_mm_area / (drr + drr_st * 60.0))
    dens = np.abs(cap / pv) * rng.uniform(0.5, 1.5, len(rr))
    st[0] = dens
    p = make_prop(s, st)
    for direct in (False, True):
        want = orc.saturation(s, 60.0, dens, rr, rr_st, drr, drr_st, kk, ll, mm, mm_st, direct=direct)
        got = p.saturation(60.0, direct, dens, rr, rr_st, drr, drr_st, kk, ll, mm, mm_st,
                           s.dkk, s.dll, s.rr_mm_area)
        assert relerr(got, want) <= 1e-12
        assert 0 < np.count_nonzero(got != (dens if direct else 0.0)) < len(dens)
    p.close()


def test_bitwise_reproducible_and_graph_equals_eager(monkeypatch):
    monkeypatch.setenv("MSGW_PERSIST", "0")      # the per-stage kernel chain is what hipGraph replays
    s, st = _random_case(200_000, 31, False, "uniform", True)
    outs = []
    for graph_steps, flags in ((0, 0), (0, 0), (4, 0)):
        p = make_prop(s, st)
        p.set_tuning(4, graph_steps)
        p.step(60.0, 9, flags)
        outs.append(gpu_state(p, st))
        if graph_steps:
            assert p.counters()["graph_steps"] == graph_steps
        p.close()
    for k, a, b, c in zip(STATE_KEYS, *outs):
        assert np.array_equal(a, b, equal_nan=True), f"run-to-run difference in {k}"
        assert np.array_equal(a, c, equal_nan=True), f"graph replay differs from eager in {k}"


@pytest.mark.parametrize("dtype,n,sorted_z", [("f64", 1_000_000, True), ("f32", 1_250_000, True), ("f64", 300_001, False)])
def test_persistent_kernel_is_reproducible_bit_for_bit(dtype, n, sorted_z):
    """The deposit finishes its cross-lane sums with LDS atomics on a wave-private row (csrc/ray_kernels.h,
    group_sum2_to_lds); unsorted rays take per-lane LDS atomics.  Both must give the same bits every time: three runs of the
    default persistent flavour at the bench sizes (and of an unsorted case) are compared bit for bit."""
    s, st = _random_case(n, 77, dtype == "f32", "uniform", sorted_z)
    st[0] = st[0] * 1e-3
    outs = []
    for _ in range(3):
        p = _capi.Propagator(len(s.grid), n, dtype=dtype)
        dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
        p.set_config(s.bvf, s.phi0, s.kappa, s.saturate_online)
        p.set_column(s.grid, s.grids, s.rhobar, s.pressure_gradient, uu, vv)
        p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, s.dkk, s.dll, s.rr_mm_area)
        p.step(60.0, 3)
        p.step(60.0, 4)
        assert p.counters()["persist_steps"] == 4
        outs.append(list(p.download_rays()) + list(p.download_column()))
        p.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b, equal_nan=True)


def test_result_independent_of_workgroup_geometry():
    """Different blocks-per-CU change the reduction tree; results stay within reduction noise."""
    s, st = _random_case(150_000, 32, False, "uniform", True)
    ref = None
    for bpc in (1, 4, 8):
        p = make_prop(s, st)
        p.set_tuning(bpc, 0)
        p.step(60.0, 2)
        got = gpu_state(p, st)
        p.close()
        if ref is None:
            ref = got
            continue
        for k, a, b in zip(STATE_KEYS, got, ref):
            if k in ("rr", "mm"):
                assert relerr(a, b) <= 1e-12, k


def test_full_size_config3_properties_and_subsample():
    """BASELINE config 3 size (1e6 rays, coupled): strided subsample against the C oracle run on
    the full set for 2 steps, plus invariants (dens, frozen slots untouched; drr*dmm constant)."""
    from msgwam_amd.spectrum import gaussian_spectrum
    n = 1_000_000
    grid = np.linspace(0, 100e3, 101)
    s0 = orc.Setup(grid)
    sp = gaussian_spectrum(n, s0.grids, s0.rhobar, alpha=0.01)
    s = orc.Setup(grid, dkk=sp["dkk"], dll=sp["dll"], rr_mm_area=sp["area"])
    uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 40e3, 10e3)
    vv = np.zeros_like(uu)
    s.set_pressure_gradient(uu, vv)
    st = [sp[k] for k in STATE_KEYS[:9]] + [uu, vv]
    want = COracle(s).step(120.0, 2, st)
    p = make_prop(s, st)
    p.step(120.0, 2)
    c = p.counters()                                           # the flavour bench.py times is the one held to the oracle
    assert c["persist_steps"] == 2 and c["persist_resident_tiles"] == 4 and c["launch_reducers"] > 0
    assert c["launch_grid"] == c["launch_ray_workgroups"] + c["launch_reducers"] + 1
    assert c["cooperative"] == 1 and c["coop_refused"] == 0        # the runtime vouches for the grid's co-residency
    got = gpu_state(p, st)
    assert np.array_equal(got[0], st[0])                       # dens constant without saturation
    check_state(got, want, 1e-10, 1e-11, "config3")
    assert not np.array_equal(got[9], st[9])                   # the mean flow really moved
    p.close()


def test_full_size_config2_fixed_background_1000_steps():
    """BASELINE config 2 size (1e5 rays, fixed background, 1000 RK3 steps): every ray against the C
    oracle (P2: rtol 1e-10), in ONE call (all steps fused in one launch) and in 10 calls of 100."""
    from msgwam_amd.spectrum import gaussian_spectrum
    n = 100_000
    grid = np.linspace(0, 100e3, 101)
    s0 = orc.Setup(grid)
    sp = gaussian_spectrum(n, s0.grids, s0.rhobar, alpha=0.01)
    s = orc.Setup(grid, dkk=sp["dkk"], dll=sp["dll"], rr_mm_area=sp["area"])
    uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 40e3, 10e3)
    vv = np.zeros_like(uu)
    s.set_pressure_gradient(uu, vv)
    st = [sp[k] for k in STATE_KEYS[:9]] + [uu, vv]
    want = COracle(s, fixed_background=True).step(120.0, 1000, st)
    p = make_prop(s, st)
    p.step(120.0, 1000, _capi.FIXED_BACKGROUND)
    c = p.counters()
    assert c["persist_steps"] == 1000
    assert c["fixed_narrow"] == 1 and c["launch_grid"] == (n + 63) // 64      # one ray per lane: 1563 wavefronts
    one = gpu_state(p, st)
    p.close()
    p = make_prop(s, st)
    for _ in range(10):
        p.step(120.0, 100, _capi.FIXED_BACKGROUND)
    ten = gpu_state(p, st)
    p.close()
    for k, a, b in zip(STATE_KEYS, one, ten):
        assert np.array_equal(a, b), f"1 call of 1000 steps differs from 10 calls of 100 in {k}"
    for i, k in ((0, "dens"), (3, "rr"), (7, "mm")):
        assert relerr(one[i], want[i]) <= 1e-10, (k, relerr(one[i], want[i]))
    assert np.array_equal(one[9], st[9]) and np.array_equal(one[10], st[10])


def test_fixed_background_forms_agree_bitwise(monkeypatch):
    """The two geometries of the fused fixed-background kernel -- one ray per lane in one-wavefront workgroups (small
    ray counts, BASELINE config 2) and two rays per lane in 256-thread workgroups -- run the same per-ray arithmetic:
    bit-identical results, at sizes on either side of every tile boundary, with and without saturation / relaunch."""
    for n, seed, sat, flags in ((1, 1, False, 0), (63, 2, True, 0), (64, 3, False, _capi.DIRECT_SAT), (65, 4, False, _capi.RELAUNCH),
                                (513, 5, True, _capi.RELAUNCH), (100_001, 6, False, 0)):
        s, st = _random_case(n, 300 + seed, sat, "vector" if seed % 2 else "uniform", False)
        res = []
        for narrow in ("1", "0"):
            monkeypatch.setenv("MSGW_FIXED_NARROW", narrow)
            p = make_prop(s, st)
            p.step(60.0, 7, _capi.FIXED_BACKGROUND | flags)
            assert p.counters()["fixed_narrow"] == int(narrow)
            res.append(gpu_state(p, st))
            p.close()
        for k, a, b in zip(STATE_KEYS, *res):
            assert np.array_equal(a, b, equal_nan=True), (n, k)
    monkeypatch.delenv("MSGW_FIXED_NARROW")
    # default choice by size
    s, st = _random_case(500_000, 9, False)
    p = make_prop(s, st)
    p.step(60.0, 1, _capi.FIXED_BACKGROUND)
    assert p.counters()["fixed_narrow"] == 0
    p.close()


def _tall_case(ngrid, n, seed=77):
    rng = np.random.default_rng(seed)
    grid = np.linspace(0, 150e3, ngrid)
    area = rng.uniform(1e-3, 1e-1, n)
    s = orc.Setup(grid, phi0=0.2, kappa=0.9, saturate_online=False, dkk=np.full(n, 1e-4), dll=np.full(n, 1e-4),
                  rr_mm_area=area)
    uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 40e3, 10e3)
    vv = 0.1 * uu[::-1].copy()
    s.set_pressure_gradient(uu, vv)
    rr = np.sort(rng.uniform(-1e3, 155e3, n))
    drr = rng.uniform(50, 1500, n)
    # weak forcing (dens <= 1e7): with dens ~ 1e9 this setup drives uu by 10 m/s per step and
    # amplifies summation-order noise x500 per step (3.7e-14 -> 9e-10 in three steps)
    st = [rng.uniform(0, 1e7, n), np.zeros(n), np.full(n, 0.2), rr, drr, rng.normal(0, 1e-4, n),
          rng.normal(0, 1e-4, n), rng.normal(0, 2e-3, n), area / drr, uu, vv]
    return s, st


@pytest.mark.parametrize("ngrid,env,persist", [(301, {}, True), (301, {"MSGW_REGTILES": "0"}, True), (301, {"MSGW_REGTILES": "2"}, True),
                                               (301, {"MSGW_PERSIST": "0"}, False), (131, {}, True),
                                               (600, {}, True), (600, {"MSGW_SERVICE": "0"}, True)])
def test_tall_column_path_vs_c_oracle(monkeypatch, ngrid, env, persist):
    """More than 130 levels: the threads of the reducer / column / exchange workgroups of the persistent kernel stride
    over the column entries (one each up to 130 levels); as a launch chain the per-level sums stay in LDS and a
    standalone column kernel runs per stage."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s, st = _tall_case(ngrid, 30_011)
    want = COracle(s).step(60.0, 3, st)
    p = make_prop(s, st)
    p.step(60.0, 3)
    check_state(gpu_state(p, st), want, 1e-10, 1e-11, "tall")
    assert p.counters()["persist_steps"] == (3 if persist else 0)
    p.close()


def test_tall_column_through_the_one_rank_exchange(monkeypatch):
    """ngrid = 301 through the exchange workgroup (rows wider than one workgroup) == without it, bit for bit."""
    s, st = _tall_case(301, 20_003, seed=5)
    p = make_prop(s, st)
    p.step(60.0, 4)
    ref = gpu_state(p, st)
    p.close()
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    p = make_prop(s, st)
    p.comm_init(_capi.comm_unique_id(), 0, 1)
    assert p.counters()["exchange"] == 1
    p.step(60.0, 4)
    got = gpu_state(p, st)
    assert p.counters()["persist_steps"] == 4
    p.close()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)


def test_collective_chain_with_one_rank_communicator(monkeypatch):
    """The multi-GPU chain exercised with a 1-rank RCCL communicator (an all-reduce over one rank is
    the identity): lagged launch chain = k_deposit_only, then per pass k_ray_stage<LAG> on stream A
    while k_column<reduce> + ncclAllReduce of the next flux run on stream B.  It deposits exactly the
    values the plain launch chain deposits (same state, same group order), so the results must be
    BITWISE equal to the plain chain (MSGW_PERSIST=0, no communicator)."""
    s, st = _random_case(120_000, 41, False, "uniform", True)
    monkeypatch.setenv("MSGW_PERSIST", "0")
    ref = make_prop(s, st)
    ref.step(60.0, 2)
    ref.step(60.0, 5)
    want = gpu_state(ref, st)
    ref.close()
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    for graph_steps in (0, 2):
        p = make_prop(s, st)
        p.comm_init(_capi.comm_unique_id(), 0, 1)
        p.set_tuning(4, graph_steps)
        p.step(60.0, 2)
        p.step(60.0, 5)
        got = gpu_state(p, st)
        assert p.counters()["nranks"] == 1
        p.close()
        for k, a, b in zip(STATE_KEYS, got, want):
            assert np.array_equal(a, b, equal_nan=True), f"lagged collective chain differs from the plain chain in {k}"
    # and with online saturation + per-ray latitude (other kernel variants of the chain)
    s2, st2 = _random_case(40_001, 42, True, "vector", True)
    monkeypatch.delenv("MSGW_FORCE_COLLECTIVE")
    ref = make_prop(s2, st2)
    ref.step(60.0, 3)
    want = gpu_state(ref, st2)
    ref.close()
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    p = make_prop(s2, st2)
    p.comm_init(_capi.comm_unique_id(), 0, 1)
    p.step(60.0, 3)
    got = gpu_state(p, st2)
    p.close()
    for k, a, b in zip(STATE_KEYS, got, want):
        assert np.array_equal(a, b, equal_nan=True), f"lagged chain (saturation) differs in {k}"


def test_persistent_kernel_equals_per_stage_kernels(monkeypatch):
    """The persistent multi-stage kernel must reproduce the chain of per-stage kernels
    (MSGW_PERSIST=0) and the oracle, over enough steps for its re-used hand-off buffers to wrap
    many times (a stale hand-off would show up as an O(1e-3) error, not 1e-11)."""
    s, st = _random_case(300_000, 51, False, "uniform", True)
    st[0] = st[0] * 1e-3                       # mild forcing: keep the comparison out of the chaotic regime
    outs = {}
    for persist in ("0", "1"):
        monkeypatch.setenv("MSGW_PERSIST", persist)
        p = make_prop(s, st)
        p.step(60.0, 3)
        p.step(60.0, 40)
        outs[persist] = gpu_state(p, st)
        assert p.counters()["persist_steps"] == (40 if persist == "1" else 0)
        p.close()
    # reducer workgroups (default) or the last arriver of each group: same rows, same order
    monkeypatch.setenv("MSGW_SERVICE", "0")
    p = make_prop(s, st)
    p.step(60.0, 3)
    p.step(60.0, 40)
    alt = gpu_state(p, st)
    assert p.counters()["persist_steps"] == 40
    p.close()
    for k, a, b in zip(STATE_KEYS, alt, outs["1"]):
        assert np.array_equal(a, b, equal_nan=True), f"last-arriver reduction differs from the reducer workgroups in {k}"
    # the two paths group the workgroup rows differently (16 vs 32 groups), so they agree to
    # summation-order noise, not bit for bit
    check_state(outs["1"], outs["0"], 1e-11, 1e-11, "persist-vs-per-stage")
    want = COracle(s).step(60.0, 43, st)
    check_state(outs["1"], want, 1e-9, 1e-9, "persist-vs-oracle")


@pytest.mark.parametrize("n,flags,fvec", [(1_000_003, 0, False), (123_457, _capi.RELAUNCH, False), (700_001, 0, True),
                                          (2_100_000, 0, False), (777, 0, False)])
def test_persistent_flavours_agree(monkeypatch, n, flags, fvec):
    """Four / two / no register-resident tiles per workgroup (MSGW_REGTILES): the same rays in the same order, only
    the grouping of the flux sums differs.  The default takes four for float64 without saturation."""
    s, st = _random_case(n, 52, False, "vector" if fvec else "uniform", True)
    st[0] = st[0] * 1e-3
    outs = {}
    for tiles in ("4", "2", "0"):
        monkeypatch.setenv("MSGW_REGTILES", tiles)
        p = make_prop(s, st)
        p.step(60.0, 2, flags)
        p.step(60.0, 9, flags)
        outs[tiles] = gpu_state(p, st)
        c = p.counters()
        assert c["persist_steps"] == 9
        want_tiles = int(tiles) if n < 8_000_000 else 0
        assert c["persist_resident_tiles"] == want_tiles, (tiles, c["persist_resident_tiles"])
        p.close()
    check_state(outs["4"], outs["2"], 1e-11, 1e-11, "4 vs 2 resident tiles")
    check_state(outs["4"], outs["0"], 1e-11, 1e-11, "4 resident tiles vs streamed")
    if not flags:
        check_state(outs["4"], COracle(s).step(60.0, 11, st), 1e-10, 1e-10, "4 resident tiles vs oracle")


@pytest.mark.parametrize("n,flags,regtiles,dtype", [(1_000_003, 0, "4", "f64"), (300_001, _capi.RELAUNCH, "2", "f64"),
                                                     (1_250_000, 0, "4", "f32"), (4_100_000, 0, "4", "f64")])
def test_table_prefetch_is_bitwise_neutral(monkeypatch, n, flags, regtiles, dtype):
    """The early poll + shear-table prefetch at the pass boundary of the resident-tile flavours (persist_publish,
    MSGW_PREFETCH=0 | 1, on by default) changes WHEN a workgroup stages the next pass's table and zeroes its wave rows,
    never a value or a summation order: the states with and without it are bit for bit the same (resident tiles only,
    resident + streamed tiles, relaunch variant, float32 rays); likewise the wave-priority balancing (MSGW_BALANCE)."""
    s, st = _random_case(n, 77, False, "uniform", True)
    st[0] = st[0] * 1e-3
    monkeypatch.setenv("MSGW_REGTILES", regtiles)
    outs = {}
    for pre, bal in (("1", "1"), ("0", "1"), ("1", "0")):     # (MSGW_BALANCE: wave priorities of a CU's two workgroups)
        monkeypatch.setenv("MSGW_PREFETCH", pre)
        monkeypatch.setenv("MSGW_BALANCE", bal)
        p = make_prop(s, st, dtype=dtype)
        p.step(60.0, 1, flags)
        p.step(60.0, 7, flags)
        outs[pre + bal] = gpu_state(p, st)
        assert p.counters()["persist_steps"] == 7
        p.close()
    for other in ("01", "10"):
        for a, b in zip(outs["11"], outs[other]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), other


@pytest.mark.parametrize("ngrid", [5, 6, 9])
def test_smallest_columns_vs_c_oracle(ngrid):
    """ngrid = 5 is the smallest column the library accepts (the scratch of the fused mean-flow update, 6*ngrid - 6
    doubles, aliases the 8*(ngrid - 2) doubles of the per-wave flux rows in LDS; msgw_create rejects ngrid < 5):
    persistent kernel and launch chain against the C oracle."""
    rng = np.random.default_rng(100 + ngrid)
    n = 5_003
    grid = np.linspace(0, 40e3, ngrid)
    area = rng.uniform(1e-3, 1e-1, n)
    s = orc.Setup(grid, phi0=0.3, kappa=0.9, dkk=np.full(n, 1e-4), dll=np.full(n, 1e-4), rr_mm_area=area)
    uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 20e3, 10e3)
    vv = 0.2 * uu[::-1].copy()
    s.set_pressure_gradient(uu, vv)
    rr = np.sort(rng.uniform(-2e3, 42e3, n))
    drr = rng.uniform(50, 3000, n)
    st = [rng.uniform(0, 1e8, n), np.zeros(n), np.full(n, 0.3), rr, drr, rng.normal(0, 1e-4, n), rng.normal(0, 1e-4, n),
          rng.normal(0, 2e-3, n), area / drr, uu, vv]
    want = COracle(s).step(60.0, 3, st)
    for env in ({}, {"MSGW_PERSIST": "0"}):
        old = os.environ.get("MSGW_PERSIST")
        os.environ.update(env)
        try:
            p = make_prop(s, st)
            p.step(60.0, 3)
            assert p.counters()["persist_steps"] == (0 if env else 3)
            check_state(gpu_state(p, st), want, 1e-10, 1e-11, (ngrid, env))
            p.close()
        finally:
            os.environ.pop("MSGW_PERSIST", None) if old is None else os.environ.__setitem__("MSGW_PERSIST", old)
    with pytest.raises(_capi.MsgwError, match="ngrid >= 5"):
        _capi.Propagator(4, 100)


def test_flux_carried_between_calls(monkeypatch):
    """A persistent launch leaves the flux of its final state behind; the next launch takes it as F_0 instead of running
    a deposit-only pre-pass -- as long as nothing has touched the state in between and the kernel flavour is the same.
    Same results as with the pre-pass (to summation order), and every way of invalidating it is honoured."""
    s, st = _random_case(200_003, 91, False, "uniform", True)
    st[0] = st[0] * 1e-3
    want = COracle(s).step(60.0, 9, st)

    def run(carry, coop=1):
        monkeypatch.setenv("MSGW_CARRY", "1" if carry else "0")
        monkeypatch.setenv("MSGW_COOP", str(coop))
        p = make_prop(s, st)
        flags = []
        for k in (2, 3, 4):
            p.step(60.0, k)
            flags.append(p.counters()["carried_flux"])
            assert p.counters()["cooperative"] == coop
        out = gpu_state(p, st)
        p.close()
        return out, flags

    a, fa = run(True)
    b, fb = run(False)
    assert fa == [0, 1, 1] and fb == [0, 0, 0]
    # a plain launch instead of hipLaunchCooperativeKernel: the same kernel, the same bits
    a0, _ = run(True, coop=0)
    for k, x, y in zip(STATE_KEYS, a0, a):
        assert np.array_equal(x, y, equal_nan=True), k
    monkeypatch.delenv("MSGW_COOP")
    check_state(a, want, 1e-10, 1e-11, "carried")
    check_state(b, want, 1e-10, 1e-11, "pre-pass")
    check_state(a, b, 1e-12, 1e-12, "carried vs pre-pass")
    monkeypatch.setenv("MSGW_CARRY", "1")
    p = make_prop(s, st)
    p.step(60.0, 1)
    p.step(60.0, 1)
    assert p.counters()["carried_flux"] == 1
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, s.dkk, s.dll, s.rr_mm_area)       # a new state
    p.step(60.0, 1)
    assert p.counters()["carried_flux"] == 0
    p.step(60.0, 1, _capi.RELAUNCH)                              # another kernel flavour
    assert p.counters()["carried_flux"] == 0
    p.step(60.0, 1, _capi.RELAUNCH)
    assert p.counters()["carried_flux"] == 1
    p.step(60.0, 1, _capi.FIXED_BACKGROUND)                      # the state advanced through another kernel
    p.step(60.0, 1, _capi.RELAUNCH)
    assert p.counters()["carried_flux"] == 0
    p.set_config(s.bvf, s.phi0, s.kappa, s.saturate_online)      # the flux depends on the configuration
    p.step(60.0, 1, _capi.RELAUNCH)
    assert p.counters()["carried_flux"] == 0
    p.close()
    # and the whole sequence above against the oracle, so that a stale F_0 anywhere would show
    p = make_prop(s, st)
    p.step(60.0, 1); p.step(60.0, 1)
    p.set_column(s.grid, s.grids, s.rhobar, s.pressure_gradient, uu, vv)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, s.dkk, s.dll, s.rr_mm_area)
    p.step(60.0, 2); p.step(60.0, 3)
    check_state(gpu_state(p, st), COracle(s).step(60.0, 5, st), 1e-10, 1e-11, "after a re-upload")
    p.close()


def test_device_sqrt_and_division_are_correctly_rounded():
    """The kernels' float64 square root, division and exact division by a constant against numpy, BIT FOR BIT, on 2e6
    arguments: the square root for 2^-767 <= x < inf and +-0, inf, NaN, negative arguments; x / y for finite operands with
    |x| >= 2^-968, 2^-1022 <= |y| <= 2^1021 and a quotient in the normal range, and every combination of zeros, infinities and NaNs
    (csrc/real.h: hipcc's IEEE expansions without the parts that serve the extreme exponents beyond); Markstein's x / d for
    d >= 1 (grid spacings in metres and the 3 of the RK scheme) over the whole exponent range."""
    rng = np.random.default_rng(123)
    n = 2_000_000
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.integers(-1074, 1024, n)
    x = np.ldexp(mant, expo)
    x[: n // 4] = rng.uniform(1e-9, 1e-3, n // 4)                   # where omega^2 lives
    x[n // 4: n // 4 + 1000] = np.ldexp(rng.uniform(1, 2, 1000), rng.integers(-767, -760, 1000))   # the edge of sqrt's domain
    sign = rng.random(n) < 0.2
    x[sign] = -x[sign]
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 2.0 ** -767, 5e-324, 2.0 ** -1022,
                        np.finfo(np.float64).max, 1.0, 4.0, 2.0, 3.0])
    x[-len(special):] = special
    # denominators: the exponent range of the physics and beyond, and the special values against every special numerator
    y = np.ldexp(rng.uniform(1.0, 2.0, n), rng.integers(-1022, 1023, n)) * np.where(rng.random(n) < 0.3, -1.0, 1.0)
    y[: n // 2] = rng.uniform(1e-12, 1e3, n // 2) * np.where(rng.random(n // 2) < 0.3, -1.0, 1.0)
    ns = len(special)
    x[-ns * ns - ns:-ns] = np.repeat(special, ns)
    y[-ns * ns - ns:-ns] = np.tile(special, ns)
    p = _capi.Propagator(11, 16)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore", under="ignore"):
        want_s = np.sqrt(x)
        want_t = x / y
        sdom = (x >= 2.0 ** -767) | (x <= 0) | np.isnan(x)         # (negative arguments: NaN either way)
        for d in (3.0, 1000.0, 250.0, 7.5, 1.0):
            got_s, got_q, got_t = p.probe_arith(x, d, y)
            want_q = x / d
            ok = sdom & ~np.isnan(want_s)
            assert np.array_equal(got_s.view(np.uint64)[ok], want_s.view(np.uint64)[ok])
            assert np.array_equal(np.isnan(got_s[sdom]), np.isnan(want_s[sdom]))
            ok = ~np.isnan(want_q)
            # (the sign of a zero quotient and quotients below 2^-960 -- where the exact remainder x - d * q of Markstein's
            # correction step is itself denormal: 1 ulp off between 2^-1022 and 2^-1019 -- are outside what div_const
            # promises: its callers divide heights in metres and RK increments)
            normal = ok & ((np.abs(want_q) >= 2.0 ** -960) | np.isinf(want_q))
            assert np.array_equal(got_q.view(np.uint64)[normal], want_q.view(np.uint64)[normal]), d
            assert np.array_equal(np.isnan(got_q), np.isnan(want_q))
            tiny = ok & ~normal
            assert np.all(np.abs(got_q[tiny] - want_q[tiny]) <= np.abs(want_q[tiny]) * 2.0 ** -52 + 5e-324 * 2)
        # x / y: bit for bit on its domain, and on every special operand
        spec = ~np.isfinite(x) | ~np.isfinite(y) | (x == 0) | (y == 0)
        dom = spec | ((np.abs(x) >= 2.0 ** -968) & (np.abs(y) >= 2.0 ** -1022) & (np.abs(y) <= 2.0 ** 1021) &
                      (np.abs(want_t) >= 2.0 ** -1022) & np.isfinite(want_t))
        assert dom.sum() > 0.8 * n and spec.sum() >= 100
        ok = dom & ~np.isnan(want_t)
        assert np.array_equal(got_t.view(np.uint64)[ok], want_t.view(np.uint64)[ok])
        assert np.array_equal(np.isnan(got_t[dom]), np.isnan(want_t[dom]))
    p.close()
