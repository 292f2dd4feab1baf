"""-m gpu: the drop-in module surface (msgwam_amd.libprop) and the headless driver
against golden vectors from the real reference."""
import numpy as np
import pytest

from helpers import STATE_KEYS, load, state_from, relerr

pytestmark = pytest.mark.gpu


def _configure_from(d, lprop, sat=None):
    lprop.HPROP_GLOBAL = False
    lprop.set_model_setup(bvf=float(d["bvf"]), rhs=lprop.rhs_default, boussinesq=False, sig_rr=10000, u0=4,
                          rr0=40000, rr1=40000, phi0=float(d["phi0"]), kappa=float(d["kappa"]),
                          saturate_online=bool(int(d["saturate_online"])) if sat is None else sat,
                          hh=8500, rhobar0=1.2)
    grid = d["grid"]
    lprop.grid, lprop.grids = grid, .5 * (grid[:-1] + grid[1:])
    lprop.set_hydrostatics()
    np.testing.assert_array_equal(lprop.rhobar, d["rhobar"])
    lprop.pressure_gradient = d["pg"].copy()
    lprop.set_statics(dkk=d["dkk"], dll=d["dll"], rr_mm_area=d["area"])


def _obj(st):
    o = np.empty(11, dtype=object)
    for i, a in enumerate(st):
        o[i] = a
    return o


def _check(got, d, prefix, tol=1e-10):
    scale = max(np.max(np.abs(d[f"{prefix}_uu"])), np.max(np.abs(d[f"{prefix}_vv"])), 1e-300)
    for i, k in enumerate(STATE_KEYS):
        want = d[f"{prefix}_{k}"]
        assert got[i].dtype == np.float64 and got[i].shape == want.shape and len(got[i]) == len(want)
        g = np.asarray(got[i])                             # (evolving slots may be DeviceArrays: copied here)
        assert isinstance(g, np.ndarray) and g.dtype == np.float64
        if k in ("uu", "vv"):
            assert np.max(np.abs(g - want)) / scale <= tol, k
        else:
            assert relerr(g, want) <= tol, (k, relerr(g, want))


def test_rk3_and_rhs_default_match_reference():
    import msgwam_amd.libprop as lprop
    d = load("g3_rk3_coupled_f45")
    _configure_from(d, lprop)
    st = _obj(state_from(d, "in"))
    out = lprop.RK3(float(d["dt"]), st)
    assert out.dtype == object and out.shape == (11,)
    _check(out, d, "s1")
    for _ in range(9):                     # feed the returned object back: state stays resident
        out = lprop.RK3(float(d["dt"]), out)
    _check(out, d, "s10")
    g = load("g1_rhs_f45_sat1")
    _configure_from(g, lprop)
    t = lprop.rhs_default(float(g["dt"]), _obj(state_from(g, "in")))
    _check(t, g, "out", tol=1e-12)
    lprop.release_device()


def test_fixed_background_hook_config1():
    """BASELINE config 1 through the module surface: rhs hook that freezes the column."""
    import msgwam_amd.libprop as lprop
    d = load("g3_rk3_fixedbg_config1")
    _configure_from(d, lprop)
    lprop.set_model_setup(rhs=lprop.rhs_fixed_background)
    out = _obj(state_from(d, "in"))
    for n in range(1, 101):
        out = lprop.RK3(float(d["dt"]), out)
        if n in (1, 10, 100):
            _check(out, d, f"s{n}")
    lprop.set_model_setup(rhs=lprop.rhs_default)
    lprop.release_device()


def test_nray_equal_to_levels_works():
    """nray == ngrid-1 crashes the reference (object-array collapse); the mirror must not."""
    import msgwam_amd.libprop as lprop
    from msgwam_amd import driver
    grid, grids, uu, vv = driver.configure(ngrid=101)
    ic = driver.initial_rays(100, grids)
    st = _obj([ic[k] for k in driver.KEYS] + [uu, vv])
    out = lprop.RK3(120.0, st)
    assert out.shape == (11,) and out[3].shape == (100,) and out[9].shape == (100,)
    assert np.all(np.isfinite(out[3]))
    lprop.release_device()


def test_zero_rays_move_the_mean_flow_alone():
    """Empty input: the reference's numpy code runs with zero rays (no deposit; Coriolis force and pressure gradient
    move the column).  RK3 and rhs_default of the mirror do the same, also as the fixed-background hook."""
    import msgwam_amd.libprop as lprop
    from msgwam_amd import driver
    from oracle import msgwam_oracle as orc
    grid, grids, uu, vv = driver.configure(ngrid=101)
    lprop.model_config['phi0'] = 0.4
    vv = 0.3 * uu[::-1].copy()
    lprop.set_pressure_gradient(uu, 0 * vv)                     # vv is out of balance: it must move
    e = np.zeros(0)
    lprop.set_statics(dkk=e, dll=e, rr_mm_area=e)
    s = orc.Setup(grid, phi0=0.4, dkk=e, dll=e, rr_mm_area=e)
    s.set_pressure_gradient(uu, 0 * vv)
    st = _obj([e] * 9 + [uu, vv])
    want = st
    got = st
    for _ in range(3):
        want = orc.rk3(s, 120.0, want)
        got = lprop.RK3(120.0, got)
    assert got.shape == (11,) and all(np.shape(got[i]) == (0,) for i in range(9))
    for i in (9, 10):
        assert np.max(np.abs(np.asarray(got[i]) - want[i])) <= 1e-13 * np.max(np.abs(want[9]))
    assert np.max(np.abs(np.asarray(got[10]) - vv)) > 1e-6
    t = lprop.rhs_default(120.0, st)
    tw = orc.rhs(s, 120.0, st)
    for i in (9, 10):
        assert np.max(np.abs(t[i] - tw[i])) <= 1e-13 * np.max(np.abs(tw[9]))
    for var in (0, 1, 2, 3, 4):                                 # the other two hot-path entry points, empty as well
        proj = lprop.wave_projection(e, e, e, e, e, e, e, e, e, e, e, e, grids, var=var)
        nG = len(grids)
        assert proj.shape == {0: (2, nG - 1), 1: (nG - 1,), 2: (nG - 1,), 3: (nG,), 4: (2, nG)}[var] and not proj.any()
    assert lprop.saturation(120.0, e, e, e, e, e, e, e, e, e, direct=True).shape == (0,)
    lprop.set_model_setup(rhs=lprop.rhs_fixed_background)
    frozen = lprop.RK3(120.0, st)
    assert np.array_equal(np.asarray(frozen[9]), uu) and np.array_equal(np.asarray(frozen[10]), vv)
    lprop.set_model_setup(rhs=lprop.rhs_default)
    lprop.model_config['phi0'] = 0.0
    lprop.release_device()


def test_scope_errors_are_loud():
    import msgwam_amd.libprop as lprop
    d = load("g3_rk3_coupled_driver")
    _configure_from(d, lprop)
    st = _obj(state_from(d, "in"))
    lprop.set_model_setup(rhs="not callable")
    with pytest.raises(TypeError):
        lprop.RK3(120.0, st)
    lprop.set_model_setup(rhs=lambda dt, v: v[:3])           # a hook must return 11 tendencies
    with pytest.raises(ValueError):
        lprop.RK3(120.0, st)
    lprop.set_model_setup(rhs=lprop.rhs_default)
    with pytest.raises(ValueError):
        lprop.wave_projection(*([np.ones(4)] * 12), d["grid"], var=5)
    lprop.release_device()


@pytest.mark.parametrize("mode,nt,every", [("dropin", 100, 1), ("resident", 1000, 10)])
def test_driver_matches_reference_loop(mode, nt, every):
    """raytracer.py's own loop (60 rays, direct saturation with the `/1` quirk)."""
    from msgwam_amd import driver
    d = load("g4_saturation_direct_driver")
    H = driver.run(nray=60, nt_max=nt, mode=mode, snapshot_every=every, diagnostics=(mode == "dropin"))
    for n in (10, 100, 710, 1000):
        if n > nt or n % every:
            continue
        for k in ("dens", "rr", "mm"):
            assert relerr(H[f"int_{k}"][n], d[f"s{n}_{k}"]) <= 1e-10, (n, k)
        scale = np.max(np.abs(d[f"s{n}_uu"]))
        assert np.max(np.abs(H["int_uu"][n] - d[f"s{n}_uu"])) / scale <= 1e-10
    if mode == "dropin":
        assert relerr(H["int_dens_prop"][100], d["s100_dens_prop"]) <= 1e-10
        # conservation diagnostic (raytracer.py:198-240) against the reference's own rows: wave action on
        # `grid` (var=2) and vertical wave-action flux on `grids` (var=1), projected from the GPU states
        assert H["wa"].shape == (nt + 1, 100) and H["flux_diag"].shape == (nt + 1, 99)
        for n in (1, 10, 100):
            for k in ("wa", "flux_diag"):
                want = d[f"s{n}_{k}"]
                assert np.max(np.abs(H[k][n] - want)) <= 1e-10 * np.max(np.abs(want)), (n, k)


def test_hprop_global_true_through_the_module_surface():
    """lprop.HPROP_GLOBAL = True (libprop's own default): RK3 and rhs_default return all 11 slots, lam, phi,
    kk, ll evolving, as the reference does (goldens generated by the reference, oracle/gen_golden.py g6)."""
    import msgwam_amd.libprop as lprop
    d = load("g6_hprop_rk3_coupled")
    _configure_from(d, lprop)
    lprop.HPROP_GLOBAL = True
    try:
        out = lprop.RK3(float(d["dt"]), _obj(state_from(d, "in")))
        _check(out, d, "s1")
        for _ in range(4):                     # feed the returned object back: state stays resident
            out = lprop.RK3(float(d["dt"]), out)
        _check(out, d, "s5")
        assert not np.array_equal(out[2], d["in_phi"]) and not np.array_equal(out[5], d["in_kk"])
        g = load("g6_hprop_rhs_sat1")
        _configure_from(g, lprop)
        lprop.HPROP_GLOBAL = True              # (_configure_from follows raytracer.py:38 and switches it off)
        t = lprop.rhs_default(float(g["dt"]), _obj(state_from(g, "in")))
        for i, k in enumerate(STATE_KEYS):
            want = g[f"out_{k}"]
            scale = np.max(np.abs(want))
            assert np.all(np.abs(t[i] - want) <= 1e-10 * np.abs(want) + 1e-12 * scale), k
    finally:
        lprop.HPROP_GLOBAL = False
        lprop.release_device()


def test_rhs_hook_around_rhs_default_is_config1():
    """The reference's own way to freeze the mean flow: a user hook that zeroes slots 9, 10 of
    rhs_default's result.  The hook is opaque Python, so the six RK lines (lib/libprop.py:693-698) run on the host;
    its rhs_default calls run on the GPU.  Against the reference's rows of config 1."""
    import msgwam_amd.libprop as lprop
    d = load("g3_rk3_fixedbg_config1")
    _configure_from(d, lprop)
    calls = []

    def frozen_background(dt, var):
        t = lprop.rhs_default(dt, var)
        calls.append(1)
        t[9], t[10] = np.zeros_like(t[9]), np.zeros_like(t[10])
        return t

    lprop.set_model_setup(rhs=frozen_background)
    try:
        out = _obj(state_from(d, "in"))
        for n in range(1, 11):
            out = lprop.RK3(float(d["dt"]), out)
            assert out.dtype == object and out.shape == (11,)
            if n in (1, 10):
                _check(out, d, f"s{n}")
        assert len(calls) == 30                              # three stages per step
        assert np.array_equal(out[9], d["in_uu"])            # the hook froze the column
    finally:
        lprop.set_model_setup(rhs=lprop.rhs_default)
        lprop.release_device()


def test_state_fed_back_stays_on_the_device_and_edits_are_seen(monkeypatch):
    """Lazy host copies: feeding the returned state back never downloads; an older state stays readable after the
    device has moved on (device snapshot); an in-place edit between two calls -- of a returned DeviceArray or of a
    frozen input array -- is uploaded, as the reference (which re-reads every slot on every call) would see it."""
    import msgwam_amd.libprop as lprop
    from msgwam_amd import _capi
    from oracle import msgwam_oracle as orc
    from helpers import setup_from
    d = load("g3_rk3_coupled_driver")
    _configure_from(d, lprop)
    dt = float(d["dt"])
    counts = {"rays": 0, "col": 0, "up": 0}
    for name, key in (("download_rays", "rays"), ("download_column", "col"), ("upload_rays", "up")):
        orig = getattr(_capi.Propagator, name)
        monkeypatch.setattr(_capi.Propagator, name,
                            lambda self, *a, _o=orig, _k=key, **kw: (counts.__setitem__(_k, counts[_k] + 1), _o(self, *a, **kw))[1])
    try:
        st = _obj(state_from(d, "in"))
        s1 = lprop.RK3(dt, st)
        assert isinstance(s1[3], lprop.DeviceArray)                        # evolving: on the device
        assert isinstance(s1[5], lprop.DeviceArray) and np.array_equal(s1[5], st[5])   # unchanged: a read-only copy,
        assert not np.asarray(s1[5]).flags.writeable and np.asarray(s1[5]) is not st[5]   # recognised in O(1) next time
        state = s1
        for _ in range(9):
            state = lprop.RK3(dt, state)
        assert counts == {"rays": 0, "col": 0, "up": 1}      # ten steps: one upload, nothing copied back
        _check(state, d, "s10")                              # first access: now it is copied
        assert counts["rays"] == 3 and counts["col"] == 2
        _check(s1, d, "s1")                                  # the state of nine steps ago, from its device snapshot
        # (1) edit a returned slot in place, (2) edit a frozen input array in place: both must take effect
        s = setup_from(d)
        state[0][7] = 0.0                                    # DeviceArray.__setitem__
        kk = state[5]
        kk[11] *= 1.5                                        # an unchanged slot, through the returned object
        with pytest.raises(ValueError):
            np.asarray(state[3])[0] = 1.0                    # the bare host copy is read-only: edits go through the object
        want = orc.rk3(s, dt, [np.asarray(a, dtype=np.float64) for a in state])
        got = lprop.RK3(dt, state)
        for i, k in enumerate(STATE_KEYS):
            g = np.asarray(got[i])
            if k in ("uu", "vv"):
                assert np.max(np.abs(g - want[i])) <= 1e-10 * np.max(np.abs(want[9])), k
            else:
                assert relerr(g, want[i]) <= 1e-10, k
        assert got[0][7] == 0.0 and counts["up"] == 2
        # (3) the caller's ORIGINAL arrays passed again after an in-place edit: seen by the content digest
        st2 = _obj(state_from(d, "in"))
        a1 = lprop.RK3(dt, st2)
        assert counts["up"] == 3
        st2[5][3] *= 2.0
        a2 = lprop.RK3(dt, st2)
        assert counts["up"] == 4 and not np.array_equal(np.asarray(a1[7]), np.asarray(a2[7]))
        # read-only input arrays are recognised by identity alone
        for i in range(11):
            st2[i].setflags(write=False)
        lprop.rhs_default(dt, st2)                           # (the device has moved on: this uploads st2 again)
        assert counts["up"] == 5
        lprop.rhs_default(dt, st2)                           # the same read-only objects: nothing to hash, nothing to upload
        assert counts["up"] == 5
        # plain ndarrays on request
        lprop.set_lazy_download(False)
        out = lprop.RK3(dt, got)
        assert all(type(a) is np.ndarray for a in out)
    finally:
        lprop.set_lazy_download(True)
        lprop.release_device()


def test_library_error_resets_what_is_assumed_resident(monkeypatch):
    """After an error of the HIP library (e.g. the persistent kernel's time-out) the mirror forgets what it assumed
    to be resident: the retry uploads again instead of failing forever on a state the library has dropped."""
    import msgwam_amd.libprop as lprop
    from msgwam_amd import _capi
    d = load("g3_rk3_coupled_driver")
    _configure_from(d, lprop)
    try:
        state = lprop.RK3(float(d["dt"]), _obj(state_from(d, "in")))
        orig = _capi.Propagator.step
        fail = {"n": 1}

        def flaky(self, *a, **kw):
            if fail["n"]:
                fail["n"] -= 1
                raise _capi.MsgwError("msgw_step: injected failure")
            return orig(self, *a, **kw)

        monkeypatch.setattr(_capi.Propagator, "step", flaky)
        with pytest.raises(_capi.MsgwError):
            lprop.RK3(float(d["dt"]), state)
        out = lprop.RK3(float(d["dt"]), state)               # uploads `state` again (copied from the device first)
        _check(out, d, "s2") if "s2_rr" in d else None
        assert np.all(np.isfinite(np.asarray(out[3])))
    finally:
        lprop.release_device()
