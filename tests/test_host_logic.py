"""CPU-only: host-side logic of the product package (no GPU compute): module surface,
column/initial-condition helpers, spectrum generator and sharding."""
import numpy as np
import pytest

from oracle import msgwam_oracle as orc
from helpers import load


def test_module_surface_mirrors_the_reference():
    import msgwam_amd.libprop as lprop
    for name in ("HPROP_GLOBAL", "grid", "grids", "rhobar", "pressure_gradient", "model_config", "statics",
                 "ROT_EARTH", "RAD_EARTH", "set_statics", "set_model_setup", "get_model_setup",
                 "set_hydrostatics", "set_pressure_gradient", "velocities_sine_homogeneous", "omega",
                 "rhs_default", "RK3", "saturation", "wave_projection"):
        assert hasattr(lprop, name), name
    # defaults as at import of lib/libprop.py:703-726
    assert lprop.model_config["rhs"] is lprop.rhs_default
    assert lprop.model_config["kappa"] == 0.95 and lprop.model_config["saturate_online"] is True
    assert lprop.statics["rr_mm_area"] == 0


def test_column_helpers_match_the_reference_values():
    import msgwam_amd.libprop as lprop
    d = load("g3_rk3_coupled_f45")
    lprop.set_model_setup(bvf=0.01, boussinesq=False, sig_rr=10000, u0=4, rr0=40000, phi0=float(d["phi0"]),
                          hh=8500, rhobar0=1.2)
    grid = d["grid"]
    lprop.grid, lprop.grids = grid, .5 * (grid[:-1] + grid[1:])
    lprop.set_hydrostatics()
    np.testing.assert_array_equal(lprop.rhobar, d["rhobar"])
    uu = lprop.velocities_sine_homogeneous(lprop.grids)
    np.testing.assert_array_equal(uu, d["in_uu"])
    lprop.set_pressure_gradient(uu, d["in_vv"])
    np.testing.assert_array_equal(lprop.pressure_gradient, d["pg"])
    kk, ll, mm = d["in_kk"], d["in_ll"], d["in_mm"]
    np.testing.assert_array_equal(lprop.omega(kk, ll, mm, float(d["phi0"])),
                                  orc.omega(kk, ll, mm, float(d["phi0"]), 0.01))


def test_spectrum_is_deterministic_and_shardable():
    from msgwam_amd.spectrum import gaussian_spectrum
    from msgwam_amd.sharding import shard_bounds, shard_state
    s0 = orc.Setup(np.linspace(0, 100e3, 101))
    full = gaussian_spectrum(8000, s0.grids, s0.rhobar, nz=20)
    parts = []
    for r in range(4):
        lo, hi = shard_bounds(8000, 4, r)
        assert lo % 2 == 0
        parts.append(gaussian_spectrum(8000, s0.grids, s0.rhobar, nz=20, start=lo, stop=hi))
    for k in full:
        np.testing.assert_array_equal(np.concatenate([p[k] for p in parts]), full[k])
    # z-major, then azimuth, then m: consecutive rays share height and direction
    assert np.all(np.diff(full["rr"]) >= 0)
    assert np.all(full["kk"][:100] == full["kk"][0]) and np.all(np.diff(full["mm"][:100]) > 0)
    assert np.all(full["dens"] > 0) and np.all(np.isfinite(full["dens"]))
    st = [full[k] for k in ("dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm")] + [np.zeros(100)] * 2
    cover = [shard_state(st, 3, r)[3] for r in range(3)]
    np.testing.assert_array_equal(np.concatenate(cover), full["rr"])
    assert [shard_bounds(7, 3, r) for r in range(3)] == [(0, 2), (2, 4), (4, 7)]


def test_rk3_drives_a_foreign_rhs_hook_with_the_reference_rk_lines():
    """model_config['rhs'] may be any callable (lib/libprop.py:691).  A hook that never touches the GPU: linear
    tendencies, for which the Williamson scheme (lib/libprop.py:693-698) has a closed form."""
    import msgwam_amd.libprop as lprop
    rng = np.random.default_rng(3)
    st = np.empty(11, dtype=object)
    for i in range(11):
        st[i] = rng.normal(size=7 if i < 9 else 5)
    lam = np.linspace(-0.9, 0.4, 11)

    def rhs(dt, var):
        out = np.empty(11, dtype=object)
        for i in range(11):
            out[i] = lam[i] * var[i]
        return out

    old = lprop.model_config["rhs"]
    lprop.set_model_setup(rhs=rhs)
    try:
        dt = 0.3
        got = lprop.RK3(dt, st)
        for i in range(11):
            q = dt * (lam[i] * st[i]); y = st[i] + q / 3
            q = dt * (lam[i] * y) - 5 / 9 * q; y = y + 15 / 16 * q
            q = dt * (lam[i] * y) - 153 / 128 * q; y = y + 8 / 15 * q
            assert np.array_equal(got[i], y), i
            z = dt * lam[i]                                  # third-order: 1 + z + z^2/2 + z^3/6
            assert np.allclose(got[i], st[i] * (1 + z + z * z / 2 + z ** 3 / 6), rtol=1e-13)
    finally:
        lprop.set_model_setup(rhs=old)


def test_device_array_is_lazy_and_behaves_like_an_ndarray():
    """DeviceArray against a stub backend (no GPU): nothing is fetched until the values are needed; then it acts as
    the float64 ndarray it stands for; in-place edits are noticed."""
    import msgwam_amd.libprop as lprop

    class Stub:
        def __init__(self):
            self.fetched = 0

        def fetch(self, a):
            self.fetched += 1
            return np.arange(6, dtype=np.float64)

    b = Stub()
    a = lprop.DeviceArray(b, "rr", (6,), 0)
    assert a.shape == (6,) and len(a) == 6 and a.dtype == np.float64 and a.ndim == 1 and a.size == 6
    assert np.shape(a) == (6,) and "on device" in repr(a) and b.fetched == 0
    assert a._pristine()
    assert np.array_equal(a + 1, np.arange(1, 7)) and b.fetched == 1
    assert np.array_equal(np.asarray(a), np.arange(6)) and a[2] == 2.0 and float(np.sum(a)) == 15.0
    assert np.array_equal(np.concatenate([a, a]), np.tile(np.arange(6.0), 2)) and a.mean() == 2.5
    hist = np.zeros((2, 6))
    hist[1] = a                                              # the driver's `int_rr[nt] = state_out[3]`
    assert np.array_equal(hist[1], np.arange(6)) and b.fetched == 1
    assert a._pristine() and not np.asarray(a).flags.writeable   # the host copy is read-only ...
    with pytest.raises(ValueError):
        np.asarray(a)[0] = 5.0
    a[3] = -1.0                                              # ... until it is written through the object
    assert not a._pristine() and np.asarray(a)[3] == -1.0
    c = lprop.DeviceArray(b, "kk", (6,), 0, host=np.arange(6.0))   # a slot the step did not change: a private copy
    assert b.fetched == 1 and c._pristine() and c[2] == 2.0
    c += 1.0                                                 # in-place arithmetic goes through the object too
    assert not c._pristine() and c[2] == 3.0
    ro = np.arange(4.0)
    assert not lprop._frozen(ro)
    ro.setflags(write=False)
    assert lprop._frozen(ro) and not lprop._frozen(np.arange(4.0)[::2]) and lprop._slot_resident(ro, lprop._slot_key(ro))
    view = np.arange(4.0)[:2]
    view.setflags(write=False)
    assert not lprop._frozen(view)                           # a read-only view of a writable array is not trusted


def test_residency_fingerprints_see_in_place_edits():
    import msgwam_amd.libprop as lprop
    a = np.linspace(0, 1, 100_000)
    for mode in ("safe", "fast"):
        lprop.set_residency(mode)
        key = lprop._slot_key(a)
        assert lprop._slot_resident(a, key)
        assert not lprop._slot_resident(a.copy(), key)       # another object: not what was uploaded
        a[0] += 1.0                                          # both modes see an edit at the ends
        assert not lprop._slot_resident(a, key)
        a[0] -= 1.0
        assert lprop._slot_resident(a, key)
    lprop.set_residency("safe")
    key = lprop._slot_key(a)
    a[12_345] = 7.0                                          # one ray in the middle: the full digest sees it
    assert not lprop._slot_resident(a, key)
    lprop.set_residency("off")
    assert lprop._slot_key(a) is None and not lprop._slot_resident(a, key)
    lprop.set_residency("safe")


def test_digest_without_xxhash_is_position_sensitive(monkeypatch):
    """the fallback digest (xxhash not installed) must see a reordering -- a swap of two rays, an
    in-place sort -- not only a change of the multiset of values."""
    import msgwam_amd.libprop as lprop
    monkeypatch.setattr(lprop, "_digest", lprop._digest_stdlib)      # what the ImportError branch installs
    lprop.set_residency("safe")
    a = np.random.default_rng(5).normal(size=10_000)
    key = lprop._slot_key(a)
    assert lprop._slot_resident(a, key)
    a[[17, 4711]] = a[[4711, 17]]                                    # swap: same sum, same xor
    assert not lprop._slot_resident(a, key)
    a[[17, 4711]] = a[[4711, 17]]
    assert lprop._slot_resident(a, key)
    a.sort()                                                         # in-place reordering
    assert not lprop._slot_resident(a, key)
    e = np.zeros(0)
    assert lprop._slot_resident(e, lprop._slot_key(e))               # empty arrays digest too


def test_committed_counter_table_matches_the_kernel_sources():
    """bench.py's roofline reads HBM bytes and VALU instruction counts per kernel flavour from profiles/traffic.json
    (rocprofv3 PMC passes).  Every entry carries the digest of csrc/ it was measured at; a kernel change without a new
    profile run makes the bench line say `traffic_stale` -- and fails here, so that it cannot be committed unnoticed."""
    import json
    import os
    import bench
    table, err = bench.load_counter_table()
    assert table is not None, err
    digest = bench.kernel_src_digest()
    keys = [k for k in table if not k.startswith("_")]
    assert {"config3:f64:1000000:res4", "config3:f64:1000000:res0", "config2:f64:100000:res0:narrow",
            "config5:f64:1250000:res4".replace("f64", "f32")} <= set(keys)
    for k in keys:
        e = table[k]
        assert e.get("src_digest") == digest, (k, "profiles/traffic.json predates the current kernel sources: re-run "
                                                  "tools/profile_all.sh + tools/make_counter_table.py")
        assert os.path.exists(os.path.join(os.path.dirname(bench.__file__), e["source"]))
        assert e.get("valu_wave_insts_per_ray_step", 0) > 0
        assert ("bytes_per_ray_step" in e) != ("bytes_per_ray_launch" in e)


def test_checkpoint_reader_checks_the_format_before_touching_the_gpu(tmp_path):
    """`Propagator.load_checkpoint` reads a plain .npz (no pickle) and refuses another format version before it creates a
    context -- checkable without a GPU."""
    from msgwam_amd import _capi
    bad = str(tmp_path / "bad.npz")
    with open(bad, "wb") as f:
        np.savez(f, format=np.array(_capi.Propagator.CHECKPOINT_FORMAT + 1), dens=np.zeros(3), ngrid=np.array(101),
                 float32_state=np.array(False))
    with pytest.raises(_capi.MsgwError, match="checkpoint format"):
        _capi.Propagator.load_checkpoint(bad)
    pickled = str(tmp_path / "obj.npz")
    with open(pickled, "wb") as f:
        np.savez(f, format=np.array([{"a": 1}], dtype=object))
    with pytest.raises(ValueError, match="allow_pickle|pickled|Object arrays"):
        _capi.Propagator.load_checkpoint(pickled)
