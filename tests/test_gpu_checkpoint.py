"""-m gpu: checkpoint / resume of the device-resident state (Propagator.save_checkpoint / load_checkpoint; SURVEY.md 5
lists the .npz snapshot as a "next" item -- the reference keeps its history in RAM, raytracer.py:125-136, and has no
counterpart).  A run that is saved, destroyed and resumed must continue exactly as the uninterrupted one."""
import numpy as np
import pytest

from msgwam_amd import _capi
from test_gpu_parity import _random_case
from test_gpu_chain import _case, make_chain_prop, chain_state

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y, equal_nan=True), (what, i)


@pytest.mark.parametrize("dtype,hprop,nz,sat,flags", [
    ("f64", False, False, False, _capi.DIRECT_SAT_QUIRK | _capi.RELAUNCH),   # the tuned kernels (persistent form)
    ("f32", False, False, True, _capi.RELAUNCH),                             # config 5's flavour
    ("f64", True, True, True, 0),                                            # the general chain, every slot evolving
    ("f64", True, False, False, _capi.DIRECT_SAT | _capi.RELAUNCH),
])
def test_resume_continues_bit_for_bit(tmp_path, monkeypatch, dtype, hprop, nz, sat, flags):
    """5 steps, checkpoint, 7 more -- against: load the checkpoint into a NEW context, 7 steps.  Bitwise equal (the
    carried flux is switched off: a resumed run starts with the deposit pre-pass, whose summation order differs from the
    lagged deposit's in the last bits, tests/test_gpu_parity.py::test_flux_carried_between_calls)."""
    monkeypatch.setenv("MSGW_CARRY", "0")
    s, st, col = _case(30_011, 500 + 2 * hprop + nz, sat, hprop, nz)
    if not sat:
        st[0] = st[0] * 1e-3
    p = make_chain_prop(s, st, hprop, col, dtype=dtype)
    p.set_relaunch(1e-5)
    p.step(60.0, 5, flags)
    path = str(tmp_path / "state.npz")
    p.save_checkpoint(path, step=5, dt=60.0, note=np.array([1.5, 2.5]))
    p.step(60.0, 7, flags)
    want = chain_state(p, st, hprop, nz)
    p.close()
    with np.load(path, allow_pickle=False) as z:              # a plain numpy container, nothing to unpickle
        assert int(z["format"]) == 1 and int(z["meta_step"]) == 5 and len(z["dens"]) == 30_011
    q, meta = _capi.Propagator.load_checkpoint(path)
    assert int(meta["step"]) == 5 and float(meta["dt"]) == 60.0 and np.array_equal(meta["note"], [1.5, 2.5])
    assert q.dtype == dtype and q.n == 30_011
    q.step(60.0, 7, flags)
    got = chain_state(q, st, hprop, nz)
    q.close()
    _same(got, want, (dtype, hprop, nz))
    # the relaunch source survived the round trip: it is the state of the ORIGINAL upload, not the checkpointed one
    if flags & _capi.RELAUNCH:
        with np.load(path, allow_pickle=False) as z:
            assert np.array_equal(z["src_rr"], st[3]) and not np.array_equal(z["rr"], st[3])


def test_checkpoint_needs_a_complete_context_and_rejects_other_formats(tmp_path):
    p = _capi.Propagator(101, 10)
    with pytest.raises(_capi.MsgwError, match="come first"):
        p.save_checkpoint(str(tmp_path / "x.npz"))
    p.close()
    bad = str(tmp_path / "bad.npz")
    with open(bad, "wb") as f:
        np.savez(f, format=np.array(99), dens=np.zeros(3), ngrid=np.array(101), float32_state=np.array(False))
    with pytest.raises(_capi.MsgwError, match="format 99"):
        _capi.Propagator.load_checkpoint(bad)


def test_driver_run_resumes_from_its_checkpoint(tmp_path):
    """The headless driver: 40 steps in one go against 25 steps + checkpoint, then resume to 40."""
    from msgwam_amd import driver
    path = str(tmp_path / "run.npz")
    full = driver.run(nray=60, nt_max=40, diagnostics=False)
    driver.run(nray=60, nt_max=25, diagnostics=False, checkpoint_path=path, checkpoint_every=10)
    part = driver.run(nray=60, nt_max=40, diagnostics=False, resume_from=path)
    assert part["stored"][0] == 26 and part["stored"][-1] == 40
    for k in ("int_dens", "int_rr", "int_mm", "int_uu"):
        assert np.allclose(part[k][26:], full[k][26:], rtol=1e-11, atol=0), k
