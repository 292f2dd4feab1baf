"""-m gpu: the node-level in-kernel flux exchange of the persistent kernel (multi-GPU path).

The development box has ONE GPU, so the ranks of these tests share it: each rank is its own
process with its own HIP context, rays sharded, column replicated, exactly as on an 8-GPU node;
only the PCIe endpoints coincide.  RCCL refuses two ranks on one device, so the ranks are set up
with MSGW_EXCHANGE_ONLY=1 (no RCCL communicator; the shared segment is the only link)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_helpers import make_prop, gpu_state
from helpers import STATE_KEYS
from msgwam_amd import _capi
from test_gpu_parity import _random_case, check_state

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_exchange_with_one_rank_is_bitwise_the_single_gpu_path(monkeypatch):
    """1-rank communicator with the exchange forced on: the final row of every flux goes through the
    host-resident segment (store, release, sequence number, poll, load) and back.  0.0 + x == x, so
    the result must be BITWISE the plain persistent kernel's."""
    s, st = _random_case(200_000, 61, False, "uniform", True)
    ref = make_prop(s, st)
    ref.step(60.0, 2)
    ref.step(60.0, 7)
    assert ref.counters()["persist_steps"] == 7
    want = gpu_state(ref, st)
    ref.close()
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    p = make_prop(s, st)
    p.comm_init(_capi.comm_unique_id(), 0, 1)
    assert p.counters()["exchange"] == 1
    p.step(60.0, 2)
    p.step(60.0, 7)
    assert p.counters()["persist_steps"] == 7
    got = gpu_state(p, st)
    p.close()
    for k, a, b in zip(STATE_KEYS, got, want):
        assert np.array_equal(a, b, equal_nan=True), f"exchange path differs from the single-GPU path in {k}"


def test_exchange_off_falls_back_to_the_allreduce_chain(monkeypatch):
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    monkeypatch.setenv("MSGW_EXCHANGE", "0")
    s, st = _random_case(20_000, 62, False, "uniform", True)
    p = make_prop(s, st)
    p.comm_init(_capi.comm_unique_id(), 0, 1)
    assert p.counters()["exchange"] == 0
    p.step(60.0, 2)
    assert p.counters()["persist_steps"] == 0          # the RCCL launch chain ran
    p.close()


@pytest.mark.parametrize("nranks,n,sat,flags", [(2, 120_000, False, 0), (3, 50_001, False, 0),
                                                 (2, 90_001, True, 0), (2, 70_000, False, _capi.DIRECT_SAT)])
def test_ranks_as_processes_sharing_the_gpu(tmp_path, nranks, n, sat, flags):
    """nranks processes, each with its shard of the rays, advance together through the in-kernel
    exchange.  Against ONE process with all rays: per-ray state and column within summation-order
    noise (the ranks' partial sums are grouped differently), and the replicated columns of the ranks
    BITWISE equal to each other (every rank adds the same rows in the same order)."""
    s, st = _random_case(n, 70 + nranks, sat, "uniform", True)
    st[0] = st[0] * 1e-3                               # mild forcing: well-posed comparison
    calls = np.array([1, 2, 6])
    ref = make_prop(s, st)
    for k in calls:
        ref.step(60.0, int(k), flags)
    want = gpu_state(ref, st)
    ref.close()
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    case = tmp_path / "case.npz"
    np.savez(case, grid=s.grid, grids=s.grids, rhobar=s.rhobar, pg=s.pressure_gradient, uu=uu, vv=vv, dens=dens,
             rr=rr, drr=drr, kk=kk, ll=ll, mm=mm, dmm=dmm, phi=phi, dkk=np.broadcast_to(s.dkk, (n,)),
             dll=np.broadcast_to(s.dll, (n,)), area=np.broadcast_to(s.rr_mm_area, (n,)), bvf=s.bvf, phi0=s.phi0,
             kappa=s.kappa, sat=s.saturate_online, dt=60.0, calls=calls, flags=flags)
    uid = _capi.comm_unique_id().hex()
    env = dict(os.environ, MSGW_EXCHANGE_ONLY="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "xch_rank_worker.py"), str(case), str(r),
                               str(nranks), uid, str(tmp_path / f"out{r}.npz")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(nranks)]
    logs = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=240)
            logs.append(out.decode(errors="replace"))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()                              # exact children only
    assert all(pr.returncode == 0 for pr in procs), "\n".join(logs)
    outs = [np.load(tmp_path / f"out{r}.npz") for r in range(nranks)]
    for o in outs:
        assert int(o["exchange"]) == 1
        assert list(o["persist"]) == list(calls)       # every call was ONE persistent launch
        assert np.array_equal(o["uu"], outs[0]["uu"]) and np.array_equal(o["vv"], outs[0]["vv"])
    got = [np.asarray(x, dtype=np.float64).copy() for x in st]
    for i, k in ((0, "dens"), (3, "rr"), (7, "mm")):
        got[i] = np.concatenate([o[k] for o in outs])
    got[9], got[10] = outs[0]["uu"], outs[0]["vv"]
    check_state(got, want, 1e-10, 1e-11, f"{nranks} ranks vs one")


def test_bench_multi_rank_launch_rehearsal(tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one rank per process),
    rehearsed on ONE GPU: --share-gpu puts both ranks on GPU 0 with the exchange-only communicator and
    --backend gloo carries the unique id / barriers.  Checks the contract's JSON line of rank 0."""
    import json
    import socket
    with socket.socket() as sk:                        # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.join(HERE, "..")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--backend", "gloo", "--share-gpu", "--rays-per-gpu", "60000"]
    r = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]                # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["rays_total"] == 120000 and d["state_finite"] is True
    assert "inside the persistent kernel" in d["config"]["parallelism"]
    assert d["roofline"]["kernel"] == "k_rk3_persist" and d["value"] > 0
