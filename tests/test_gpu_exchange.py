"""-m gpu: the node-level in-kernel flux exchange of the persistent kernel (multi-GPU path).

The development box has ONE GPU, so the ranks of these tests share it: each rank is its own
process with its own HIP context, rays sharded, column replicated, exactly as on an 8-GPU node;
only the memory the peers' IPC mappings point at coincides (one HBM instead of eight linked by xGMI).
RCCL refuses two ranks on one device, so the ranks are set up with MSGW_EXCHANGE_ONLY=1 (no RCCL
communicator; the exchange buffers are the only link).  Both transports are exercised: the
device-resident one (default) and the host-segment fallback (MSGW_XCH_TRANSPORT=shm)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import msgwam_oracle as orc
from oracle.c_oracle import COracle
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
    assert p.counters()["exchange"] == 1 and p.counters()["transport"] == 3    # device-resident (HIP IPC) transport
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


def _run_ranks(tmp_path, case, nranks, env_extra=None, per_rank_env=None, timeout=240):
    """Start one worker process per rank on the shared GPU; returns (return codes, logs, outputs)."""
    uid = _capi.comm_unique_id().hex()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, MSGW_EXCHANGE_ONLY="1")
        env.update(env_extra or {})
        env.update((per_rank_env or {}).get(r, {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "xch_rank_worker.py"), str(case), str(r),
                                       str(nranks), uid, str(tmp_path / f"out{r}.npz")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=timeout)
            logs.append(out.decode(errors="replace"))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()                              # exact children only
    codes = [pr.returncode for pr in procs]
    outs = [np.load(tmp_path / f"out{r}.npz") if codes[r] == 0 else None for r in range(nranks)]
    return codes, logs, outs


def _save_case(path, s, st, n, calls, flags, dtype="f64", dt=60.0, **extra):
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = st
    np.savez(path, grid=s.grid, grids=s.grids, rhobar=s.rhobar, pg=s.pressure_gradient, uu=uu, vv=vv, dens=dens,
             rr=rr, drr=drr, kk=kk, ll=ll, mm=mm, dmm=dmm, phi=phi, dkk=np.broadcast_to(s.dkk, (n,)),
             dll=np.broadcast_to(s.dll, (n,)), area=np.broadcast_to(s.rr_mm_area, (n,)), bvf=s.bvf, phi0=s.phi0,
             kappa=s.kappa, sat=s.saturate_online, dt=dt, calls=calls, flags=flags, dtype=dtype, **extra)


@pytest.mark.parametrize("nranks,n,sat,flags,transport", [
    (2, 120_000, False, 0, "device_ipc"), (3, 50_001, False, 0, "device_ipc"), (2, 90_001, True, 0, "device_ipc"),
    (2, 70_000, False, _capi.DIRECT_SAT, "device_ipc"), (2, 120_000, False, 0, "host_shm"), (3, 50_001, True, 0, "host_shm")])
def test_ranks_as_processes_sharing_the_gpu(tmp_path, nranks, n, sat, flags, transport):
    """nranks processes, each with its shard of the rays, advance together through the in-kernel exchange, over
    the device-resident transport (every rank writes its row into the HIP-IPC-mapped buffers of all ranks) and
    over the host-segment fallback.  The concatenated shards are held to the C ORACLE run on the full ray set
    (rtol 1e-10: the ranks' partial sums are grouped differently from any single-process order), and the
    replicated columns of the ranks must be BITWISE equal to each other (every rank adds the same rows in rank
    order)."""
    s, st = _random_case(n, 70 + nranks, sat, "uniform", True)
    st[0] = st[0] * 1e-3                               # mild forcing: well-posed comparison
    calls = np.array([1, 2, 6])
    want = COracle(s).step(60.0, int(calls.sum()), st, direct_sat=2 if flags & _capi.DIRECT_SAT else 0)
    case = tmp_path / "case.npz"
    _save_case(case, s, st, n, calls, flags)
    codes, logs, outs = _run_ranks(tmp_path, case, nranks,
                                   {"MSGW_XCH_TRANSPORT": "shm"} if transport == "host_shm" else None)
    assert all(c == 0 for c in codes), "\n".join(logs)
    for o in outs:
        assert int(o["exchange"]) == 1 and _capi.TRANSPORTS[int(o["transport"])] == transport
        assert int(o["tenants"]) == nranks                 # the ranks found out that they share one device
        assert list(o["persist"]) == list(calls)       # every call was ONE persistent launch
        assert np.array_equal(o["uu"], outs[0]["uu"]) and np.array_equal(o["vv"], outs[0]["vv"])
    got = [np.asarray(x, dtype=np.float64).copy() for x in st]
    for i, k in ((0, "dens"), (3, "rr"), (7, "mm")):
        got[i] = np.concatenate([o[k] for o in outs])
    got[9], got[10] = outs[0]["uu"], outs[0]["vv"]
    check_state(got, want, 1e-10, 1e-11, f"{nranks} ranks vs the oracle")


def test_two_ranks_float32_saturation_and_relaunch_vs_oracle(tmp_path):
    """BASELINE config 5's kernel variant (float32 state, online saturation, relaunch extension) on two ranks that
    exchange their flux rows inside the persistent kernel, against the float64 oracle's step -> relaunch loop on
    the full ray set (float32 tolerances, see test_gpu_f32.py)."""
    from test_gpu_f32 import _spectrum_case, check32
    n = 200_000
    s, st = _spectrum_case(n, 0.01, True, kappa=0.008)
    src = (st[0].copy(), st[3].copy(), st[7].copy())
    co = COracle(s)
    want, recycled = st, 0
    for _ in range(3):
        want = co.step(120.0, 1, want)
        want, mask = orc.relaunch(s, want, src, 0.5)
        recycled += int(mask.sum())
    assert recycled > 1000
    case = tmp_path / "case.npz"
    _save_case(case, s, st, n, np.array([1, 2]), _capi.RELAUNCH, dtype="f32", dt=120.0, relaunch_frac=0.5)
    codes, logs, outs = _run_ranks(tmp_path, case, 2)
    assert all(c == 0 for c in codes), "\n".join(logs)
    assert all(list(o["persist"]) == [1, 2] for o in outs)
    assert np.array_equal(outs[0]["uu"], outs[1]["uu"])
    got = [np.asarray(x, dtype=np.float64).copy() for x in st]
    for i, k in ((0, "dens"), (3, "rr"), (7, "mm")):
        got[i] = np.concatenate([o[k] for o in outs])
    got[9], got[10] = outs[0]["uu"], outs[0]["vv"]
    check32(got, want, 1e-4, 1e-4, "2 ranks f32", outliers=1e-3)


@pytest.mark.parametrize("per_rank_env", [{0: {"MSGW_PERSIST": "0"}, 1: {"MSGW_PERSIST": "0"}}, {1: {"MSGW_PERSIST": "0"}}])
def test_ranks_without_rccl_fail_loudly_when_a_step_leaves_the_persistent_kernel(tmp_path, per_rank_env):
    """A communicator without RCCL (MSGW_EXCHANGE_ONLY=1) can only sum the ranks' rows inside the persistent
    kernel.  When a step cannot take it -- here MSGW_PERSIST=0 on both ranks, or on ONE rank only (the ranks agree
    on the path through the segment before anything is launched, so the other rank does not spin into its
    time-out) -- every rank must raise instead of advancing its column with its local flux."""
    import time
    s, st = _random_case(40_000, 75, False, "uniform", True)
    case = tmp_path / "case.npz"
    _save_case(case, s, st, 40_000, np.array([2]), 0)
    t0 = time.perf_counter()
    codes, logs, _ = _run_ranks(tmp_path, case, 2, per_rank_env=per_rank_env, timeout=120)
    assert time.perf_counter() - t0 < 60                   # no rank waited for an exchange time-out (20 s per wait)
    assert all(c != 0 for c in codes), "\n".join(logs)
    assert all("without RCCL" in log for log in logs), "\n".join(logs)


@pytest.mark.parametrize("transport", ["device_ipc", "host_shm"])
def test_config4_shard_size_through_the_exchange_vs_c_oracle(monkeypatch, transport):
    """BASELINE config 4's per-GPU shard (1.25e6 rays of the synthetic spectrum, float64, coupled) through the
    multi-rank code path with a 1-rank communicator (MSGW_FORCE_COLLECTIVE=1: exchange workgroup, sequence
    numbers, row through the transport and back) against the C oracle on every ray, 2 steps."""
    from test_gpu_f32 import _spectrum_case
    monkeypatch.setenv("MSGW_FORCE_COLLECTIVE", "1")
    if transport == "host_shm":
        monkeypatch.setenv("MSGW_XCH_TRANSPORT", "shm")
    n = 1_250_000
    s, st = _spectrum_case(n, 0.01, False)
    want = COracle(s).step(120.0, 2, st)
    p = make_prop(s, st)
    p.comm_init(_capi.comm_unique_id(), 0, 1)
    c = p.counters()
    assert c["exchange"] == 1 and _capi.TRANSPORTS[c["transport"]] == transport
    p.step(120.0, 2)
    assert p.counters()["persist_steps"] == 2
    got = gpu_state(p, st)
    p.close()
    check_state(got, want, 1e-10, 1e-11, "config4 shard")
    assert not np.array_equal(got[9], st[9])


def _rehearse_bench(extra, size=("--rays-per-gpu", "60000")):
    import json
    import socket
    with socket.socket() as sk:                        # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.join(HERE, "..")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--backend", "gloo", "--share-gpu"] + list(size) + extra
    r = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]                # rank 0 only
    return json.loads(lines[0])


def test_bench_strong_scaling_rehearsal():
    """"also run at 1, 2, 4 GPUs with the same total N".  `--total-rays` divides one ray set over
    the ranks (the line then says "scaling": "strong"); with N > 1 and no --workload the bench runs config 4."""
    d = _rehearse_bench(["--total-rays", "100400"], size=())          # (the spectrum tiles 100 x 4 bins: multiples of 400)
    assert d["scaling"] == "strong" and d["n_gpus"] == 2
    assert d["config"]["rays_total"] == 100400 and d["config"]["rays_per_gpu"] == 50200
    assert d["config"]["workload"].startswith("config4") and d["dtype"] == "f64"
    assert d["config"]["launch_ray_workgroups"] > 0 and d["config"]["launch_workgroups"] > d["config"]["launch_ray_workgroups"]
    assert d["state_finite"] is True and d["value"] > 0


def test_bench_falls_back_to_the_next_transport_when_the_exchange_fails_at_run_time():
    """A rank that reports a failure of the device-resident transport inside its first timed region: every rank learns
    of it at the same fence (nobody is left in a collective), the measurement is repeated through the host
    shared-memory transport, and the JSON line says so."""
    d = _rehearse_bench(["--inject-exchange-failure", "device_ipc"])
    assert d["config"]["transport"] == "host_shm" and d["n_gpus"] == 2 and d["value"] > 0
    assert len(d["config"]["transport_fallbacks"]) == 1        # (rank 0's view: "another rank reported a failure")
    assert d["state_finite"] is True and d["roofline"]["kernel"] == "k_rk3_persist"


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


def test_bench_checks_the_exchange_against_the_rccl_chain_before_timing():
    """bench.py, whenever it times a multi-rank path with an RCCL communicator at hand: ONE step through the in-kernel
    exchange against the same step through the all-reduce chain first (the device-resident transport has never run on
    distinct GPUs: the first 8-GPU run must not report throughput of a wrong exchange).  Rehearsed with a 1-rank
    communicator."""
    import json
    root = os.path.join(HERE, "..")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--steps", "4", "--warmup", "2",
           "--rays-per-gpu", "100000", "--no-cpu-baseline", "--no-size-sweep", "--no-streamed-leg"]
    r = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    d = json.loads([l for l in out.splitlines() if l.startswith("{") and '"metric"' in l][-1])
    chk = d["config"]["exchange_check"]
    assert chk["ok"] is True and chk["transports"] == ["device_ipc", "rccl"], chk
    assert chk["column_max_diff_over_scale"] <= 1e-12 and chk["ray_max_rel_diff"] <= 1e-12
    assert d["config"]["transport"] == "device_ipc" and d["config"]["persist_steps"] == 4
