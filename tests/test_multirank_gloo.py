"""CPU, world_size 2, gloo: the multi-GPU decomposition (contiguous ray shards, replicated
column, all-reduce of the 2 x (ngrid-2) flux profile once per RK stage) reproduces the
single-process result.  The compute here is the ORACLE (tests may use it); what is under
test is the product's host-side sharding logic (msgwam_amd.sharding, spectrum shards,
unique-id exchange) and the decomposition itself, which is what msgw_comm_init +
ncclAllReduce implement on the GPUs."""
import copy
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from oracle import msgwam_oracle as orc
    from msgwam_amd.spectrum import gaussian_spectrum
    n = 4000
    grid = np.linspace(0, 100e3, 101)
    s0 = orc.Setup(grid, phi0=0.3)
    sp = gaussian_spectrum(n, s0.grids, s0.rhobar, alpha=0.05, nz=20, phi0=0.3)
    s = orc.Setup(grid, phi0=0.3, dkk=sp["dkk"], dll=sp["dll"], rr_mm_area=sp["area"])
    uu = orc.velocities_sine_homogeneous(s.grids, 4.0, 40e3, 10e3)
    vv = 0.2 * uu[::-1].copy()
    s.set_pressure_gradient(uu, vv)
    keys = ["dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm"]
    return s, [sp[k] for k in keys] + [uu, vv]


def _worker(rank, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "python-msgwam_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from oracle import msgwam_oracle as orc
    from msgwam_amd.sharding import shard_bounds, shard_state, shard_statics, exchange_unique_id
    s, st = _case()
    n = len(st[0])
    lo, hi = shard_bounds(n, WORLD, rank)
    local = shard_state(st, WORLD, rank)
    ls = copy.copy(s)
    sh = shard_statics(dict(dkk=s.dkk, dll=s.dll, rr_mm_area=s.rr_mm_area), n, WORLD, rank)
    ls.dkk, ls.dll, ls.rr_mm_area = sh["dkk"], sh["dll"], sh["rr_mm_area"]
    calls = [0]

    def allreduce(P):
        t = torch.from_numpy(np.ascontiguousarray(P))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        calls[0] += 1
        return t.numpy()

    for _ in range(3):
        local = orc.rk3(ls, 120.0, local, flux_reduce=allreduce)
    uid = exchange_unique_id(dist, rank, lambda: bytes(range(128)))
    q.put((rank, lo, hi, [np.asarray(a) for a in local], calls[0], uid))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_decomposition_matches_single_process():
    from oracle import msgwam_oracle as orc
    s, st = _case()
    want = st
    for _ in range(3):
        want = orc.rk3(s, 120.0, want)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == len(st[0])
    for r in res:
        assert r[4] == 9                       # one all-reduce per RK stage: 3 per step
        assert r[5] == bytes(range(128))       # the RCCL unique id reaches every rank
    for i in (0, 3, 7):                        # dens, rr, mm: shards concatenate to the full result
        got = np.concatenate([r[3][i] for r in res])
        assert np.max(np.abs(got - want[i]) / np.maximum(np.abs(want[i]), 1e-300)) <= 1e-12
    scale = np.max(np.abs(want[9]))
    for r in res:                              # the replicated column is identical on every rank
        assert np.max(np.abs(r[3][9] - want[9])) / scale <= 1e-12
        assert np.max(np.abs(r[3][10] - want[10])) / scale <= 1e-12
    np.testing.assert_array_equal(res[0][3][9], res[1][3][9])
    np.testing.assert_array_equal(res[0][3][10], res[1][3][10])
