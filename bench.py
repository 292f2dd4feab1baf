#!/usr/bin/env python3
"""
bench.py -- ray-steps/s of the MI355X ray-propagation path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3|config4|config2|config5] [--repeats R]
                    [--rays-per-gpu M | --total-rays T]

A "step" is one lprop.RK3 step (3 RHS stages incl. flux deposit and mean-flow update) of every resident ray.
Workloads (BASELINE.json configs; synthetic Gaussian source spectrum of SURVEY 8d, inputs resident in HBM before the
timed region):

  config3 (default at N = 1)  1e6 rays per GPU, interactive mean flow, float64            280 B per ray-step
  config4 (default at N > 1)  1.25e6 rays per GPU (1e7 rays on 8 GPUs), otherwise config3  280 B per ray-step
  config2            1e5 rays per GPU, fixed background (pure propagation), float64       48 B per ray-step
  config5            1.25e6 rays per GPU, float32 state, online saturation + the
                     relaunch extension (alpha = 0.5), interactive mean flow              180 B per ray-step

For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds `--rays-per-gpu` rays (weak
scaling, the default: N = 8 with 1.25e6 rays per GPU is config 4 / config 5) or `--total-rays / N` rays (strong
scaling: SURVEY 8d's "1, 2, 4 GPUs with the same total N" table; the line then says "scaling": "strong"), and the
2 x (ngrid-2) flux profile is summed over the ranks once per RK stage inside the C library (`config.parallelism` names
the transport).

Timing: R repeats of the same experiment -- fresh initial state, W untimed warm-up steps, then EXACTLY K steps timed,
bracketed by a barrier + synchronisation on both sides, MAX over the ranks; `value` and `ms_per_step` are the MEDIAN
repeat (min / max are reported beside it).  R defaults to what makes the timed region >= 50 ms (at least 5).  Every
repeat covers the same steps W+1 .. W+K, so the repeats are comparable (step time depends on how far the packet has
dispersed: see `late_time`).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-msgwam_amd"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# VALU issue peak: 1024 SIMDs, one wave64 VALU instruction per 4 clocks of the 2.4 GHz maximum clock (float64 FMA:
# 78.6 TF = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz, i.e. 16 lanes per clock per SIMD), in 1e9 wave-instructions/s
VALU_ISSUE_PEAK_GINST = 1024 * 2.4 / 4
DT = 120.0                      # raytracer.py:46

WORKLOADS = {
    # algorithmic bytes per ray per RK3 step: SURVEY 8d / BASELINE.md section 4
    "config3": dict(bytes=280.0, dtype="f64", rays=1_000_000, alpha=0.01, sat=False, flags=0, kernel="k_rk3_persist",
                    text="config3: 1e6 rays/GPU, interactive mean flow (flux deposit + u,v update every RK stage), "
                         "synthetic Gaussian spectrum, fp64"),
    "config4": dict(bytes=280.0, dtype="f64", rays=1_250_000, alpha=0.01, sat=False, flags=0, kernel="k_rk3_persist",
                    text="config4: 1.25e6 rays/GPU (1e7 rays over 8 GPUs), interactive mean flow, flux profile summed over "
                         "the ranks once per RK stage, synthetic Gaussian spectrum, fp64"),
    "config2": dict(bytes=48.0, dtype="f64", rays=100_000, alpha=0.01, sat=False, flags="fixed", kernel="k_ray_step_fixed",
                    text="config2: 1e5 rays/GPU, fixed background, pure propagation, fp64"),
    "config5": dict(bytes=180.0, dtype="f32", rays=1_250_000, alpha=0.5, sat=True, flags="relaunch", kernel="k_rk3_persist",
                    text="config5: 1.25e6 rays/GPU, fp32 ray state, online saturation + source relaunch, interactive "
                         "mean flow, synthetic Gaussian spectrum (alpha = 0.5)"),
}
ALIASES = {"coupled": "config3", "fixed": "config2"}


def column(ngrid=101, grid_max=100e3):
    """Driver column (raytracer.py:36-37, :53-64, :74-99): host-side numpy setup."""
    from msgwam_amd import libprop as lprop
    lprop.HPROP_GLOBAL = False
    lprop.set_model_setup(bvf=0.01, rhs=lprop.rhs_default, boussinesq=False, sig_rr=10000, u0=4,
                          rr0=40000, rr1=40000, phi0=0.0, kappa=1., saturate_online=False,
                          hh=8500, rhobar0=1.2)
    grid = np.linspace(0, grid_max, ngrid)
    grids = .5 * (grid[:-1] + grid[1:])
    lprop.grid, lprop.grids = grid, grids
    uu = lprop.velocities_sine_homogeneous(grids)
    vv = np.zeros(uu.shape)
    lprop.set_hydrostatics()
    lprop.set_pressure_gradient(uu, vv)
    return lprop, grid, grids, uu, vv


def cpu_baseline(wl, grid, uu, vv, budget_s=12.0):
    """Time the CPU restatement of the reference path on this box's host cores (rank 0, N=1 only).  kind "port": the
    reference is pure Python and never travels; what is timed is oracle/msgwam_oracle.py in its `loop=True` mode,
    i.e. the reference's algorithmic structure incl. the interpreted per-ray deposit loop
    (lib/libprop.py:151-163), single thread, float64 (the reference has no float32 mode; config5's relaunch is the
    oracle's definition of the extension)."""
    from oracle import msgwam_oracle as orc
    from oracle.c_oracle import COracle
    from msgwam_amd.spectrum import gaussian_spectrum
    n = 4000
    w = WORKLOADS[wl]
    s0 = orc.Setup(grid)
    sp = gaussian_spectrum(n, s0.grids, s0.rhobar, alpha=w["alpha"], nz=100, nd=4)
    s = orc.Setup(grid, saturate_online=w["sat"], dkk=sp["dkk"], dll=sp["dll"], rr_mm_area=sp["area"])
    s.set_pressure_gradient(uu, vv)
    keys = ["dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm"]
    st0 = [sp[k] for k in keys] + [uu, vv]
    src = (sp["dens"].copy(), sp["rr"].copy(), sp["mm"].copy())
    fixed = w["flags"] == "fixed"
    rl = w["flags"] == "relaunch"

    def post(st):
        return orc.relaunch(s, st, src, 1e-6)[0] if rl else st

    def run(fn, min_steps=1):
        st, steps, t0 = st0, 0, time.perf_counter()
        while steps < min_steps or time.perf_counter() - t0 < budget_s / 3:
            st = post(fn(st))
            steps += 1
        return n * steps / (time.perf_counter() - t0), steps

    loop_rate, loop_steps = run(lambda st: orc.rk3(s, DT, st, loop=True, fixed_background=fixed))
    vec_rate, _ = run(lambda st: orc.rk3(s, DT, st, loop=False, fixed_background=fixed))
    co = COracle(s, fixed_background=fixed)
    c_rate, _ = run(lambda st: co.step(DT, 1, st), min_steps=3)
    return {"value": loop_rate, "unit": "ray-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} rays x {loop_steps} RK3 steps, synthetic Gaussian spectrum, ngrid 101, "
                      f"oracle/msgwam_oracle.py loop=True (reference's per-ray Python deposit loop), float64",
            "host_cores": os.cpu_count(),
            "vectorised_numpy_value": vec_rate, "c_port_value": c_rate}


def coupled_workload(w):
    return w["flags"] != "fixed"


def kernel_src_digest():
    """Digest of the device code's sources (every header, tile_body.inc and the kern_*.hip translation units that
    instantiate the kernels; not msgwam_hip.hip, the host side of the C ABI, which holds no device code):
    profiles/traffic.json entries carry the digest they were measured at, so a line can say when its committed counter
    figures predate a kernel change (`traffic_stale`).  The launch geometry, which the host side chooses, is part of
    every entry's key."""
    import glob
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    d = os.path.join(ROOT, "python-msgwam_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "kern_*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def load_counter_table():
    """profiles/traffic.json: per kernel flavour the HBM bytes (FETCH_SIZE doubled + WRITE_SIZE) and the wave-level
    VALU instructions (SQ_INSTS_VALU) per ray-step from committed rocprofv3 --pmc passes (tools/make_counter_table.py)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))), None
    except Exception as e:      # noqa: BLE001
        return None, str(e)


def copy_ceiling(torch, nbytes=1 << 30, reps=10):
    """Measured streaming ceiling of this GPU: device-to-device copy of a buffer far larger than the 256 MiB
    Infinity Cache, read + written bytes over the HIP-event time (SURVEY 8d asks for it beside the 8 TB/s spec)."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    a.fill_(1.0)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed RK3 steps (default 200; config2: 1000, SURVEY 8d \">= 1000 steps timed\")")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of K steps (0 = until the timed region is >= 50 ms, at least 5)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + sorted(ALIASES), default=None,
                    help="default: config3 at N = 1 (the configuration BASELINE's metric is quoted on), config4 at N > 1")
    ap.add_argument("--rays-per-gpu", type=int, default=0, help="0 = the workload's BASELINE size (weak scaling)")
    ap.add_argument("--total-rays", type=int, default=0,
                    help="strong scaling: this many rays in total, divided evenly over the N ranks (SURVEY 8d: the "
                         "same-total-N table of config 4)")
    ap.add_argument("--no-exchange-check", action="store_true",
                    help="N>1: skip the one-step comparison of the in-kernel exchange with the RCCL chain before timing")
    ap.add_argument("--no-streamed-leg", action="store_true",
                    help="skip the all-rays-streamed measurement (MSGW_REGTILES=0) behind roofline.hbm_streamed")
    ap.add_argument("--ngrid", type=int, default=101)
    ap.add_argument("--blocks-per-cu", type=int, default=int(os.environ.get("MSGW_BLOCKS_PER_CU", 4)))
    ap.add_argument("--graph-steps", type=int, default=int(os.environ.get("MSGW_GRAPH_STEPS", 4)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-size-sweep", action="store_true", help="skip the extra 4x-rays and late-time measurements (N=1 only)")
    ap.add_argument("--late-steps", type=int, default=600, help="pre-steps before the late-time measurement")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 only: run the multi-GPU path with a 1-rank communicator (diagnostic; "
                         "MSGW_EXCHANGE=0 selects the RCCL launch chain instead of the in-kernel exchange)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N>1 launch (only carries the unique id, barriers and the "
                         "max-over-ranks time; the flux exchange is the library's own)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="diagnostic: all ranks use GPU 0 (rehearsal of the N>1 path on a 1-GPU box; implies the "
                         "exchange-only communicator because RCCL refuses two ranks on one device; use --backend gloo)")
    ap.add_argument("--inject-exchange-failure", choices=["device_ipc", "host_shm"], default=None,
                    help="diagnostic (N>1): the last rank reports a failure of this transport in its first timed region, "
                         "to exercise the fall-back ladder device_ipc -> host_shm -> rccl")
    ap.add_argument("--kernel-events", choices=["separate", "same", "none"], default="same",
                    help="where the per-launch HIP-event timing of the dominant kernel is taken")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    wl = args.workload or ("config3" if max(world, args.gpus) == 1 else "config4")
    wl = ALIASES.get(wl, wl)
    W = WORKLOADS[wl]
    if args.total_rays and args.rays_per_gpu:
        raise SystemExit("--total-rays and --rays-per-gpu exclude each other")
    strong = args.total_rays > 0
    if (args.total_rays or 0) % 400 or (args.rays_per_gpu or 0) % 400:
        raise SystemExit("ray counts must be multiples of 400 (the synthetic spectrum tiles 100 height x 4 azimuth bins)")
    rays_per_gpu = args.rays_per_gpu or W["rays"]
    if args.steps <= 0:
        args.steps = 1000 if wl == "config2" else 200

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    import torch
    from msgwam_amd import _capi
    from msgwam_amd.spectrum import gaussian_spectrum
    from msgwam_amd.sharding import shard_bounds

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
        os.environ["MSGW_EXCHANGE_ONLY"] = "1"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    lprop, grid, grids, uu, vv = column(args.ngrid)
    flags = {0: 0, "fixed": _capi.FIXED_BACKGROUND, "relaunch": _capi.RELAUNCH}[W["flags"]]
    same = args.kernel_events == "same"
    if args.force_collective and world == 1:
        os.environ["MSGW_FORCE_COLLECTIVE"] = "1"

    def fresh_uid():
        """A communicator id of its own for every context that joins one (all ranks call this alike)."""
        if world > 1:
            box = [_capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return _capi.comm_unique_id() if args.force_collective else None

    class ExchangeFailed(RuntimeError):
        """Some rank's library call failed inside a multi-rank measurement; raised on EVERY rank at the same fence."""

    def all_flag(flag):
        """max over the ranks of a 0/1 flag (the fence's barrier)"""
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0.5)

    def measure(rays, steps, warmup, kernel_events, repeats, pre_steps=0, total=0):
        """`repeats` timed regions of `steps` RK3 steps of `rays` rays per rank (or `total` rays over all ranks),
        state resident."""
        n_total = total or rays * world
        lo, hi = shard_bounds(n_total, world, rank)
        sp = gaussian_spectrum(n_total, grids, lprop.rhobar, alpha=W["alpha"], start=lo, stop=hi)
        n_local = hi - lo
        p = _capi.Propagator(args.ngrid, n_local, device=local_rank, dtype=W["dtype"])
        p.set_config(0.01, 0.0, 1.0, W["sat"])
        p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
        p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"],
                      sp["dkk"], sp["dll"], sp["area"])
        p.set_tuning(args.blocks_per_cu, args.graph_steps)
        # Several ranks: a library error on one rank (a time-out in the flux exchange, say) must not leave the others
        # in a collective.  Library calls are guarded; the error is agreed on at the next fence and raised everywhere.
        err = [None]

        def safe(fn, *a):
            if err[0] is None:
                try:
                    return fn(*a)
                except _capi.MsgwError as e:
                    if dist is None:
                        raise
                    err[0] = str(e)
            return None

        uid = fresh_uid()
        if uid is not None:
            safe(p.comm_init, uid, rank, world)
            if dist is not None and all_flag(err[0] is not None):
                try:
                    p.close()
                except Exception:      # noqa: BLE001
                    pass
                raise ExchangeFailed(err[0] or "another rank could not set up its communicator")
        if args.inject_exchange_failure and dist is not None and rank == world - 1 and \
                _capi.TRANSPORTS.get(p.counters().get("transport", 0)) == args.inject_exchange_failure:
            injected = f"injected failure of the {args.inject_exchange_failure} transport"   # reported at the next fence
        else:
            injected = None

        def fence():
            safe(p.sync)
            torch.cuda.synchronize()
            if dist is not None and all_flag(err[0] is not None or injected is not None):
                try:
                    p.close()
                except Exception:      # noqa: BLE001
                    pass
                raise ExchangeFailed(err[0] or injected or "another rank reported a failure")

        def reset():
            """every repeat times the SAME steps: fresh state, `pre_steps` (late-time measurement only), warm-up"""
            safe(p.set_column, grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
            safe(p.upload_rays, sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"],
                 sp["dkk"], sp["dll"], sp["area"])
            done = 0
            while done < pre_steps:                        # late-time measurement: let the packet spread first
                k = min(200, pre_steps - done)
                safe(p.step, DT, k, flags)
                done += k
            safe(p.step, DT, warmup, flags)

        walls, kern = [], []
        target_s, r = 0.05, 0
        while True:
            reset()
            fence()
            c_prev = p.counters()
            t0 = time.perf_counter()
            safe(p.step, DT, steps, flags | (_capi.TIME_KERNELS if kernel_events == "same" else 0))
            fence()
            wall = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([wall], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall = float(t.item())
            walls.append(wall)
            c1 = p.counters()
            if kernel_events == "same":
                nl = c1["ray_kernel_launches"] - c_prev["ray_kernel_launches"]
                kern.append(((c1["ray_kernel_ms_sum"] - c_prev["ray_kernel_ms_sum"]) / max(nl, 1), nl))
            c_prev = c1
            r += 1
            if repeats > 0:
                stop = r >= repeats
            else:
                stop = r >= 5 and sum(walls) >= target_s
                if dist is not None:                       # all ranks must agree on when to stop
                    t = torch.tensor([1.0 if stop else 0.0], dtype=torch.float64,
                                     device="cuda" if args.backend == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    stop = bool(t.item() > 0.5)
            if stop or r >= 50:
                break
        if kernel_events == "separate":                    # HIP events on the library's own stream, second pass
            safe(p.step, DT, steps, flags | _capi.TIME_KERNELS)
            fence()
            c1 = p.counters()
            nl = c1["ray_kernel_launches"] - c_prev["ray_kernel_launches"]
            kern.append(((c1["ray_kernel_ms_sum"] - c_prev["ray_kernel_ms_sum"]) / max(nl, 1), nl))
        kern_ms = float(np.median([k[0] for k in kern])) if kern else None
        launches = kern[0][1] if kern else 0
        finite = bool(np.all(np.isfinite(p.download_rays()[1])))
        c1 = p.counters()
        p.close()
        return dict(n_total=n_total, n_local=n_local, walls=walls, kern_ms=kern_ms, launches=launches,
                    counters=c1, finite=finite, steps=steps)

    # Several ranks: if the flux exchange fails at run time (it has passed its self-test when the communicator was set
    # up), every rank learns of it at the same fence and the measurement is repeated one rung down the ladder
    # device-resident (HIP IPC) -> host shared memory -> RCCL launch chain; the line then says which transport ran.
    fell_back = []
    ladder = [None, {"MSGW_XCH_TRANSPORT": "shm"}, {"MSGW_EXCHANGE": "0"}]
    if os.environ.get("MSGW_EXCHANGE_ONLY"):
        ladder = ladder[:2]                                    # no RCCL communicator to fall back on

    def one_step_state(env):
        """One RK3 step of this rank's shard through the transport `env` selects; (rays, column) or an ExchangeFailed."""
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            n_total = args.total_rays or rays_per_gpu * world
            lo, hi = shard_bounds(n_total, world, rank)
            sp = gaussian_spectrum(n_total, grids, lprop.rhobar, alpha=W["alpha"], start=lo, stop=hi)
            p = _capi.Propagator(args.ngrid, hi - lo, device=local_rank, dtype=W["dtype"])
            err = None
            try:
                p.set_config(0.01, 0.0, 1.0, W["sat"])
                p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
                p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"],
                              sp["dkk"], sp["dll"], sp["area"])
                p.comm_init(fresh_uid(), rank, world)
                tr = _capi.TRANSPORTS.get(p.counters().get("transport", 0))
                p.step(DT, 1, flags)
                rays, col = p.download_rays(), p.download_column()
            except _capi.MsgwError as e:
                err, rays, col, tr = str(e), None, None, None
            try:
                p.close()
            except Exception:      # noqa: BLE001
                pass
            if dist is not None and all_flag(err is not None):
                raise ExchangeFailed(err or "another rank reported a failure")
            if err is not None:
                raise ExchangeFailed(err)
            return rays, col, tr
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    # Before any multi-rank timing: ONE step through the in-kernel exchange against the same step through the RCCL
    # all-reduce chain, from the same state (the device-resident transport has only ever run with rank
    # processes sharing one GPU).  Same per-ray arithmetic, flux rows summed in a different order: 1e-12 is generous.  A
    # mismatch takes the in-kernel exchange out of the ladder; the line says so either way (`config.exchange_check`).
    exchange_check = None
    if (world > 1 or args.force_collective) and coupled_workload(W) and not os.environ.get("MSGW_EXCHANGE_ONLY") \
            and os.environ.get("MSGW_EXCHANGE", "1") != "0" and not args.no_exchange_check:
        try:
            (r1, c1_, t1), (r2, c2_, t2) = one_step_state(None), one_step_state({"MSGW_EXCHANGE": "0"})
            scale = max(float(np.max(np.abs(c2_[0]))), float(np.max(np.abs(c2_[1]))), 1e-300)
            d_col = max(float(np.max(np.abs(c1_[0] - c2_[0]))), float(np.max(np.abs(c1_[1] - c2_[1])))) / scale
            d_ray = max(float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) for a, b in zip(r1[1:], r2[1:]))
            tol = 1e-12 if W["dtype"] == "f64" else 1e-5
            bad = not (d_col <= tol and d_ray <= tol)
            if dist is not None:
                bad = all_flag(bad)
            exchange_check = {"transports": [t1, t2], "column_max_diff_over_scale": d_col, "ray_max_rel_diff": d_ray,
                              "tolerance": tol, "ok": not bad}
            if bad:
                ladder = ladder[2:]
                if rank == 0:
                    print("bench.py: the in-kernel exchange disagrees with the RCCL chain; timing the RCCL chain", file=sys.stderr, flush=True)
        except ExchangeFailed as e:
            exchange_check = {"ok": False, "error": str(e)}
            ladder = ladder[2:]
    m = None
    for env in ladder:
        if env:
            os.environ.update(env)
        try:
            m = measure(rays_per_gpu, args.steps, args.warmup, args.kernel_events, args.repeats, total=args.total_rays)
            break
        except ExchangeFailed as e:
            fell_back.append(str(e))
            if rank == 0:
                print(f"bench.py: flux exchange failed ({e}); trying the next transport", file=sys.stderr, flush=True)
    if m is None:
        raise SystemExit("bench.py: the flux exchange failed on every transport: " + " | ".join(fell_back))
    n_total, n_local, kern_ms, c1, finite = m["n_total"], m["n_local"], m["kern_ms"], m["counters"], m["finite"]
    walls = np.array(m["walls"])
    wall = float(np.median(walls))
    extra, late, streamed = None, None, None
    coupled = W["flags"] != "fixed"
    if world == 1 and coupled and not args.no_size_sweep:
        ks = max(args.steps // 4, 20)
        big = measure(4 * rays_per_gpu, ks, max(args.warmup // 4, 5), "none", 3)
        v = big["n_total"] * ks / float(np.median(big["walls"]))
        extra = {"rays_per_gpu": big["n_total"], "value": v,
                 "effective_algorithmic_over_hbm_peak": v * W["bytes"] / 1e9 / HBM_PEAK_GBS,
                 "note": "same workload at 4x the rays: the streamed working set no longer fits the 256 MiB Infinity Cache"}
        lt = measure(rays_per_gpu, args.steps, args.warmup, "none", 0, pre_steps=args.late_steps)
        lw = np.array(lt["walls"])
        late = {"pre_steps": args.late_steps + args.warmup, "value": lt["n_total"] * args.steps / float(np.median(lw)),
                "ms_per_step": float(np.median(lw)) / args.steps * 1e3, "state_finite": lt["finite"],
                "note": "same workload measured after the packet has dispersed over many levels (deposit spans widen)"}
    if world == 1 and coupled and not args.no_streamed_leg and c1.get("persist_resident_tiles", 0) > 0 \
            and not args.force_collective:
        # the HBM-honest figure, taken in the same run: the same workload with EVERY ray streamed through HBM each RK
        # stage (MSGW_REGTILES=0: no register-resident tiles), whose counter traffic equals the algorithmic bytes
        old = os.environ.get("MSGW_REGTILES")
        os.environ["MSGW_REGTILES"] = "0"
        try:
            sm = measure(rays_per_gpu, args.steps, args.warmup, "same", 3)
        finally:
            if old is None:
                os.environ.pop("MSGW_REGTILES", None)
            else:
                os.environ["MSGW_REGTILES"] = old
        streamed = sm

    if rank == 0:
        value = n_total * args.steps / wall
        bps = W["bytes"]
        persist_steps = c1.get("persist_steps", 0)
        nres = c1.get("persist_resident_tiles", 0)
        narrow = c1.get("fixed_narrow", 0)
        table, table_err = load_counter_table()
        digest = kernel_src_digest()

        def counters_for(key):
            t = table.get(key) if table else None
            if not isinstance(t, dict):
                return None, f"no PMC measurement committed for {key} (profiles/traffic.json)" + (f": {table_err}" if table_err else "")
            return t, None

        def hbm_traffic(t, rays, steps_per_launch):
            if t is None:
                return None
            if "bytes_per_ray_launch" in t:
                return t["bytes_per_ray_launch"] * rays
            if "bytes_per_ray_step" in t:
                return t["bytes_per_ray_step"] * rays * steps_per_launch
            return None

        if persist_steps:        # one launch covers persist_steps RK3 steps (persistent coupled kernel / fused fixed kernel)
            per_launch_bytes = bps * persist_steps * n_local
            kernel_name = W["kernel"]
            steps_per_launch = persist_steps
        else:
            per_launch_bytes = bps / 3 * n_local
            kernel_name = "k_ray_stage"
            steps_per_launch = 1.0 / 3
        key = f"{wl if wl != 'config4' else 'config3'}:{W['dtype']}:{n_local}:res{nres}" + (":narrow" if narrow else "")
        roofline = None
        if kern_ms:
            ksec = kern_ms * 1e-3
            eff = per_launch_bytes / ksec / 1e9
            t, t_note = counters_for(key)
            traffic = hbm_traffic(t, n_local, steps_per_launch if persist_steps else 1.0 / 3)
            stale = bool(t is not None and t.get("src_digest") not in (None, digest))
            issue_bound = bool(persist_steps) and (wl == "config2" or nres > 0)
            common = {
                "kernel": kernel_name, "kernel_ms_avg": kern_ms,
                "traffic": traffic, "traffic_source": (t or {}).get("source", t_note),
                "traffic_stale": stale,
                "traffic_note": "HBM bytes per launch by the PMC counters of the committed rocprofv3 passes (separate --pmc "
                                "runs, FETCH_SIZE doubled + WRITE_SIZE, x1024; MI355X_MICROARCH.md) scaled to this launch's "
                                "rays x steps; `traffic_stale`: the kernel sources have changed since those passes",
                "hbm_counter_gbs": None if traffic is None else traffic / ksec / 1e9,
                "hbm_counter_frac": None if traffic is None else traffic / ksec / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": per_launch_bytes,
                "effective_algorithmic_gbs": eff,
                "effective_algorithmic_over_hbm_peak": eff / HBM_PEAK_GBS,
                "effective_algorithmic_is": "SURVEY 8d words per ray-step x rays x steps of one launch / the launch's HIP-event "
                                            "duration: an EFFECTIVE rate, not a utilisation (a kernel that keeps the evolving "
                                            "state in registers moves fewer bytes than that, so it can exceed the HBM peak)",
                "events": "HIP events around every launch, " +
                          ("inside the timed region (median over the repeats)" if same else "second pass of the same K steps"),
            }
            if issue_bound:
                vi = None if t is None else t.get("valu_wave_insts_per_ray_step")
                ach = None if vi is None else vi * n_local * persist_steps / ksec / 1e9
                roofline = {"bound": "valu_issue", "achieved": ach, "peak": VALU_ISSUE_PEAK_GINST,
                            "unit": "1e9 wave64 VALU instructions/s",
                            "frac": None if ach is None else ach / VALU_ISSUE_PEAK_GINST,
                            "achieved_is": "wave-level VALU instructions of one launch (SQ_INSTS_VALU of the committed rocprofv3 "
                                           "pass for this kernel flavour, per ray-step, x this launch's rays x steps) / the "
                                           "launch's HIP-event duration measured in this run; peak = 1024 SIMDs x 2.4 GHz / 4 "
                                           "clocks per wave64 instruction",
                            "why_not_hbm": ("the fused kernel keeps rr, mm in registers for all steps of a launch"
                                            if wl == "config2" else
                                            "the persistent kernel keeps the evolving ray state in registers for the whole "
                                            "launch: HBM carries only the static factors (see traffic / hbm_counter_frac), so "
                                            "the HBM roofline does not bound it; the all-rays-streamed flavour of the same "
                                            "kernel, measured in this run, is under hbm_streamed"),
                            **common}
            else:
                roofline = {"bound": "hbm", "achieved": eff, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": eff / HBM_PEAK_GBS,
                            "achieved_is": "ALGORITHMIC bytes (SURVEY 8d) of one launch / its HIP-event duration; every ray "
                                           "streams through HBM once per RK stage on this path", **common}
            roofline["limiter"] = {
                "config3": "VALU issue inside the tiles (a pass of a ray workgroup: table 1.8 + tiles 6.8 + publish 1.2 us; the two "
                           "workgroups of a CU do the first and the last at the same time) and the reduce chain, 9.4 us from the "
                           "last row to the release, which is as long as a pass (DESIGN.md 6, 8; profiles/r03_persist_timeline.txt)",
                "config4": "as config3, plus the rank sum on the reduce chain (DESIGN.md 5a)",
                "config5": "VALU issue (online saturation is a second table look-up and two more divisions per ray-stage); the "
                           "deposit of a dispersed packet (wavefronts whose rays span many levels)",
                "config2": "VALU issue (84-92 % of the step's own issue time from two wavefronts per SIMD on); at 1e5 rays the "
                           "chip holds 1.53 wavefronts per SIMD (one ray per lane): 539 SIMDs with two, 485 with one that finish "
                           "early and idle (DESIGN.md 4, K1f; profiles/r03_config2_wave_count_probe.txt)"}.get(wl) if persist_steps else None
            if streamed is not None and streamed["kern_ms"]:
                sc = streamed["counters"]
                s_sec = streamed["kern_ms"] * 1e-3
                s_bytes = bps * sc.get("persist_steps", 0) * streamed["n_local"]
                s_key = f"{wl if wl != 'config4' else 'config3'}:{W['dtype']}:{streamed['n_local']}:res0"
                st_, _ = counters_for(s_key)
                s_tr = hbm_traffic(st_, streamed["n_local"], sc.get("persist_steps", 0))
                s_wall = float(np.median(streamed["walls"]))
                roofline["hbm_streamed"] = {
                    "what": "the same workload with MSGW_REGTILES=0 (every ray streamed through HBM once per RK stage), "
                            "measured in this run: the HBM-honest yardstick of this path",
                    "bound": "hbm", "achieved": s_bytes / s_sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": s_bytes / s_sec / 1e9 / HBM_PEAK_GBS, "kernel_ms_avg": streamed["kern_ms"],
                    "ms_per_step": s_wall / streamed["steps"] * 1e3,
                    "value": streamed["n_total"] * streamed["steps"] / s_wall,
                    "register_resident_tiles_per_workgroup": sc.get("persist_resident_tiles", 0),
                    "traffic": s_tr, "traffic_stale": bool(st_ is not None and st_.get("src_digest") not in (None, digest)),
                    "hbm_counter_frac": None if s_tr is None else s_tr / s_sec / 1e9 / HBM_PEAK_GBS}
            try:
                roofline["copy_ceiling_gbs"] = copy_ceiling(torch)
            except Exception as e:      # noqa: BLE001
                roofline["copy_ceiling_gbs"] = None
                roofline["copy_ceiling_error"] = str(e)
        transport = _capi.TRANSPORTS.get(c1.get("transport", 0), "?")
        if world > 1 or args.force_collective:
            how = {"device_ipc": "inside the persistent kernel: every rank writes its row into the HBM of all ranks "
                                 "(HIP IPC peer mappings over xGMI), rank-order sum of the local copy",
                   "host_shm": "inside the persistent kernel through a node-shared host segment (PCIe), rank-order sum",
                   "rccl": "by ncclAllReduce once per RK stage (lagged launch chain)"}.get(transport, transport)
            if not persist_steps and transport != "rccl":
                how = "by ncclAllReduce once per RK stage (lagged launch chain; the persistent kernel declined)"
            par = f"rays sharded x{world}, column replicated; flux summed over the ranks {how}"
        else:
            par = "single GPU"
        out = {
            "metric": "ray-steps/sec", "value": value, "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": len(walls),
            "ms_per_step": wall / args.steps * 1e3,
            "ms_per_step_min": float(walls.min()) / args.steps * 1e3,
            "ms_per_step_max": float(walls.max()) / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": W["dtype"], "data": "synthetic",
            "config": {"workload": W["text"],
                       "rays_total": n_total, "rays_per_gpu": n_total // world if strong else rays_per_gpu,
                       "ngrid": args.ngrid,
                       "dt": DT, "parallelism": par, "transport": transport,
                       **({"transport_fallbacks": fell_back} if fell_back else {}),
                       **({"exchange_check": exchange_check} if exchange_check is not None else {}),
                       "graph_steps": c1["graph_steps"], "persist_steps": persist_steps,
                       "launch_workgroups": c1.get("launch_grid", 0),
                       "launch_ray_workgroups": c1.get("launch_ray_workgroups", 0),
                       "launch_reducer_workgroups": c1.get("launch_reducers", 0),
                       "cooperative_launch": bool(c1.get("cooperative", 0)),
                       "per_stage_kernel_workgroups": c1["blocks"],
                       "register_resident_tiles_per_workgroup": nres,
                       **({"one_ray_per_lane": bool(narrow)} if wl == "config2" else {}),
                       "kernel_src_digest": digest,
                       "timing": "median of `repeats` identical experiments (fresh state, `warmup` steps, then `steps` steps timed)"},
            "effective_algorithmic_over_hbm_peak": value * bps / 1e9 / (HBM_PEAK_GBS * world),
            "state_finite": finite,
            "roofline": roofline,
        }
        if extra is not None:
            out["larger_problem"] = extra
        if late is not None:
            out["late_time"] = late
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, grid, uu, vv)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
