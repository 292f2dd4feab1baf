#!/usr/bin/env python3
"""
bench.py -- ray-steps/s of the MI355X ray-propagation path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3|config2|config5] [--repeats R]

A "step" is one lprop.RK3 step (3 RHS stages incl. flux deposit and mean-flow update) of every resident ray.
Workloads (BASELINE.json configs; synthetic Gaussian source spectrum of SURVEY 8d, inputs resident in HBM before the
timed region):

  config3 (default)  1e6 rays per GPU, interactive mean flow, float64                    280 B per ray-step
  config2            1e5 rays per GPU, fixed background (pure propagation), float64       48 B per ray-step
  config5            1.25e6 rays per GPU, float32 state, online saturation + the
                     relaunch extension (alpha = 0.5), interactive mean flow              180 B per ray-step

For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds `--rays-per-gpu` rays (weak
scaling; N = 8 with 1.25e6 rays per GPU is config 4 / config 5) and the 2 x (ngrid-2) flux profile is summed over the
ranks once per RK stage inside the C library (`config.parallelism` names the transport).

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
DT = 120.0                      # raytracer.py:46

WORKLOADS = {
    # algorithmic bytes per ray per RK3 step: SURVEY 8d / BASELINE.md section 4
    "config3": dict(bytes=280.0, dtype="f64", rays=1_000_000, alpha=0.01, sat=False, flags=0, kernel="k_rk3_persist",
                    text="config3: 1e6 rays/GPU, interactive mean flow (flux deposit + u,v update every RK stage), "
                         "synthetic Gaussian spectrum, fp64"),
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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of K steps (0 = until the timed region is >= 50 ms, at least 5)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + sorted(ALIASES), default="config3")
    ap.add_argument("--rays-per-gpu", type=int, default=0, help="0 = the workload's BASELINE size")
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
    wl = ALIASES.get(args.workload, args.workload)
    W = WORKLOADS[wl]
    rays_per_gpu = args.rays_per_gpu or W["rays"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    def measure(rays, steps, warmup, kernel_events, repeats, pre_steps=0):
        """`repeats` timed regions of `steps` RK3 steps of `rays` rays per rank, state resident."""
        n_total = rays * world
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
    m = None
    for env in ladder:
        if env:
            os.environ.update(env)
        try:
            m = measure(rays_per_gpu, args.steps, args.warmup, args.kernel_events, args.repeats)
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
    extra, late = None, None
    coupled = W["flags"] != "fixed"
    if world == 1 and coupled and not args.no_size_sweep:
        ks = max(args.steps // 4, 20)
        big = measure(4 * rays_per_gpu, ks, max(args.warmup // 4, 5), "none", 3)
        v = big["n_total"] * ks / float(np.median(big["walls"]))
        extra = {"rays_per_gpu": big["n_total"], "value": v,
                 "whole_job_hbm_frac": v * W["bytes"] / 1e9 / HBM_PEAK_GBS,
                 "note": "same workload at 4x the rays: the streamed working set no longer fits the 256 MiB Infinity Cache"}
        lt = measure(rays_per_gpu, args.steps, args.warmup, "none", 0, pre_steps=args.late_steps)
        lw = np.array(lt["walls"])
        late = {"pre_steps": args.late_steps + args.warmup, "value": lt["n_total"] * args.steps / float(np.median(lw)),
                "ms_per_step": float(np.median(lw)) / args.steps * 1e3, "state_finite": lt["finite"],
                "note": "same workload measured after the packet has dispersed over many levels (deposit spans widen)"}

    if rank == 0:
        value = n_total * args.steps / wall
        bps = W["bytes"]
        persist_steps = c1.get("persist_steps", 0)
        fused_note = None
        if persist_steps and wl == "config2":
            # independent rays: all steps of the call run in ONE launch with rr, mm in registers, so the state
            # touches HBM once per launch; SURVEY 8d counts 48 B per ray-step (state materialised every step),
            # which this kernel does not move -- the HBM roofline does not bound it (FP64 VALU does)
            per_launch_bytes = bps * persist_steps * n_local
            kernel_name = W["kernel"]
            fused_note = (f"SURVEY 8d accounting (48 B per ray-step: state materialised once per step) x rays x the "
                          f"{persist_steps} steps of one launch; the kernel fuses those steps in registers and moves "
                          "48 B per ray per LAUNCH, so this is an effective rate -- the bound is the FP64 issue rate "
                          "(PMC: VALU 41 % busy at 1e5 rays, one wavefront per SIMD on 784 of 1024 SIMDs; "
                          "profiles/r02_config2_summary.md)")
        elif persist_steps:      # one persistent launch covers persist_steps RK3 steps (3 stages each)
            per_launch_bytes = bps * persist_steps * n_local
            kernel_name = "k_rk3_persist"
        else:
            per_launch_bytes = bps / 3 * n_local
            kernel_name = "k_ray_stage"
        roofline = None
        if kern_ms:
            achieved = per_launch_bytes / (kern_ms * 1e-3) / 1e9
            # HBM bytes actually moved per launch: committed rocprofv3 PMC result (profiles/traffic.json, separate
            # --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes), keyed by workload:dtype:rays:resident tiles
            traffic, traffic_note = None, None
            key = f"{wl}:{W['dtype']}:{n_local}:res{c1.get('persist_resident_tiles', 0)}"
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            try:
                t = json.load(open(tpath)).get(key)
                if isinstance(t, dict) and persist_steps and "bytes_per_ray_launch" in t:
                    traffic = t["bytes_per_ray_launch"] * n_local          # steps fused in registers: once per launch
                    traffic_note = t.get("source")
                elif isinstance(t, dict) and persist_steps:
                    traffic = t["bytes_per_ray_step"] * n_local * persist_steps
                    traffic_note = t.get("source")
                else:
                    traffic_note = f"no PMC measurement committed for {key} on this kernel path (profiles/traffic.json)"
            except Exception as e:      # noqa: BLE001
                traffic_note = f"profiles/traffic.json unreadable: {e}"
            roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "achieved_is": "ALGORITHMIC bytes (SURVEY 8d words per ray-step x rays x steps of one launch) / "
                                       "the launch's HIP-event duration -- an effective rate: a persistent kernel that "
                                       "keeps tiles in registers moves fewer bytes than that (see traffic)",
                        "traffic": traffic, "traffic_source": traffic_note,
                        "hbm_counter_gbs": None if traffic is None else traffic / (kern_ms * 1e-3) / 1e9,
                        "hbm_counter_frac": None if traffic is None else traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "limiter": {"config3": "dependent LDS / memory round trips inside the tile body at 2 wavefronts per SIMD, and the younger of a "
                                               "CU's two workgroups (its loop -- table 1.3 + tiles 8.2 + publish 1.3 us -- is the pass period; the "
                                               "reduce chain, 9.2 us from the last row to the release, is close behind); the persistent kernel "
                                               "keeps the evolving ray state in registers, so the measured HBM traffic is well below the "
                                               "algorithmic bytes (DESIGN.md 6)",
                                    "config5": "latency at 2 wavefronts per SIMD (PMC: VALU 54 % busy, waves waiting 52 % of their "
                                               "cycles); the deposit of a dispersed packet (wavefronts whose rays span many levels)",
                                    "config2": "1e5 rays are 784 wavefronts for 1024 SIMDs, each issuing its ~200 FP64 instructions "
                                               "per ray-stage alone (PMC: VALU 41 % busy); the state never leaves the registers"
                                    }.get(wl) if persist_steps else None,
                        "kernel_ms_avg": kern_ms,
                        "algorithmic_bytes_per_launch": per_launch_bytes,
                        "events": "HIP events around every launch, " +
                                  ("inside the timed region (median over the repeats)" if same else "second pass of the same K steps")}
            if fused_note:
                roofline["note"] = fused_note
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
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": W["dtype"], "data": "synthetic",
            "config": {"workload": W["text"],
                       "rays_total": n_total, "rays_per_gpu": rays_per_gpu, "ngrid": args.ngrid,
                       "dt": DT, "parallelism": par, "transport": transport,
                       **({"transport_fallbacks": fell_back} if fell_back else {}),
                       "graph_steps": c1["graph_steps"], "persist_steps": persist_steps, "blocks": c1["blocks"],
                       "register_resident_tiles_per_workgroup": c1.get("persist_resident_tiles", 0),
                       "timing": "median of `repeats` identical experiments (fresh state, `warmup` steps, then `steps` steps timed)"},
            "whole_job_hbm_frac": value * bps / 1e9 / (HBM_PEAK_GBS * world),
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
