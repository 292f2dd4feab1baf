#!/usr/bin/env python3
"""
bench.py -- ray-steps/s of the MI355X ray-propagation path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload coupled|fixed]

A "step" is one lprop.RK3 step (3 RHS stages incl. flux deposit and mean-flow
update) of every resident ray.  Default workload at N=1 is BASELINE config 3:
1e6 rays of the synthetic Gaussian source spectrum, interactive
mean flow, fp64.  For N>1 (launched by torch.distributed.run, one rank per GPU)
every rank holds `--rays-per-gpu` rays (weak scaling; N=8 with 1.25e6 rays/GPU
is config 4) and the 2x(ngrid-2) flux profile is RCCL all-reduced once per RK
stage inside the C library.  Inputs are resident in HBM before the timed region.

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

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic fp64 bytes per ray per RK3 step (SURVEY 8d / BASELINE.md section 4)
BYTES_PER_RAY_STEP = {"coupled": 280.0, "fixed": 48.0}
LAUNCHES_PER_STEP = {"coupled": 3, "fixed": 1}
KERNEL_NAME = {"coupled": "k_ray_stage", "fixed": "k_ray_step_fixed"}
DT = 120.0                      # raytracer.py:46


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


def cpu_baseline(workload, grid, uu, vv, budget_s=12.0):
    """Time the CPU restatement of the reference path on this box's host cores
    (rank 0, N=1 only).  kind "port": the reference is pure Python and never
    travels; what is timed is oracle/msgwam_oracle.py in its `loop=True` mode,
    i.e. the reference's algorithmic structure incl. the interpreted per-ray
    deposit loop (lib/libprop.py:151-163), single thread."""
    from oracle import msgwam_oracle as orc
    from oracle.c_oracle import COracle
    from msgwam_amd.spectrum import gaussian_spectrum
    n = 4000
    s0 = orc.Setup(grid)
    sp = gaussian_spectrum(n, s0.grids, s0.rhobar, alpha=0.01, nz=100, nd=4)
    s = orc.Setup(grid, dkk=sp["dkk"], dll=sp["dll"], rr_mm_area=sp["area"])
    s.set_pressure_gradient(uu, vv)
    keys = ["dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm"]
    st0 = [sp[k] for k in keys] + [uu, vv]
    fixed = workload == "fixed"

    def run(fn, min_steps=1):
        st, steps, t0 = st0, 0, time.perf_counter()
        while steps < min_steps or time.perf_counter() - t0 < budget_s / 3:
            st = fn(st)
            steps += 1
        return n * steps / (time.perf_counter() - t0), steps

    loop_rate, loop_steps = run(lambda st: orc.rk3(s, DT, st, loop=True, fixed_background=fixed))
    vec_rate, _ = run(lambda st: orc.rk3(s, DT, st, loop=False, fixed_background=fixed))
    co = COracle(s, fixed_background=fixed)
    c_rate, _ = run(lambda st: co.step(DT, 1, st), min_steps=3)
    return {"value": loop_rate, "unit": "ray-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} rays x {loop_steps} RK3 steps, synthetic Gaussian spectrum, ngrid 101, "
                      f"oracle/msgwam_oracle.py loop=True (reference's per-ray Python deposit loop)",
            "host_cores": os.cpu_count(),
            "vectorised_numpy_value": vec_rate, "c_port_value": c_rate}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["coupled", "fixed"], default="coupled")
    ap.add_argument("--rays-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--ngrid", type=int, default=101)
    ap.add_argument("--blocks-per-cu", type=int, default=int(os.environ.get("MSGW_BLOCKS_PER_CU", 4)))
    ap.add_argument("--graph-steps", type=int, default=int(os.environ.get("MSGW_GRAPH_STEPS", 4)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-size-sweep", action="store_true", help="skip the extra 4x-rays measurement (N=1 only)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 only: run the multi-GPU path with a 1-rank communicator (diagnostic; "
                         "MSGW_EXCHANGE=0 selects the RCCL launch chain instead of the in-kernel exchange)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N>1 launch (only carries the unique id, barriers and the "
                         "max-over-ranks time; the flux exchange is the library's own)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="diagnostic: all ranks use GPU 0 (rehearsal of the N>1 path on a 1-GPU box; implies the "
                         "exchange-only communicator because RCCL refuses two ranks on one device; use --backend gloo)")
    ap.add_argument("--kernel-events", choices=["separate", "same", "none"], default="same",
                    help="where the per-launch HIP-event timing of the dominant kernel is taken")
    args = ap.parse_args()

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
    flags = _capi.FIXED_BACKGROUND if args.workload == "fixed" else 0
    same = args.kernel_events == "same"
    uid = None
    if args.force_collective and world == 1:
        os.environ["MSGW_FORCE_COLLECTIVE"] = "1"
        uid = _capi.comm_unique_id()
    if world > 1:
        box = [_capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]

    def measure(rays_per_gpu, steps, warmup, kernel_events):
        """One timed region: `steps` RK3 steps of rays_per_gpu rays per rank, state resident."""
        n_total = rays_per_gpu * world
        lo, hi = shard_bounds(n_total, world, rank)
        sp = gaussian_spectrum(n_total, grids, lprop.rhobar, alpha=0.01, start=lo, stop=hi)
        n_local = hi - lo
        p = _capi.Propagator(args.ngrid, n_local, device=local_rank)
        p.set_config(0.01, 0.0, 1.0, False)
        p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
        p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"],
                      sp["dkk"], sp["dll"], sp["area"])
        p.set_tuning(args.blocks_per_cu, args.graph_steps)
        if uid is not None:
            p.comm_init(uid, rank, world)

        def fence():
            p.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        p.step(DT, warmup, flags)
        fence()
        c0 = p.counters()
        t0 = time.perf_counter()
        p.step(DT, steps, flags | (_capi.TIME_KERNELS if kernel_events == "same" else 0))
        fence()
        wall = time.perf_counter() - t0
        c1 = p.counters()
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        kern_ms, launches = None, 0
        if kernel_events != "none":                    # HIP events on the library's own stream
            if kernel_events != "same":
                p.step(DT, steps, flags | _capi.TIME_KERNELS)
                p.sync()
                c1 = p.counters()
            launches = c1["ray_kernel_launches"] - c0["ray_kernel_launches"]
            kern_ms = (c1["ray_kernel_ms_sum"] - c0["ray_kernel_ms_sum"]) / max(launches, 1)
        finite = bool(np.all(np.isfinite(p.download_rays()[1])))
        p.close()
        return dict(n_total=n_total, n_local=n_local, wall=wall, kern_ms=kern_ms, launches=launches,
                    counters=c1, finite=finite)

    m = measure(args.rays_per_gpu, args.steps, args.warmup, args.kernel_events)
    n_total, n_local, wall, kern_ms, c1, finite = (m["n_total"], m["n_local"], m["wall"], m["kern_ms"],
                                                  m["counters"], m["finite"])
    extra = None
    if world == 1 and args.workload == "coupled" and not args.no_size_sweep:
        big = measure(4 * args.rays_per_gpu, max(args.steps // 4, 20), max(args.warmup // 4, 5), "none")
        v = big["n_total"] * max(args.steps // 4, 20) / big["wall"]
        extra = {"rays_per_gpu": big["n_total"], "value": v,
                 "whole_job_hbm_frac": v * BYTES_PER_RAY_STEP["coupled"] / 1e9 / HBM_PEAK_GBS,
                 "note": "same workload at 4x the rays"}

    if rank == 0:
        value = n_total * args.steps / wall
        bps = BYTES_PER_RAY_STEP[args.workload]
        persist_steps = c1.get("persist_steps", 0)
        fused_note = None
        if persist_steps and args.workload == "fixed":
            # independent rays: all steps of the call run in ONE launch with rr, mm in registers, so the state
            # touches HBM once per launch; SURVEY 8d counts 48 B per ray-step (state materialised every step),
            # which this kernel does not move -- the HBM roofline does not bound it (FP64 VALU does)
            per_launch_bytes = bps * n_local
            kernel_name = KERNEL_NAME[args.workload]
            fused_note = (f"{persist_steps} steps fused in registers: bytes = one pass over the state per launch; "
                          "FP64-VALU bound, the HBM fraction is not a quality measure here")
        elif persist_steps:      # one persistent launch covers persist_steps RK3 steps (3 stages each)
            per_launch_bytes = bps * persist_steps * n_local
            kernel_name = "k_rk3_persist"
        else:
            per_launch_bytes = bps / LAUNCHES_PER_STEP[args.workload] * n_local
            kernel_name = KERNEL_NAME[args.workload]
        roofline = None
        if kern_ms:
            achieved = per_launch_bytes / (kern_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    t = json.load(open(tpath)).get(f"{kernel_name}:{n_local}")
                    if isinstance(t, dict) and persist_steps:      # committed PMC result, scaled to this launch
                        traffic = t["bytes_per_ray_step"] * n_local * persist_steps
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "kernel_ms_avg": kern_ms,
                        "algorithmic_bytes_per_launch": per_launch_bytes,
                        "events": "HIP events around every launch, " +
                                  ("inside the timed region" if same else "second pass of the same K steps")}
            if fused_note:
                roofline["note"] = fused_note
        out = {
            "metric": "ray-steps/sec", "value": value, "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("config3: 1e6 rays/GPU, interactive mean flow (flux deposit + u,v update "
                                    "every RK stage), synthetic Gaussian spectrum, fp64"
                                    if args.workload == "coupled" else
                                    "config2-style: fixed background, pure propagation, fp64"),
                       "rays_total": n_total, "rays_per_gpu": args.rays_per_gpu, "ngrid": args.ngrid,
                       "dt": DT, "parallelism": (f"rays sharded x{world}, column replicated; flux summed over the ranks " +
                                                 ("inside the persistent kernel (node-shared segment, rank order)"
                                                  if c1.get("exchange") and persist_steps else
                                                  "by ncclAllReduce once per RK stage (lagged launch chain)"))
                       if (world > 1 or args.force_collective) else "single GPU",
                       "graph_steps": c1["graph_steps"], "persist_steps": persist_steps, "blocks": c1["blocks"],
                       "register_resident_tiles_per_workgroup": c1.get("persist_resident_tiles", 0)},
            "whole_job_hbm_frac": None if fused_note else value * bps / 1e9 / (HBM_PEAK_GBS * world),
            "state_finite": finite,
            "roofline": roofline,
        }
        if extra is not None:
            out["larger_problem"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, grid, uu, vv)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
