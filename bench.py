#!/usr/bin/env python3
"""Headline benchmark: C/GMRES control steps/s for a batch of arm_type_inverted_pendulum controllers
(BASELINE.json: batch 4096 per GPU, N = dv = 50 horizon stages, k_max = 10, fp64), closed loop with the
example's forward-Euler plant, everything resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One "step" = one control tick of the whole per-GPU batch; the closed loop runs on the device and the tick kernel
advances cgmres_cpp_amd.TICKS_PER_LAUNCH (10) consecutive ticks per launch.  Weak scaling:
every rank owns `--batch` controllers; rank 0 draws the whole job's seeded inputs and the shards are
scattered over RCCL (cgmres_cpp_amd/sharding.py); there is no collective inside the timed region because
controller instances are independent.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MODEL, DV, KMAX = "pendulum", 50, 10
DIM_X, DIM_U, DIM_P = 4, 3, 2
# rocprofv3 PMC summary of this same command (tools/profile_bench.sh + tools/summarise_profile.py)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_v10_wg_bench_pmc.json")


def algorithmic_bytes(k, L=DIM_U * DV, scalar=8):
    """HBM bytes one instance-tick must move when every length-L operand of a fused
    vector op is one transfer: S*[L*(8 + 6k + k(k-1)/2) + (3+k)*(dim_p*(dv+1) + dim_x) + dim_u]."""
    return scalar * (L * (8 + 6 * k + k * (k - 1) // 2) + (3 + k) * (DIM_P * (DV + 1) + DIM_X) + DIM_U)


def cpu_baseline(batch, tol, warm, seconds_budget):
    """The oracle (CPU restatement, bit-exact with the reference) on this host's cores: same seeded inputs,
    closed loop, instances statically partitioned over all available threads."""
    from oracle import orc
    if not os.path.exists(orc.ORACLE_SO):
        orc.build(ref=False)
    threads = len(os.sched_getaffinity(0))
    x0, u0, p = orc.batch_scenario(orc.PENDULUM, batch)
    ctrls = []
    for i in range(batch):
        c = orc.Controller(orc.PENDULUM, DV, KMAX, tol)
        orc.start_controller(c, x0[i], u0[i], p[i])
        ctrls.append(c)
    secs_w, _, x = orc.run_closed_loop(ctrls, x0, warm, threads)
    # size the timed part from the warm-up rate so the leg stays within its budget
    ticks = int(max(5, min(200, seconds_budget / max(secs_w / warm, 1e-6))))
    secs, _, _ = orc.run_closed_loop(ctrls, x, ticks, threads)
    ks = np.array([c.last_solve()[0] for c in ctrls])
    return {"value": batch * ticks / secs, "unit": "control steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle/liboracle.so (CPU restatement, bit-exact vs reference), {batch} controllers x "
                      f"{ticks} closed-loop ticks after {warm} warm-up ticks, tol={tol:g}, {threads} std::threads, "
                      f"mean Arnoldi iterations last tick {ks.mean():.2f}"}


def measured_traffic(kernel_variant, ticks_per_launch):
    """HBM bytes per launch from the committed PMC passes of this command (null when none matches)."""
    try:
        s = json.load(open(PMC_SUMMARY))
    except OSError:
        return None, None
    want = "tick_wg_kernel" if kernel_variant == 2 else "tick_lane_kernel"
    if want not in s.get("kernel", "") or s.get("ticks_per_launch", 1) != ticks_per_launch:
        return None, None
    return s["hbm_bytes_per_launch"], os.path.relpath(PMC_SUMMARY, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="controllers per GPU")
    ap.add_argument("--tol", type=float, default=0.0,
                    help="0 = fixed-k mode (always k_max Arnoldi iterations, deterministic work; headline); "
                         "1e-6 = the reference's early-exit mode")
    ap.add_argument("--variant", type=int, default=0, help="kernel mapping: 0 default, 1 lane, 2 wg")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="wall budget of the timed CPU-baseline part")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-mode", action="store_true", help="skip the secondary tol=1e-6 measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import cgmres_cpp_amd as cg
    from cgmres_cpp_amd import scenarios
    from cgmres_cpp_amd.sharding import scatter_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    B = args.batch
    # ---- inputs: rank 0 draws the whole job (seeded), shards go out over RCCL (the only exchange) ------
    full = scenarios.batch(MODEL, B * world) if rank == 0 else (None, None, None)
    x_d = scatter_rows(full[0], B * world, DIM_X, world, rank, dev, dist)
    u0_d = scatter_rows(full[1], B * world, DIM_U, world, rank, dev, dist)
    p_d = scatter_rows(full[2], B * world, DIM_P, world, rank, dev, dist)
    torch.cuda.synchronize()
    x0_h, u0_h, p_h = x_d.cpu().numpy(), u0_d.cpu().numpy(), p_d.cpu().numpy()
    assert x0_h.shape == (B, DIM_X)

    stream = torch.cuda.current_stream().cuda_stream
    resolved = {}

    def run(tol, steps, warmup):
        ctrl = cg.CgmresBatch(MODEL, batch=B, dv=DV, k_max=KMAX, tol=tol, device=local, stream=stream,
                              variant=args.variant)
        resolved["variant"] = ctrl.variant
        ctrl.set_ptau_repeat(p_h)
        ctrl.init_u0(u0_h)
        ctrl.init_u0_newton(u0_h, x0_h, p_h, 10)
        x = torch.from_numpy(x0_h).to(dev)
        u = torch.zeros(B, DIM_U, dtype=torch.float64, device=dev)
        ctrl.closed_loop_device(x, u, warmup)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctrl.timer_start()
        ctrl.closed_loop_device(x, u, steps)
        kernel_ms = ctrl.timer_stop()  # HIP events on the launch stream, around exactly the K launches
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        n_ax, reason = ctrl.get_status()
        finite = bool(torch.isfinite(u).all().item()) and bool(torch.isfinite(x).all().item())
        ctrl.close()
        tt = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt[0].item(), tt[1].item(), n_ax, finite

    wall, kernel_ms, n_ax, finite = run(args.tol, args.steps, args.warmup)
    if not finite:
        sys.exit("non-finite control output: refusing to report a number")
    value = B * world * args.steps / wall
    ms_per_step = wall * 1e3 / args.steps
    # one launch of the wg mapping = TICKS_PER_LAUNCH consecutive ticks of the batch (the lane mapping: one tick)
    tpl = cg.TICKS_PER_LAUNCH if resolved["variant"] == 2 else 1
    n_launches = -(-args.steps // tpl)
    launch_ms = kernel_ms / n_launches
    bytes_per_tick = float(sum(algorithmic_bytes(int(k)) for k in n_ax)) if args.tol > 0 else \
        float(B * algorithmic_bytes(KMAX))
    bytes_per_launch = bytes_per_tick * args.steps / n_launches
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(resolved["variant"], args.steps / n_launches) \
        if B == 4096 and args.tol == 0.0 else (None, None)

    ref_mode = None
    if not args.no_ref_mode and args.tol == 0.0:
        w2, _, n2, _ = run(1e-6, args.steps, args.warmup)
        ref_mode = {"tol": 1e-6, "value": B * world * args.steps / w2, "ms_per_step": w2 * 1e3 / args.steps,
                    "mean_arnoldi_last_tick": float(np.mean(n2))}

    kernel_name = {1: "tick_lane_kernel", 2: "tick_wg_kernel"}[resolved["variant"]]
    out = {
        "metric": "C/GMRES control steps/sec, batch=4096 N=50 kmax=10; HBM GB/s vs roofline",
        "value": value, "unit": "control steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "arm_type_inverted_pendulum controllers, closed loop with on-device Euler plant",
                   "batch_per_gpu": B, "global_batch": B * world, "N": DV, "kmax": KMAX, "tol": args.tol,
                   "mode": "fixed-k (tol=0, every instance runs k_max Arnoldi iterations)" if args.tol == 0
                   else "reference early-exit", "variant": resolved["variant"], "parallelism": f"batch-shard x{world}",
                   "inputs": "splitmix64(12345) perturbed x0/targets, Newton-initialised U0 (SURVEY.md §8d)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": f"{kernel_name} ({tpl} control step(s) of the batch per launch)", "launch_ms": launch_ms,
                     "ticks_per_launch": args.steps / n_launches, "algorithmic_bytes_per_launch": bytes_per_launch},
    }
    if ref_mode:
        out["reference_mode"] = ref_mode
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, args.tol, args.warmup, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
