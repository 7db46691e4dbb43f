#!/usr/bin/env python3
"""Secondary measurements (not bench lines): the other BASELINE.json configurations, closed loop on the device with
the example plants, fixed-k mode (tol = 0) and reference mode (tol = 1e-6).

    python tools/bench_configs.py [--config all|1|2|3|4|5|headline|gmres] [--steps K] [--warmup W] > gpurun_out/configs.jsonl
    python tools/bench_configs.py --gpus N --config 4        (N > 1: starts its own torch.distributed.run child, like bench.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_configs.py --gpus N --config 4

Every line is parity-gated before it is printed (tests/parity_gate.py, the oracle as the checker, outside the timed
region): a spread sample of EVERY member's instances is continued from the state the timed region left behind — one
teacher-forced tick at 1e-9 (fp32 members: 1e-4 against the fp32 oracle), then 10 fused free-running ticks at 1e-6
(fp32: 2e-3) — and the script refuses to print a number on mismatch.

Config 4 (multiple_controller: 4096 Model1 = MSD + 4096 Model2 = pendulum controllers, both N = 50) under
torch.distributed: EACH model's sub-batch is split over all ranks (cgmres_cpp_amd.sharding.shard_bounds — an MSD tick
moves twice the bytes of a pendulum tick, so model-per-GPU would imbalance), every rank steps its two
shards on two HIP streams (cgmres_cpp_amd.multi.MultipleController), no collective inside the timed region; the time is
the max over ranks between barriers.  Config 5 (65536 pendulum controllers, N = 100, k = 20, fp32) is likewise split
over the ranks (8192 per GPU at 8)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {0: "pendulum", 1: "msd", 2: "semiactive"}
DIMS = {0: (4, 3, 2), 1: (4, 6, 2), 2: (2, 3, 0)}


def alg_bytes(model, dv, k, scalar):
    nx, nu, npar = DIMS[model]
    L = nu * dv
    return scalar * (L * (8 + 6 * k + k * (k - 1) // 2) + (3 + k) * (npar * (dv + 1) + nx) + nu)


def job_bytes(model, dv, kmax, dtype, tol, n_ax, B):
    s = 8 if dtype == "f64" else 4
    return float(sum(alg_bytes(model, dv, int(k), s) for k in n_ax)) if tol > 0 else float(B * alg_bytes(model, dv, kmax, s))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="all")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--tols", default="0,1e-6")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal with ranks sharing GPUs")
    ap.add_argument("--gpus", type=int, default=0, help="ranks (one per GPU); 0 = whatever WORLD_SIZE says (1 without it)")
    ap.add_argument("--check-sample", type=int, default=24, help="instances per member and rank checked against the oracle")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # before torch is imported: the parent never touches the GPU
        from bench import self_launch
        sys.exit(self_launch(os.path.abspath(__file__), args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from cgmres_cpp_amd import scenarios
    from cgmres_cpp_amd.multi import MultipleController
    from cgmres_cpp_amd.sharding import shard_bounds
    from tests.parity_gate import OracleSample, ParityError, gate_continuation
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus and args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE = {world}")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(label, members, tol, dtype="f64"):
        """members: list of (model, global batch, dv, kmax); every member's batch is sharded over the ranks and the
        members of one rank run on their own streams."""
        specs, shards = [], []
        for model, Bg, dv, kmax in members:
            lo, hi = shard_bounds(Bg, world, rank)
            specs.append(dict(model=model, batch=hi - lo, dv=dv, k_max=kmax, tol=tol, dtype=dtype, device=local))
            shards.append((lo, hi))
        mc = MultipleController(specs, device=local)  # (creates one stream per member, alternating priorities)
        tdt = torch.float64 if dtype == "f64" else torch.float32
        xs, us, hosts = [], [], []
        for m, (model, Bg, dv, kmax), (lo, hi) in zip(mc.members, members, shards):
            x0, u0, p = scenarios.batch(NAMES[model], Bg)  # seeded: every rank draws the job and keeps its slice
            x0, u0, p = x0[lo:hi], u0[lo:hi], p[lo:hi]
            hosts.append((x0, u0, p))
            if DIMS[model][2]:
                m.set_ptau_repeat(p)
            m.init_u0(u0)
            m.init_u0_newton(u0, x0, p, 10)
            xs.append(torch.from_numpy(np.ascontiguousarray(x0)).to(dev, tdt))
            us.append(torch.zeros(hi - lo, DIMS[model][1], dtype=tdt, device=dev))
        torch.cuda.synchronize()  # the torch allocations/copies above ran on torch's stream, the launches do not
        mc.closed_loop_device(xs, us, args.warmup)
        mc.synchronize()
        sync_all()
        t0 = time.perf_counter()
        mc.closed_loop_device(xs, us, args.steps)
        mc.synchronize()
        sync_all()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        by = torch.zeros(1, dtype=torch.float64, device=cdev)
        ok = torch.tensor([1 if all(bool(torch.isfinite(u).all().item()) for u in us) else 0], device=cdev)
        kmean = []
        for m, (model, Bg, dv, kmax) in zip(mc.members, members):
            n_ax, _ = m.get_status()
            by += job_bytes(model, dv, kmax, dtype, tol, n_ax, m.batch)
            kmean.append(float(np.mean(n_ax)))
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(by, op=dist.ReduceOp.SUM)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        variants = [m.variant for m in mc.members]
        # parity gate (outside the timed region): every member, from the state the timed region left behind, with the
        # OTHER members' kernels still co-resident for the fused ticks (the joint mode is what was timed)
        gate, err = [], None
        f32 = dtype == "f32"
        try:
            if args.check_sample:
                for m, (model, Bg, dv, kmax), (x0, u0, p), x, u in zip(mc.members, members, hosts, xs, us):
                    chk = OracleSample(model, dv, kmax, tol, x0, u0, p, args.check_sample, dtype)
                    gate.append(gate_continuation(m, chk, x, u, mc.synchronize, 1e-4 if f32 else 1e-9,
                                                  2e-3 if f32 else 1e-6, tol == 0.0 and not f32))
        except ParityError as e:
            err = str(e)
        bad = torch.tensor([0 if (err is None and ok.item()) else 1], dtype=torch.int32, device=cdev)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        mc.close()
        if err is not None:
            sys.stderr.write(f"[rank {rank}] PARITY FAILURE ({label}, tol={tol:g}): {err}\n")
        if bad.item():
            if world > 1:
                dist.destroy_process_group()
            sys.exit("parity check against the oracle failed: refusing to report a number")
        total = sum(b for _, b, _, _ in members)
        ms = tt.item() / args.steps * 1e3
        out = {"config": label, "members": [f"{NAMES[m]} B={b} dv={dv} k={k}" for m, b, dv, k in members], "dtype": dtype,
               "tol": tol, "n_gpus": world, "variants": variants, "ms_per_tick": ms,
               "steps_per_s": total * args.steps / tt.item(), "mean_arnoldi_last_tick_rank0": kmean,
               "algorithmic_GBps": by.item() / (ms * 1e-3) / 1e9,
               "frac_of_8TBps_per_gpu": by.item() / (ms * 1e-3) / 1e9 / 8000.0 / world, "finite": bool(ok.item()),
               "parity_rank0": gate}
        if rank == 0:
            print(json.dumps(out), flush=True)

    table = {
        "3": ("cfg3 semiactive", [(2, 4096, 50, 10)], "f64"),
        "4msd": ("cfg4 MSD half alone", [(1, 4096, 50, 10)], "f64"),
        "headline": ("headline pendulum", [(0, 4096, 50, 10)], "f64"),
        "4": ("cfg4 multiple_controller: MSD + pendulum on two streams", [(1, 4096, 50, 10), (0, 4096, 50, 10)], "f64"),
        "5": (f"cfg5 pendulum N=100 k=20 fp32 ({8192 * world} controllers = 8192 per GPU)",
              [(0, 8192 * world, 100, 20)], "f32"),
        "2": ("cfg2 pendulum B=256", [(0, 256, 50, 10)], "f64"),
        "1": ("cfg1 one MSD controller N=20 k=5", [(1, 1, 20, 5)], "f64"),
    }
    def run_gmres(batch=4096):
        """Stand-alone Gmres with a caller-supplied operator (reference include/gmres.hpp:8-129) on the device: `batch`
        independent systems through cgmres_hip_gmres_user, host pointers in and out (PCIe and the launch included),
        checked against numpy's dense solve of the same systems."""
        import cgmres_cpp_amd as cg
        from cgmres_cpp_amd import plugin
        ops = os.path.join(ROOT, "tests", "user_models", "gmres_ops.hpp")
        for cls, name, L in (("ConvDiffOp150", "convdiff150", 150), ("ConvDiffOp300", "convdiff300", 300)):
            oid = plugin.register_operator(plugin.build_operator(ops, cls, name=name))
            e = np.arange(L)
            i = np.arange(batch)[:, None]
            P = np.concatenate([0.4 + 0.07 * (i % 12), 0.35 - 0.02 * (i % 12)], axis=1)
            Bv = np.sin(0.3 * e[None, :] + 0.5 * i) + 0.1 * e[None, :]
            X0 = 0.01 * (e[None, :] - (i % 12))
            for kmax, tol in ((10, 0.0), (30, 1e-6)):
                cg.gmres_user(oid, X0, Bv, kmax, tol, P)  # (first call: plugin workspace, code upload)
                t0 = time.perf_counter()
                x, n_ax, why = cg.gmres_user(oid, X0, Bv, kmax, tol, P)
                dt = time.perf_counter() - t0
                # residual of a sample against the dense operator
                worst = 0.0
                for b in (0, batch // 2, batch - 1):
                    A = np.zeros((L, L))
                    for r in range(L):
                        A[r, r] += 2.0 + P[b, 0] + 0.01 * r
                        if r > 0:
                            A[r, r - 1] -= 1.0 + P[b, 1]
                        if r + 1 < L:
                            A[r, r + 1] -= 1.0 - P[b, 1]
                        A[r, (r * 7 + 3) % L] += 0.05
                    worst = max(worst, float(np.linalg.norm(A @ x[b] - Bv[b]) / np.linalg.norm(Bv[b])))
                if rank == 0:
                    print(json.dumps({"config": f"stand-alone Gmres, {cls} (len {L}), k_max {kmax}, tol {tol:g}", "batch": batch,
                                      "ms_per_call_host_pointers": dt * 1e3, "systems_per_s": batch / dt,
                                      "mean_arnoldi": float(np.mean(n_ax)), "worst_relative_residual_of_3": worst,
                                      "kernel": "gmres_wave_kernel (one wavefront per system)"}), flush=True)

    keys = list(table) if args.config == "all" else args.config.split(",")
    if "gmres" in keys:
        run_gmres()
        keys = [k for k in keys if k != "gmres"]
    for k in keys:
        label, members, dtype = table[k]
        for tol in [float(t) for t in args.tols.split(",")]:
            run(label, members, tol, dtype)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
