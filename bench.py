#!/usr/bin/env python3
"""Headline benchmark: C/GMRES control steps/s for a batch of 4096 arm_type_inverted_pendulum controllers
(BASELINE.json: N = dv = 50 horizon stages, k_max = 10, fp64), closed loop with the example's forward-Euler
plant, everything resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One "step" = one control tick of the whole batch; the closed loop runs on the device and the tick kernel advances
cgmres_cpp_amd.TICKS_PER_LAUNCH (10) consecutive ticks per launch.

Scaling: the metric is a FIXED batch of 4096 controllers at 1/2/4/8 GPUs, so `value` is STRONG scaling — the 4096
instances are split into contiguous shards (cgmres_cpp_amd.sharding.shard_bounds), one process per GPU; rank 0
draws the seeded job and the shards go out once over RCCL; there is no collective inside the timed region
(instances are independent).  For N > 1 the weak-scaling figure (4096 controllers per GPU) is measured in the same
run and reported under "weak_scaling".

The timed region (exactly K steps, barrier + synchronize on both sides, max over ranks) is repeated --reps times
back to back on the continuing closed loop; `value`/`ms_per_step` are the MEDIAN repetition.

Parity gate: before the line is printed, a spread sample of the instances this process just timed is checked against
the oracle (oracle/liboracle.so, the checker — never the thing measured): free-running over the warm-up ticks
(<= 100 ticks: 1e-9 on u and x), and — from the controller/plant state left by the timed
region — one teacher-forced tick at 1e-9 followed by a full fused launch of 10 free-running ticks at 1e-6 (the
closed loop amplifies rounding differences; see the comment at the check).
On mismatch the script exits without a value.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: vector fp64 (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
MODEL, MODEL_ID, DV, KMAX = "pendulum", 0, 50, 10
DIM_X, DIM_U, DIM_P = 4, 3, 2
GLOBAL_BATCH = 4096
# rocprofv3 PMC summary of this same command (tools/profile_bench.sh + tools/summarise_profile.py), newest first
ISSUE_MODELS = [os.path.join(ROOT, "profiles", n) for n in ("r03_issue_model.json", "r02_issue_model.json")]
INSTRUCTION_COUNTERS = os.path.join(ROOT, "profiles", "r04_wg_instruction_counters.json")
KERNEL_OF_VARIANT = {1: "tick_lane_kernel", 2: "tick_wg_kernel", 3: "tick_wg_kernel", 4: "tick_wave_kernel"}


def algorithmic_flops(k, L=DIM_U * DV):
    """SURVEY.md §8(d), secondary (VALU) view: fp64 operations of one instance-tick, transcendentals not counted:
    (3+k) dv (phi + 16) + (3+k) 4L + 4L k(k+1)/2 + 3Lk + (k+4) 2L with phi_pendulum = 95."""
    return (3 + k) * DV * (95 + 16) + (3 + k) * 4 * L + 4 * L * k * (k + 1) // 2 + 3 * L * k + (k + 4) * 2 * L


def library_hash():
    """sha256 (16 hex digits) of the libcgmres_hip.so this process loads: profiles carry the hash of the library they
    were taken with (tools/summarise_profile.py), and a counter figure is only quoted next to a timing of the same build."""
    import hashlib
    import cgmres_cpp_amd as cg
    try:
        return hashlib.sha256(open(cg.lib_path(), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def algorithmic_bytes(k, L=DIM_U * DV, scalar=8):
    """HBM bytes one instance-tick must move when every length-L operand of a fused
    vector op is one transfer: S*[L*(8 + 6k + k(k-1)/2) + (3+k)*(dim_p*(dv+1) + dim_x) + dim_u]."""
    return scalar * (L * (8 + 6 * k + k * (k - 1) // 2) + (3 + k) * (DIM_P * (DV + 1) + DIM_X) + DIM_U)


def cpu_baseline(batch, tol, warm, seconds_budget, n_all=None):
    """The oracle (CPU restatement, bit-exact with the reference) on this host's cores: same seeded inputs, closed
    loop.  Two legs: all hardware threads available to the process on the whole batch (or, for a long
    warm-up, on its first `n_all` instances — instances are independent, the rate per instance is what is measured),
    and ONE thread on the first 256 instances of the same batch."""
    from oracle import orc
    if not os.path.exists(orc.ORACLE_SO):
        orc.build(ref=False)
    threads = len(os.sched_getaffinity(0))
    x0, u0, p = orc.batch_scenario(orc.PENDULUM, batch)
    batch = min(batch, n_all or batch)

    def leg(n_inst, nthreads, budget):
        ctrls = []
        for i in range(n_inst):
            c = orc.Controller(orc.PENDULUM, DV, KMAX, tol)
            orc.start_controller(c, x0[i], u0[i], p[i])
            ctrls.append(c)
        secs_w, _, x = orc.run_closed_loop(ctrls, x0[:n_inst], warm, nthreads)
        # size the timed part from the warm-up rate so the leg stays within its budget
        ticks = int(max(5, min(200, budget / max(secs_w / warm, 1e-6))))
        secs, _, _ = orc.run_closed_loop(ctrls, x, ticks, nthreads)
        ks = np.array([c.last_solve()[0] for c in ctrls])
        return n_inst * ticks / secs, ticks, float(ks.mean())

    v_all, t_all, k_all = leg(batch, threads, seconds_budget)
    n1 = min(256 if warm <= 100 else 16, batch)
    v_one, t_one, _ = leg(n1, 1, seconds_budget)
    return {"value": v_all, "unit": "control steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle/liboracle.so (CPU restatement, bit-exact vs reference), {batch} controllers x "
                      f"{t_all} closed-loop ticks after {warm} warm-up ticks, tol={tol:g}, {threads} std::threads, "
                      f"mean Arnoldi iterations last tick {k_all:.2f}",
            "one_thread": {"value": v_one, "unit": "control steps/s", "cores": 1,
                           "sample": f"first {n1} controllers of the same batch x {t_one} ticks after {warm} "
                                     f"warm-up ticks, 1 thread"}}


def committed_traffic(kernel_variant, ticks_per_launch, batch):
    """HBM bytes per launch from the COMMITTED rocprofv3 PMC passes of this command at this per-GPU batch (not measured
    in this run): profiles/r<NN>_(wg|wave)_bench[_B<batch>]_pmc.json, newest round first.  Only a profile taken with the
    very library this process times is quoted (`library_sha256_16` in the profile): returns (bytes, path, why_not)."""
    import glob
    import re
    if kernel_variant not in (2, 3, 4):
        return None, None, "no committed counter profile for this mapping"
    cands = []
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_bench*_pmc.json")):
        m = re.match(r"r(\d+)_(?:wg|wave)_bench(?:_B(\d+))?_pmc\.json$", os.path.basename(path))
        if m and int(m.group(2) or GLOBAL_BATCH) == batch:
            cands.append((int(m.group(1)), path))
    mine = library_hash()
    why = "no committed counter profile of this command at this batch"
    for _, path in sorted(cands, reverse=True):
        try:
            s = json.load(open(path))
        except (OSError, ValueError):
            continue
        kern = s.get("kernel", "")
        if KERNEL_OF_VARIANT[kernel_variant] not in kern or s.get("ticks_per_launch", 1) != ticks_per_launch:
            continue
        if kernel_variant in (2, 3):
            lean = "true, " in kern.split("16, 10,")[-1][:8]  # <..., 16, 10, LEAN, PAR>
            if lean != (kernel_variant == 3):
                continue
        if s.get("library_sha256_16") != mine:
            why = f"{os.path.relpath(path, ROOT)} was taken with another build of the library"
            continue
        return s["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), None
    return None, None, why


from tests.parity_gate import OracleSample, ParityError, gate_continuation, sample_of  # noqa: E402  (the checker)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(script, n, argv, port):
    """The command `python3 <script> --gpus N ...` turns into when it was started WITHOUT torch.distributed.run:
    one rank per GPU of this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def self_launch(script, n, argv):
    """--gpus N > 1 from a plain `python3 bench.py`: start the N ranks as a CHILD process (never an exec: this runs
    before anything has touched torch or HIP, and the child is a fresh interpreter), relay its stdout — rank 0's one
    JSON line — and hand back its return code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = launcher_command(script, n, argv, _free_port())
    if os.environ.get("CGMRES_BENCH_PRINT_LAUNCH"):  # tests: show the child command instead of running it
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    sys.stderr.write("[bench] --gpus %d without WORLD_SIZE: launching %s\n" % (n, " ".join(cmd)))
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed K-step region (median reported)")
    ap.add_argument("--batch", type=int, default=GLOBAL_BATCH, help="GLOBAL batch (split over the GPUs)")
    ap.add_argument("--tol", type=float, default=0.0,
                    help="0 = fixed-k mode (always k_max Arnoldi iterations, deterministic work; headline); "
                         "1e-6 = the reference's early-exit mode")
    ap.add_argument("--variant", type=int, default=0,
                    help="kernel mapping: 0 = the library's choice, 1 lane, 2 wg, 3 wg-lean (two workgroups per CU)")
    ap.add_argument("--flags", type=int, default=0, help="cgmres_hip_config.flags (A/B measurements; 0 = library defaults)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="wall budget of each timed CPU-baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-mode", action="store_true", help="skip the secondary tol=1e-6 measurements")
    ap.add_argument("--ref-warmup", type=int, default=5500,
                    help="warm-up ticks of the reference-mode (tol=1e-6) leg: long enough for the Arnoldi counts of the "
                         "seeded batch to spread (all 10 until tick ~4000, 5..10 around 5500: gmres.hpp:93-95)")
    ap.add_argument("--binning-batch", type=int, default=16384,
                    help="global batch of the early-exit placement measurement (needs more workgroups than the GPU "
                         "holds at once; 0 = skip)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the secondary weak-scaling measurement")
    ap.add_argument("--check-sample", type=int, default=48, help="instances per rank checked against the oracle")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo = rehearsal of the N > 1 path on a box with fewer "
                         "GPUs than ranks (ranks share devices, collectives go through host tensors)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the process group, rank 0 prints the world it sees (the "
                         "CPU rehearsal of the N > 1 entry; needs no GPU)")
    args = ap.parse_args()

    # N > 1 started like N = 1 (no torch.distributed.run around it): become the launcher.  Decided BEFORE torch is
    # imported, so the parent never initialises the GPU.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(os.path.abspath(__file__), args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    import cgmres_cpp_amd as cg
    from cgmres_cpp_amd import scenarios
    from cgmres_cpp_amd.sharding import scatter_rows, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch with torch.distributed.run "
                 f"--nproc-per-node {args.gpus} (or plain `python3 bench.py --gpus {args.gpus}`, which does that itself)")
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
            seen = [None] * world
            dist.all_gather_object(seen, (rank, local))
            dist.barrier()
        else:
            seen = [(0, 0)]
        if rank == 0:
            print(json.dumps({"launch_check": True, "world_size": dist.get_world_size() if world > 1 else 1,
                              "ranks": sorted(r for r, _ in seen), "gpus": args.gpus}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU path")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collective buffers live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus

    stream = torch.cuda.current_stream().cuda_stream
    resolved = {}

    def shard_inputs(n_global):
        """rank 0 draws the whole job (seeded); the shards go out over RCCL — the only exchange of the job"""
        full = scenarios.batch(MODEL, n_global) if rank == 0 else (None, None, None)
        parts = [scatter_rows(full[k], n_global, w, world, rank, cdev, dist) for k, w in enumerate((DIM_X, DIM_U, DIM_P))]
        torch.cuda.synchronize()
        lo, hi = shard_bounds(n_global, world, rank)
        out = [t.cpu().numpy() for t in parts]
        assert out[0].shape == (hi - lo, DIM_X)
        return out

    def oracle_diverges(tol, x0, u0, p, ticks):
        """True when the oracle's free-running closed loop of this instance is non-finite (or beyond |u| = 1e6) within
        `ticks` ticks: the divergence belongs to the algorithm on this trajectory, not to the device code."""
        from oracle import orc
        c = orc.Controller(orc.PENDULUM, DV, KMAX, tol)
        orc.start_controller(c, x0, u0, p)
        us, _, _, _ = orc.closed_loop(c, x0, ticks)
        return bool((~np.isfinite(us)).any() or np.nanmax(np.abs(us)) > 1e6)

    def measure(inputs, tol, steps, warmup, reps, check, flags=None, fatal=True, also_fixed_k=False):
        """Closed loop of this rank's shard: warm-up, then `reps` timed regions of exactly `steps` ticks.
        A failed parity gate ends the script (the headline: no number without parity) or, for a SECONDARY leg
        (fatal=False), comes back as {"error": ...} so that the leg is reported as failed without a number.
        also_fixed_k: a second controller batch with tol = 0 takes over the state the warm-up left (t, U, dUdt, x) and
        is timed for `steps` ticks from there — the fixed-k figure for the SAME controller/plant state as the early-exit
        leg that follows (its result is not fed back)."""
        x0_h, u0_h, p_h = inputs
        B = len(x0_h)
        ctrl = cg.CgmresBatch(MODEL, batch=B, dv=DV, k_max=KMAX, tol=tol, device=local, stream=stream,
                              variant=args.variant, flags=args.flags if flags is None else flags)
        resolved["variant"] = ctrl.variant
        resolved["variant_name"] = ctrl.variant_name
        ctrl.set_ptau_repeat(p_h)
        ctrl.init_u0(u0_h)
        ctrl.init_u0_newton(u0_h, x0_h, p_h, 10)
        x = torch.from_numpy(x0_h).to(dev)
        u = torch.zeros(B, DIM_U, dtype=torch.float64, device=dev)
        chk = OracleSample(MODEL, DV, KMAX, tol, x0_h, u0_h, p_h, check) if check else None
        parity = {}
        ctrl.closed_loop_device(x, u, warmup)
        torch.cuda.synchronize()
        err = None
        try:
            if chk and 0 < warmup <= 100:
                chk.advance(warmup)
                parity["warmup_free_running_max_err"] = chk.compare(
                    f"free-running warm-up ({warmup} ticks)", x.cpu().numpy(), u.cpu().numpy(), ctrl.get_status()[0],
                    1e-9, tol == 0.0)
                parity["arnoldi_count_flips"] = chk.flips
        except ParityError as e:
            err = str(e)
        fixed_k_ms = None
        if also_fixed_k and err is None:
            c0 = cg.CgmresBatch(MODEL, batch=B, dv=DV, k_max=KMAX, tol=0.0, device=local, stream=stream,
                                variant=args.variant, flags=args.flags if flags is None else flags)
            c0.set_ptau_repeat(p_h)
            t_now, U_now, d_now = ctrl.get_state()
            best = None
            for _ in range(max(1, reps)):
                c0.set_state(t_now, U_now, d_now)
                x0c, u0c = x.clone(), u.clone()
                torch.cuda.synchronize()
                c0.timer_start()
                c0.closed_loop_device(x0c, u0c, steps)
                ms = c0.timer_stop()
                best = ms if best is None else min(best, ms)
            fixed_k_ms = {"ms_per_step": best / steps, "variant_name": c0.variant_name}
            c0.close()
        walls, kernels, own = [], [], []
        for _ in range(reps):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctrl.timer_start()
            ctrl.closed_loop_device(x, u, steps)
            kernel_ms = ctrl.timer_stop()  # HIP events on the launch stream, around exactly the K-step launches
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            own.append(kernel_ms)
            tt = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=cdev)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            walls.append(tt[0].item()), kernels.append(tt[1].item())
        n_ax, why = ctrl.get_status()
        # Non-finite instances.  C/GMRES itself can diverge: on this seeded scenario the REFERENCE's own closed loop of
        # instance 3599 runs to |u| ~ 1e12 and NaN at tick 4695 in early-exit mode (the slack input crosses zero; with
        # tol = 0 it survives), and the device loop does the same a few dozen ticks apart.  Such an instance carries the
        # status CGMRES_HIP_EXIT_NONFINITE; it is accepted only when it is rare AND the oracle, free-running from the
        # same start, has diverged by the same tick as well — anything else is a failure.
        okf = (torch.isfinite(u).all(dim=1) & torch.isfinite(x).all(dim=1)).cpu().numpy()
        diverged = [int(i) for i in np.nonzero(~okf)[0]]
        finite = True
        if diverged:
            ticks_run = warmup + reps * steps
            finite = len(diverged) <= max(4, B // 1000) and all(why[i] == cg.EXIT_NONFINITE for i in diverged) and \
                all(oracle_diverges(tol, x0_h[i], u0_h[i], p_h[i], ticks_run + 200) for i in diverged)
            parity["diverged_instances_confirmed_on_the_oracle"] = diverged if finite else []
            n_ax = np.where(okf, n_ax, KMAX)  # (byte accounting: count them as full solves)
        try:
            if chk and err is None:
                # The state the timed region left behind, continued on both sides (teacher-forced from the device's own
                # state): ONE tick at the single-tick tolerance of SURVEY.md §8(c), 1e-9, then a full fused launch of 10
                # ticks across a launch boundary.  The closed loop amplifies rounding differences between any two
                # builds (measured on this scenario around tick 320, where |u| reaches its bound: x1e5 within 11 ticks
                # for ~1 % of the instances while every single tick agrees to 2e-14 — tools/accuracy_scan.py), so the
                # free-running 10 ticks are held to 1e-6: far below anything a hand-over bug produces, above the chaos.
                flips0 = parity.get("arnoldi_count_flips", 0)
                parity.update(gate_continuation(ctrl, chk, x, u, torch.cuda.synchronize, 1e-9, 1e-6, tol == 0.0))
                parity["arnoldi_count_flips"] += flips0
        except ParityError as e:
            err = str(e)
        ctrl.close()
        bad = torch.tensor([0 if (err is None and finite) else 1], dtype=torch.int32, device=cdev)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if err is not None or not finite:
            sys.stderr.write(f"[rank {rank}] PARITY FAILURE: {err or 'non-finite output'}\n")
        if bad.item():
            if not fatal:
                return {"error": err or ("non-finite output" if not finite else "parity failure on another rank")}
            if world > 1:
                dist.destroy_process_group()
            sys.exit("parity check against the oracle failed: refusing to report a number")
        med = statistics.median_low(walls)
        # every rank's own HIP-event time of the median repetition (the line's time is the max over ranks)
        mine = torch.tensor([own[walls.index(med)]], dtype=torch.float64, device=cdev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)] if world > 1 else [mine]
        if world > 1:
            dist.all_gather(per_rank, mine)
        return {"B": B, "wall": med, "kernel_ms": kernels[walls.index(med)], "walls": walls, "n_ax": n_ax,
                "parity": parity, "rank_kernel_ms": [t.item() for t in per_rank], "variant_name": resolved["variant_name"],
                "fixed_k_same_state": fixed_k_ms}

    n_check = args.check_sample if world == 1 else max(8, args.check_sample // world)
    inputs = shard_inputs(args.batch)
    m = measure(inputs, args.tol, args.steps, args.warmup, args.reps, n_check)
    head = dict(resolved)  # (the mapping of the HEADLINE leg: the secondary legs below resolve their own)
    B = m["B"]
    value = args.batch * args.steps / m["wall"]
    ms_per_step = m["wall"] * 1e3 / args.steps
    # one launch of the wg mapping = TICKS_PER_LAUNCH consecutive ticks of the batch (the lane mapping: one tick)
    tpl = cg.TICKS_PER_LAUNCH if head["variant"] in (2, 3, 4) else 1  # (3 = wg-lean: same kernel, half the LDS; 4 = wave)
    n_launches = -(-args.steps // tpl)
    launch_ms = m["kernel_ms"] / n_launches
    bytes_per_tick = float(sum(algorithmic_bytes(int(k)) for k in m["n_ax"])) if args.tol > 0 else \
        float(B * algorithmic_bytes(KMAX))
    bytes_per_launch = bytes_per_tick * args.steps / n_launches
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    traffic, traffic_src, traffic_why = committed_traffic(head["variant"], args.steps / n_launches, B) \
        if args.tol == 0.0 else (None, None, "early-exit mode: the profiles are fixed-k")
    flops_per_tick = float(sum(algorithmic_flops(int(k)) for k in m["n_ax"])) if args.tol > 0 else \
        float(B * algorithmic_flops(KMAX))

    def k_hist(n_ax):
        return [int(v) for v in np.bincount(np.asarray(n_ax, dtype=np.int64), minlength=KMAX + 1)[:KMAX + 1]]

    def early_exit_leg(m2, n_global, warm):
        """The reference's own mode (tol = 1e-6: the Arnoldi loop ends when |rho_e[k+1]| < tol, gmres.hpp:93-95), timed
        where the counts of the batch are SPREAD; bytes = sum of bytes(k_b) over the executed counts."""
        if "error" in m2:
            return {"tol": 1e-6, "warmup_ticks": warm, "global_batch": n_global, "value": None,
                    "parity_failure": m2["error"], "note": "this leg failed its oracle gate: no number is reported for it"}
        by = float(sum(algorithmic_bytes(int(k)) for k in m2["n_ax"])) * world  # (this rank's shard x ranks)
        return {"tol": 1e-6, "warmup_ticks": warm, "global_batch": n_global, "variant_name": m2["variant_name"],
                "value": n_global * args.steps / m2["wall"], "ms_per_step": m2["wall"] * 1e3 / args.steps,
                "fixed_k_from_the_same_state": m2.get("fixed_k_same_state"),
                "mean_arnoldi_last_tick": float(np.mean(m2["n_ax"])), "k_histogram_last_tick_rank0": k_hist(m2["n_ax"]),
                "algorithmic_frac_of_peak": by / (m2["wall"] / args.steps) / 1e9 / HBM_PEAK_GBS / world,
                "parity": m2["parity"]}

    def soft(leg):
        """A secondary leg must never cost the headline its line: anything it raises is reported inside the leg."""
        try:
            return leg()
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001 (reported, not swallowed)
            if world > 1:
                raise  # (ranks must stay in lock-step: a rank-local exception cannot be papered over)
            return {"error": f"{type(e).__name__}: {e}"}

    ref_mode = binning = None
    if not args.no_ref_mode and args.tol == 0.0:
        reps2 = max(1, min(args.reps, 3))
        ref_mode = early_exit_leg(soft(lambda: measure(inputs, 1e-6, args.steps, args.ref_warmup, reps2, n_check,
                                                       fatal=False, also_fixed_k=True)), args.batch, args.ref_warmup)
        ref_mode.setdefault("note", "")
        ref_mode["note"] += ("wg mapping: every workgroup is resident (one per CU) and a launch lasts as long as its "
                            "slowest workgroup, so WHERE the instances sit cannot shorten it (placement by count is "
                            "measured below on a batch that needs several rounds of workgroups); wave mapping: every "
                            "controller is its own wavefront and simply leaves its Arnoldi loop")
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # the reference's own mode on the host cores, from the same warmed-up state (one instance per thread: the
            # long warm-up is what bounds the sample)
            ref_mode["cpu_baseline"] = soft(lambda: cpu_baseline(args.batch, 1e-6, args.ref_warmup, args.cpu_seconds,
                                                                 n_all=len(os.sched_getaffinity(0))))
        if args.binning_batch and world == 1:
            big = shard_inputs(args.binning_batch)
            legs = {}
            for name, fl in (("binned_by_last_count", args.flags & ~cg.FLAG_NO_BINNING),
                             ("caller_order", args.flags | cg.FLAG_NO_BINNING)):
                legs[name] = early_exit_leg(soft(lambda fl=fl: measure(big, 1e-6, args.steps, args.ref_warmup, reps2, 8, fl,
                                                                       fatal=False)), args.binning_batch, args.ref_warmup)
            both = legs["binned_by_last_count"]["value"] and legs["caller_order"]["value"]
            binning = dict(legs, speedup=(legs["binned_by_last_count"]["value"] / legs["caller_order"]["value"]) if both else None)
    weak = None
    if world > 1 and not args.no_weak:
        mw = measure(shard_inputs(args.batch * world), args.tol, args.steps, args.warmup, max(1, min(args.reps, 3)), 0,
                     fatal=False)
        weak = {"scaling": "weak", "global_batch": args.batch * world, "value": None, "parity_failure": mw["error"]} \
            if "error" in mw else \
            {"scaling": "weak", "batch_per_gpu": mw["B"], "global_batch": args.batch * world,
             "value": args.batch * world * args.steps / mw["wall"], "ms_per_step": mw["wall"] * 1e3 / args.steps}

    kernel_name = "tick_lane_kernel" if head["variant"] == 1 else \
        f"{KERNEL_OF_VARIANT[head['variant']]} [{head['variant_name']}]"
    flops_per_launch = flops_per_tick * args.steps / n_launches
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source_committed_profile": traffic_src,
                "traffic_unavailable_because": traffic_why,
                "measured_GBps": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None,
                # what the counters say HBM really moved, and the fp64 VALU view of the same launch: neither binds
                "hbm_measured_frac": (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "valu_frac": flops_per_launch / (launch_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "algorithmic_fp64_flops_per_launch": flops_per_launch, "valu_peak_TFLOPs": FP64_VALU_PEAK_TFLOPS,
                "bound_observed": "issue",
                "bound_observed_note": "at this batch a tick is a chain of instruction issues (one instruction per ~4.4 "
                                       "cycles and wave, DESIGN.md 4.1/4.5): `frac` is the metric's algorithmic-byte model / "
                                       "time, `hbm_measured_frac` what the counters say HBM moved, `valu_frac` the fp64 view",
                "library_sha256_16": library_hash(),
                "kernel": f"{kernel_name} ({tpl} control step(s) of the batch per launch)", "launch_ms": launch_ms,
                "ticks_per_launch": args.steps / n_launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                "rank": 0, "instances_per_launch": B}
    if "row-newton" in head["variant_name"]:
        # the row-parallel kernel has no serial stage loop to model: what the SQ counters say it issues (own --pmc pass of
        # this command, tools/profile_round4.sh), quoted only next to a timing of the same build
        try:
            ic = json.load(open(INSTRUCTION_COUNTERS))
            if ic.get("library_sha256_16") == library_hash():
                roofline["instruction_counters"] = {k: ic[k] for k in ("per_controller_and_tick", "cycles_per_tick",
                                                                       "valu_issue_utilisation_per_simd", "note")}
                roofline["instruction_counters"]["source"] = os.path.relpath(INSTRUCTION_COUNTERS, ROOT)
        except (OSError, ValueError, KeyError):
            pass
    else:
        for path in ISSUE_MODELS:
            try:
                roofline["issue_slot_model"] = json.load(open(path))
                break
            except OSError:
                pass
    out = {
        "metric": "C/GMRES control steps/sec, batch=4096 N=50 kmax=10; HBM GB/s vs roofline",
        "value": value, "unit": "control steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "reps": args.reps,
        "rep_ms_per_step": [w * 1e3 / args.steps for w in m["walls"]],
        "world_size": dist.get_world_size() if world > 1 else 1,
        "rank_ms_per_step": [t / args.steps for t in m["rank_kernel_ms"]],
        "config": {"workload": "arm_type_inverted_pendulum controllers, closed loop with on-device Euler plant",
                   "global_batch": args.batch, "batch_per_gpu": B, "N": DV, "kmax": KMAX, "tol": args.tol,
                   "mode": "fixed-k (tol=0, every instance runs k_max Arnoldi iterations)" if args.tol == 0
                   else "reference early-exit", "variant": head["variant"],
                   "variant_name": head["variant_name"],
                   "parallelism": f"batch-shard x{world} (fixed global batch)" +
                                  ("" if args.backend == "nccl" else " [gloo rehearsal: ranks share GPUs]"),
                   "inputs": "splitmix64(12345) perturbed x0/targets, Newton-initialised U0 (SURVEY.md §8d)"},
        "roofline": roofline,
        "parity": m["parity"],
    }
    if ref_mode:
        out["reference_mode"] = ref_mode
    if binning:
        out["early_exit_placement"] = binning
    if weak:
        out["weak_scaling"] = weak
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.batch, args.tol, min(args.warmup, 50) or 5, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
