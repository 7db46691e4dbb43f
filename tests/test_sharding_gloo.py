"""CPU, world_size 2 over gloo: the N>1 path of bench.py — rank 0 draws the seeded job, shards are scattered,
every rank advances its own controllers, controls are gathered — gives exactly what one process computes for the
whole batch.  (No GPU here: the per-shard compute is the oracle; the sharding/collective logic is what is tested.)"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    from cgmres_cpp_amd.sharding import shard_bounds
    for n in (1, 7, 8, 4096, 4099, 8192):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_c_abi_shard_bounds_is_the_same_rule():
    """cgmres_hip_shard_bounds (what include/cgmres_batch.hpp's CgmresBatchSharded cuts its host vectors by) ==
    cgmres_cpp_amd.sharding.shard_bounds (what bench.py's ranks use); pure host arithmetic, no GPU."""
    import ctypes as C
    import cgmres_cpp_amd as cg
    from cgmres_cpp_amd.sharding import shard_bounds
    L = cg.load()
    L.cgmres_hip_shard_bounds.argtypes = [C.c_int32] * 3 + [C.POINTER(C.c_int32)] * 2
    for n in (2, 7, 8, 4096, 4099, 8192):
        for world in (1, 2, 3, 4, 8):
            if n < world:
                continue
            for r in range(world):
                lo, hi = C.c_int32(), C.c_int32()
                assert L.cgmres_hip_shard_bounds(n, world, r, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_bounds(n, world, r)
    lo, hi = C.c_int32(), C.c_int32()
    assert L.cgmres_hip_shard_bounds(10, 2, 2, C.byref(lo), C.byref(hi)) == -1   # CGMRES_HIP_EINVAL
    assert L.cgmres_hip_shard_bounds(10, 0, 0, C.byref(lo), C.byref(hi)) == -1


WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, os.environ["REPO_ROOT"])
    import numpy as np, torch, torch.distributed as dist
    from cgmres_cpp_amd import scenarios
    from cgmres_cpp_amd.sharding import scatter_rows, gather_rows, shard_bounds
    from oracle import orc
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, TICKS, DV, KM = 13, 4, 8, 3          # uneven split on purpose (7 + 6)
    full = scenarios.batch("pendulum", N) if rank == 0 else (None, None, None)
    dev = torch.device("cpu")
    x = scatter_rows(full[0], N, 4, world, rank, dev, dist).numpy()
    u0 = scatter_rows(full[1], N, 3, world, rank, dev, dist).numpy()
    p = scatter_rows(full[2], N, 2, world, rank, dev, dist).numpy()
    lo, hi = shard_bounds(N, world, rank)
    assert x.shape == (hi - lo, 4)
    us = []
    for i in range(hi - lo):
        c = orc.Controller(orc.PENDULUM, DV, KM)
        orc.start_controller(c, x[i], u0[i], p[i])
        u, _, _, _ = orc.closed_loop(c, x[i], TICKS)
        us.append(u[-1])
    got = gather_rows(torch.tensor(np.array(us)), N, world, rank, dist)
    if rank == 0:
        np.save(os.environ["OUT_NPY"], got)
    dist.barrier()
    dist.destroy_process_group()
''')


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_job_equals_single_process(tmp_path, orc):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "u.npy"
    env = dict(os.environ, REPO_ROOT=ROOT, OUT_NPY=str(out), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    from cgmres_cpp_amd import scenarios
    x0, u0, p = scenarios.batch("pendulum", 13)
    want = []
    for i in range(13):
        c = orc.Controller(orc.PENDULUM, 8, 3)
        orc.start_controller(c, x0[i], u0[i], p[i])
        u, _, _, _ = orc.closed_loop(c, x0[i], 4)
        want.append(u[-1])
    assert np.array_equal(got, np.array(want))


MIXED_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.environ["REPO_ROOT"])
    import numpy as np, torch, torch.distributed as dist
    from cgmres_cpp_amd import scenarios
    from cgmres_cpp_amd.sharding import gather_rows, shard_bounds
    from oracle import orc
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    TICKS, DV, KM = 3, 8, 3
    out = {}
    # BASELINE configs[3] shape: EACH model's sub-batch is split over all ranks (tools/bench_configs.py --config 4)
    for name, model, n in (("msd", orc.MSD, 5), ("pendulum", orc.PENDULUM, 7)):
        x0, u0, p = scenarios.batch(name, n)       # seeded: every rank draws the job and keeps its slice
        lo, hi = shard_bounds(n, world, rank)
        us = []
        for i in range(lo, hi):
            c = orc.Controller(model, DV, KM)
            orc.start_controller(c, x0[i], u0[i], p[i])
            u, _, _, _ = orc.closed_loop(c, x0[i], TICKS)
            us.append(u[-1])
        got = gather_rows(torch.tensor(np.array(us)), n, world, rank, dist)
        if rank == 0:
            out[name] = got
    if rank == 0:
        np.savez(os.environ["OUT_NPY"], **out)
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_rank_mixed_batch_equals_single_process(tmp_path, orc):
    """multiple_controller's heterogeneous batch over 2 ranks: each model's sub-batch sharded over both ranks (uneven
    splits 3+2 and 4+3), gathered per model == what one process computes."""
    script = tmp_path / "worker_mixed.py"
    script.write_text(MIXED_WORKER)
    out = tmp_path / "u.npz"
    env = dict(os.environ, REPO_ROOT=ROOT, OUT_NPY=str(out), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    from cgmres_cpp_amd import scenarios
    for name, model, n in (("msd", orc.MSD, 5), ("pendulum", orc.PENDULUM, 7)):
        x0, u0, p = scenarios.batch(name, n)
        want = []
        for i in range(n):
            c = orc.Controller(model, 8, 3)
            orc.start_controller(c, x0[i], u0[i], p[i])
            u, _, _, _ = orc.closed_loop(c, x0[i], 3)
            want.append(u[-1])
        assert np.array_equal(got[name], np.array(want)), name
