"""GPU: the N > 1 entry of bench.py exactly as the driver would start it WITHOUT a launcher — `python3 bench.py
--gpus 2` — on the one-GPU box: the two ranks share the card and talk through gloo (`--backend gloo`, the rehearsal
mode; on a multi-GPU node the same entry runs one rank per GPU over RCCL).  Checks that the parent starts its own
torch.distributed.run child, that both ranks reach the process group, that the shards are parity-gated against the
oracle, and that rank 0's single JSON line comes back through the parent."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_self_launch_on_one_card():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20",
                        "--warmup", "10", "--reps", "2", "--no-ref-mode", "--check-sample", "16"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 4096 and out["config"]["batch_per_gpu"] == 2048
    assert len(out["rank_ms_per_step"]) == 2 and all(t > 0 for t in out["rank_ms_per_step"])
    assert out["value"] > 1e6 and out["roofline"]["frac"] > 0
    assert out["parity"]["instances_checked"] >= 8 and out["parity"]["continuation_1tick_max_err"] <= 1e-9
    assert out["weak_scaling"]["batch_per_gpu"] == 4096
    assert "cpu_baseline" not in out  # (rank 0 at N = 1 only)


def test_bench_configs_gpus_2_self_launch_on_one_card():
    """tools/bench_configs.py started like a single-GPU run with --gpus 2: same self-launch as bench.py; each member's
    sub-batch is split over the ranks and every rank's members pass the oracle gate before the line is printed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "--gpus", "2", "--backend", "gloo",
                        "--config", "4", "--steps", "20", "--warmup", "10", "--tols", "0", "--check-sample", "8"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["finite"]
    assert lines[0]["members"] == ["msd B=4096 dv=50 k=10", "pendulum B=4096 dv=50 k=10"]
    assert len(lines[0]["parity_rank0"]) == 2 and all(g["continuation_1tick_max_err"] <= 1e-9 for g in lines[0]["parity_rank0"])
