"""CPU: the bookkeeping of bench.py — algorithmic bytes of SURVEY.md §8(d), the spread sample of the parity gate —
and of cgmres_cpp_amd.multi (when MultipleController asks for the CU-sharing mapping)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_the_survey():
    b = _bench()
    # pendulum N = 50, k = 10, fp64: 113*150 + 13*106 + 3 = 18 331 scalars = 146.6 KB
    assert b.algorithmic_bytes(10) == 8 * 18331
    assert b.algorithmic_bytes(10) * 4096 == 600670208                     # "600.7 MB per batch tick"
    assert b.algorithmic_bytes(0) == 8 * (8 * 150 + 3 * 106 + 3)           # residual-exit tick: preamble + update only
    # the secondary configurations (tools/bench_configs.py), SURVEY.md §8(d): cfg 5 fp32 long horizon 100 141 scalars,
    # cfg 3 semiactive 16 979, cfg 4 MSD 35 284, cfg 1 MSD N = 20 k = 5 6 134
    spec = importlib.util.spec_from_file_location("bench_cfg_mod", os.path.join(ROOT, "tools", "bench_configs.py"))
    c = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(c)
    assert c.alg_bytes(0, 100, 20, 4) == 4 * 100141
    assert c.alg_bytes(2, 50, 10, 8) == 8 * 16979
    assert c.alg_bytes(1, 50, 10, 8) == 8 * 35284
    assert c.alg_bytes(1, 20, 5, 8) == 8 * 6134
    assert c.alg_bytes(0, 50, 10, 8) == b.algorithmic_bytes(10)


def test_parity_sample_is_spread_and_covers_the_edges():
    b = _bench()
    for B, n in ((4096, 48), (512, 8), (33, 48), (1, 48)):
        s = b.sample_of(B, n)
        assert s == sorted(set(s)) and s[0] == 0 and s[-1] == B - 1 and all(0 <= i < B for i in s)
        assert len(s) <= n + 3
        if B > 16:
            assert 15 in s and 16 in s  # both sides of a workgroup edge


def test_multiple_controller_shares_cus_only_when_needed(monkeypatch):
    import cgmres_cpp_amd.multi as multi
    asked = []

    class Fake:
        def __init__(self, **kw):
            asked.append(kw.get("variant", 0))

    monkeypatch.setattr(multi, "CgmresBatch", Fake)
    monkeypatch.setattr(multi, "_cu_count", lambda device: 256)
    multi.MultipleController([dict(model="msd", batch=4096), dict(model="pendulum", batch=4096)])
    assert asked == [3, 3]                      # 512 workgroups for 256 CUs: the lean mapping
    asked.clear()
    multi.MultipleController([dict(model="msd", batch=512), dict(model="pendulum", batch=512)])
    assert asked == [0, 0]                      # an 8-GPU shard: every workgroup gets its own CU
    asked.clear()
    multi.MultipleController([dict(model="msd", batch=8192)])
    assert asked == [0]                         # a single member: the library's own choice


# ---- the N > 1 entry: `python3 bench.py --gpus N` from a plain invocation starts its own torch.distributed.run ----
def _run_plain(script, *argv, env=None, timeout=300):
    import subprocess
    import sys
    e = dict(os.environ, **(env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + list(argv), env=e, capture_output=True,
                          text=True, timeout=timeout, cwd="/tmp")


def test_bench_gpus_n_launches_its_own_ranks_command_line():
    import json
    import sys
    for script, extra in (("bench.py", ["--steps", "20", "--warmup", "10"]),
                          (os.path.join("tools", "bench_configs.py"), ["--config", "4"])):
        r = _run_plain(script, "--gpus", "4", *extra, env={"CGMRES_BENCH_PRINT_LAUNCH": "1"})
        assert r.returncode == 0, r.stderr[-2000:]
        cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
        assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
        assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
        k = cmd.index(os.path.join(ROOT, script))          # the script itself, then its own arguments unchanged
        assert cmd[k + 1:] == ["--gpus", "4"] + extra
        assert "torch" not in r.stderr.lower() or "launching" in r.stderr  # the parent never got as far as importing torch


def test_bench_gpus_2_from_a_plain_invocation_reaches_a_world_of_two():
    """No GPU needed: --launch-check stops after the rendezvous (gloo).  The parent relays rank 0's line and the
    children's return code."""
    import json
    r = _run_plain("bench.py", "--gpus", "2", "--backend", "gloo", "--launch-check")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"launch_check": True, "world_size": 2, "ranks": [0, 1], "gpus": 2}
    # a failing child is reported, not swallowed (no GPU here: the real bench refuses to run)
    r = _run_plain("bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "10")
    assert r.returncode != 0


def test_bench_handles_every_fused_variant_in_its_bookkeeping(monkeypatch):
    b = _bench()
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'head["variant"] in (2, 3, 4)' in src           # wg-lean and wave fuse 10 ticks per launch like wg
    # Committed PMC traffic is looked up per per-GPU batch, per mapping and per LDS plan, and is only quoted when the
    # profile was taken with the very library being timed.
    import json
    prof = json.load(open(os.path.join(ROOT, "profiles", "r03_wg_bench_pmc.json")))
    monkeypatch.setattr(b, "library_hash", lambda: prof.get("library_sha256_16"))
    t, path, why = b.committed_traffic(2, 10, 4096)
    assert t and path.startswith("profiles/") and why is None
    monkeypatch.setattr(b, "library_hash", lambda: "0123456789abcdef")
    t, path, why = b.committed_traffic(2, 10, 4096)
    assert t is None and path is None and "another build" in why
    assert b.committed_traffic(1, 1, 4096)[:2] == (None, None)
    assert b.KERNEL_OF_VARIANT[4] == "tick_wave_kernel"
    # the secondary (VALU) view: SURVEY.md 8(d)'s flop count of one pendulum instance-tick at k = 10
    assert b.algorithmic_flops(10) == 13 * 50 * 111 + 13 * 600 + 4 * 150 * 55 + 4500 + 14 * 300
