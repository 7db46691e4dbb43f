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
