"""CPU: the oracle's C++ (the restatement that every parity claim rests on) under AddressSanitizer + UBSan
(the reference ships no sanitizer run; GPU ASan is not available on this pool, so the CPU side is where
memory errors of the restated algorithm can be caught).

  * oracle/oracle_capi.cpp  -> a sanitized liboracle, driven through the same ctypes front end (oracle/orc.py) in a child
    python with libasan preloaded: closed loops of all three models at the golden sizes (fp64 and fp32), white-box
    F/Ax/gmres calls, the threaded batch runner — outputs must still equal the committed reference vectors bit for bit;
  * tests/user_models/vdp_oracle.cpp (the generic controller on a user model) -> sanitized executable, output must equal
    tests/golden/user_vdp_closed_loop.txt.
Any sanitizer report aborts the child (-fno-sanitize-recover, halt_on_error) and fails the test."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _libasan():
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    p = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not p or not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("libasan not installed")
    return os.path.realpath(p)


CHILD = textwrap.dedent('''
    import glob, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["REPO_ROOT"])
    from oracle import orc
    orc.ORACLE_SO = os.environ["SAN_ORACLE_SO"]          # the sanitized build, same C API
    sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
    from conftest import golden_files, load_golden
    n = 0
    for path in golden_files():
        g = load_golden(path)
        c = g["_case"]
        ctrl = orc.Controller(c["model"], c["dv"], c["kmax"], c["tol"], c["dtype"])
        un = orc.start_controller(ctrl, g["x0"], g["u0_guess"], g["p"])
        ticks = min(len(g["loop_u"]), 60)
        us, xs, ks, _ = orc.closed_loop(ctrl, g["x0"], ticks)
        if c["dtype"] == "f64":
            assert np.array_equal(un, g["u0_newton"]), path
            assert np.array_equal(us, g["loop_u"][:ticks]) and np.array_equal(ks, g["loop_k"][:ticks]), path
        else:
            np.testing.assert_allclose(us[:20], g["loop_u"][:20], rtol=5e-3, atol=5e-3)
        # white-box entry points on a teacher-forced record (exercises v_mat / h_mat / g_vec indexing at k = k_max)
        tick = g["_ticks"][-1]
        p = f"tick{tick}_"
        w = orc.Controller(c["model"], c["dv"], c["kmax"], c["tol"], c["dtype"])
        if c["dim_p"]:
            w.set_ptau(g["ptau"])
        w.set_state(g[p + "t"][0], g[p + "U"], g[p + "dUdt"])
        w.F(g[p + "U"], g[p + "x"], g[p + "t"][0]); w.prepare(g[p + "x"]); w.Ax(g[p + "Ax_v"])
        sol = w.gmres(g[p + "dUdt"], g[p + "b"])
        w.krylov()
        if c["dtype"] == "f64":
            assert np.array_equal(sol, g[p + "gmres_x"]), path
        n += 1
    # the threaded batch driver of the CPU baseline (4 threads, ragged split)
    x0, u0, p = orc.batch_scenario(orc.PENDULUM, 7)
    cs = []
    for i in range(7):
        cc = orc.Controller(orc.PENDULUM, 50, 10, 0.0)
        orc.start_controller(cc, x0[i], u0[i], p[i]); cs.append(cc)
    orc.run_closed_loop(cs, x0, 5, 4)
    print("sanitized oracle ok:", n, "golden cases")
''')


def test_oracle_capi_under_asan_ubsan(tmp_path):
    asan = _libasan()
    so = tmp_path / "liboracle_san.so"
    subprocess.run(["g++", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared"] + SAN +
                   ["-o", str(so), os.path.join(ROOT, "oracle", "oracle_capi.cpp")], check=True)
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, REPO_ROOT=ROOT, SAN_ORACLE_SO=str(so), LD_PRELOAD=asan, **SAN_ENV)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "sanitized oracle ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]


def test_generic_oracle_controller_on_a_user_model_under_asan_ubsan(tmp_path):
    _libasan()
    exe = tmp_path / "vdp_oracle_san"
    subprocess.run(["g++", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}"] + SAN +
                   [os.path.join(ROOT, "tests", "user_models", "vdp_oracle.cpp"), "-o", str(exe)], check=True)
    golden = os.path.join(ROOT, "tests", "golden", "user_vdp_closed_loop.txt")
    want = open(golden).read()
    last = want.strip().splitlines()[-1].split()
    B, ticks = int(last[0]) + 1, int(last[1]) + 1
    r = subprocess.run([str(exe), str(B), str(ticks)], env=dict(os.environ, **SAN_ENV), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-6000:]
    assert r.stdout == want
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
