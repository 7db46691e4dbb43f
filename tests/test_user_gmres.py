"""The stand-alone solver of the reference's class surface — `class Gmres` with a caller-supplied Ax_func (reference
include/gmres.hpp:8-129) — on the device: csrc/user_operator.hip.h + cgmres_hip_gmres_user, behind the facade
include/gmres.hpp.  The fixtures tests/golden/user_gmres_*.txt are the output of the UNMODIFIED reference solver
(oracle/gmres_ref.cpp) for a symmetric positive definite and a nonsymmetric operator, three (k_max, tol) cases each:
convergence inside the loop, a full Krylov space (k_max >= len: breakdown on an exact solve), and a fixed short one."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import cgmres_cpp_amd as cg
from cgmres_cpp_amd import plugin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = os.path.join(ROOT, "tests", "user_models", "gmres_ops.hpp")
CASES = {"spd": ("SpdTridiagOp", 24, 1, (12, 30, 5)), "convdiff": ("ConvDiffOp", 40, 2, (20, 30, 5)),
         # the vector lengths of the controller's own solves (dim_u*dv = 150 / 300): several elements per lane of the
         # one-wave-per-system solver, 4 instances per (k_max, tol) case
         "convdiff150": ("ConvDiffOp150", 150, 2, (20, 30, 10)), "convdiff300": ("ConvDiffOp300", 300, 2, (20, 30, 10))}
N_INST = {"spd": 12, "convdiff": 12, "convdiff150": 4, "convdiff300": 4}
TOLS = (1e-9, 1e-6, 0.0)


def scenario(name, i):
    cls, L, npar, _ = CASES[name]
    e = np.arange(L)
    p = [0.3 + 0.11 * i] if name == "spd" else [0.4 + 0.07 * i, 0.35 - 0.02 * i]  # (every convdiff* case)
    return np.array(p), np.sin(0.3 * e + 0.5 * i) + 0.1 * e, 0.01 * (e - i)


def fixture(name):
    rows = np.loadtxt(os.path.join(ROOT, "tests", "golden", f"user_gmres_{name}.txt"))
    assert rows.shape == (3 * N_INST[name], 3 + CASES[name][1])
    return rows


def dense(name, p):
    cls, L, npar, _ = CASES[name]
    A = np.zeros((L, L))
    for i in range(L):
        if name == "spd":
            A[i, i] += 2.0 + p[0]
            if i > 0:
                A[i, i - 1] -= 1.0
            if i + 1 < L:
                A[i, i + 1] -= 1.0
        else:
            A[i, i] += 2.0 + p[0] + 0.01 * i
            if i > 0:
                A[i, i - 1] -= 1.0 + p[1]
            if i + 1 < L:
                A[i, i + 1] -= 1.0 - p[1]
            A[i, (i * 7 + 3) % L] += 0.05
    return A  # (name: "spd" or any "convdiff*")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="/root/reference not mounted (GPU box)")
@pytest.mark.parametrize("name", list(CASES))
def test_fixture_is_what_the_reference_solver_prints(name):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "gmres_ref"], check=True, capture_output=True)
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmres_ref"), name], check=True, capture_output=True,
                         text=True).stdout
    assert out == open(os.path.join(ROOT, "tests", "golden", f"user_gmres_{name}.txt")).read()


@pytest.mark.parametrize("name", list(CASES))
def test_fixture_solves_its_systems(name):
    """Sanity of the vectors themselves (numpy, no GPU): with k_max >= len the reference's GMRES has solved A x = b."""
    rows = fixture(name)
    L = CASES[name][1]
    for r in rows:
        i, kmax = int(r[0]), int(r[1])
        p, b, _ = scenario(name, i)
        res = np.linalg.norm(dense(name, p) @ r[3:] - b) / np.linalg.norm(b)
        if kmax >= 30 and L <= 40:
            assert res < 1e-5, (i, kmax, res)   # (this case runs with tol = 1e-6: the loop ends on |rho_e| < tol)
        assert res < 1.0


def test_operator_plugin_builds_and_registers():
    if not os.path.exists(plugin._build.HIPCC):
        pytest.skip("hipcc not available")
    so = plugin.build_operator(OPS, "ConvDiffOp", name="convdiff")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    for s in ("cgmres_hip_opplugin_abi", "cgmres_hip_opplugin_info", "cgmres_hip_opplugin_solve", "cgmres_hip_opplugin_last_error"):
        assert s in syms
    oid = plugin.register_operator(so)
    assert oid >= 0 and plugin.register_operator(so) == oid
    with pytest.raises(cg.CgmresHipError):
        plugin.register_operator(os.path.join(ROOT, "cgmres_cpp_amd", "lib", "libcgmres_hip.so"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_gmres_with_a_user_operator_vs_the_reference(name):
    """Every (instance, k_max, tol) record of the reference in ONE batched call per (k_max, tol): x within 1e-9 relative
    (observed ~1e-14), plus the exit bookkeeping.  All four operators fit the one-wave-per-system solver
    (gmres_wave_kernel: vectors over the lanes, wave-wide sums; len = 150 / 300 = 3 / 5 elements per lane)."""
    cls, L, npar, kmaxs = CASES[name]
    oid = plugin.register_operator(plugin.build_operator(OPS, cls, name=name))
    rows = fixture(name)
    for c, (kmax, tol) in enumerate(zip(kmaxs, TOLS)):
        n = N_INST[name]
        P, Bv, X0 = zip(*[scenario(name, i) for i in range(n)])
        x, n_ax, why = cg.gmres_user(oid, np.array(X0), np.array(Bv), kmax, tol, np.array(P))
        ref = rows[n * c:n * c + n]
        assert np.array_equal(ref[:, 0], np.arange(n)) and np.all(ref[:, 1] == kmax)
        scale = np.max(np.abs(ref[:, 3:]))
        assert np.max(np.abs(x - ref[:, 3:])) <= 1e-9 * scale, (name, kmax, tol, np.max(np.abs(x - ref[:, 3:])))
        assert np.all(n_ax <= kmax) and np.all(n_ax >= 1)
        if tol == 0.0 and kmax < L:
            assert np.all(n_ax == kmax) and np.all(why == cg.EXIT_NATURAL)
        assert np.all((why == cg.EXIT_NATURAL) | (why == cg.EXIT_CONVERGED) | (why == cg.EXIT_BREAKDOWN))
        if tol > 0 and kmax >= L:
            assert np.all(why != cg.EXIT_NATURAL)   # a full Krylov space: converged (or broke down on the exact solve) before k_max
    # a second call with a different batch size reuses nothing (stateless) and a zero right-hand side leaves at the
    # residual test with x untouched (gmres.hpp:39-41)
    x0 = np.zeros((3, L))
    x, n_ax, why = cg.gmres_user(oid, x0, np.zeros((3, L)), 5, 1e-6, np.array([scenario(name, 0)[0]] * 3))
    assert np.array_equal(x, x0) and np.all(n_ax == 0) and np.all(why == cg.EXIT_SMALL_RESIDUAL)


@pytest.mark.gpu
def test_facade_gmres_subclass_runs_on_the_device(tmp_path):
    """tests/user_models/gmres_main.cpp — a class derived from Gmres with its own Ax_func — compiled against THIS
    repository's include/gmres.hpp: gmres() forwards to the device solver once the operator plugin is named, and ends the
    program (no CPU fallback) when it is not."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    so = plugin.build_operator(OPS, "ConvDiffOp", name="convdiff")
    lib_dir = os.path.join(ROOT, "cgmres_cpp_amd", "lib")
    exe = tmp_path / "gmres_main"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/tests/user_models",
                    os.path.join(ROOT, "tests", "user_models", "gmres_main.cpp"), f"-L{lib_dir}", f"-Wl,-rpath,{lib_dir}",
                    "-lcgmres_hip", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), so], check=True, capture_output=True, text=True).stdout
    got = np.array([[float(v) for v in l.split()] for l in out.strip().split("\n") if l[0].isdigit()])
    ref = fixture("convdiff")
    assert got.shape == ref.shape and np.max(np.abs(got[:, 3:] - ref[:, 3:])) <= 1e-9 * np.max(np.abs(ref[:, 3:]))
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode != 0 and "no device operator registered" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="/root/reference not mounted (GPU box)")
def test_the_same_main_links_against_the_reference_headers(tmp_path):
    """gmres_main.cpp is source-compatible with the reference: built against /root/reference/include it runs the host
    solver and prints the fixture."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "gmres_main_ref"
    subprocess.run(["g++", "-O3", "-std=c++17", "-ffp-contract=off", "-I/root/reference/include",
                    f"-I{ROOT}/tests/user_models", os.path.join(ROOT, "tests", "user_models", "gmres_main.cpp"),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = np.array([[float(v) for v in l.split()] for l in out.strip().split("\n")])
    ref = fixture("convdiff")
    # (this main has no zero-filling operator new[]: the reference reads uninitialised h_mat entries it never uses)
    assert np.max(np.abs(got[:, 3:] - ref[:, 3:])) <= 1e-12 * np.max(np.abs(ref[:, 3:]))
