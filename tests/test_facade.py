"""Drop-in boundary: the C++ façade headers (include/cgmres.hpp, gmres.hpp, matrix.hpp) over the C ABI.
CPU part: the repo's example program and — where /root/reference is mounted — the reference's four UNMODIFIED
main.cpp files compile warning-free and link against libcgmres_hip.so.  GPU part: they run and write the
reference's trajectory text format with the reference's numbers."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "examples", "_build")
REF_MAINS = os.path.join(ROOT, "oracle", "_ref", "mains")
REF_PRESENT = os.path.isdir("/root/reference")


def _make(*targets):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples")] + list(targets), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "warning" not in (r.stdout + r.stderr).lower(), r.stdout + r.stderr
    return r


def test_example_program_builds():
    import cgmres_cpp_amd as cg
    if not os.path.exists(cg.lib_path()):
        from cgmres_cpp_amd import build
        build.build()
    _make("all")
    assert os.path.exists(os.path.join(BUILD, "closed_loop"))


@pytest.mark.skipif(not REF_PRESENT, reason="/root/reference not mounted (GPU box)")
def test_unmodified_reference_mains_link_against_facade():
    _make("reference")
    for name in ("semiactive_damper", "mass_spring_damper", "arm_type_inverted_pendulum", "multiple_controller"):
        exe = os.path.join(REF_MAINS, name)
        assert os.path.exists(exe)
        needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
        assert "libcgmres_hip.so" in needed  # the tick goes through the C ABI, not through inlined CPU code
        syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
        for fn in ("cgmres_hip_create", "cgmres_hip_control", "cgmres_hip_init_u0_newton", "cgmres_hip_model_probe"):
            assert fn in syms, (name, fn)


def test_matrix_helpers_semantics(tmp_path):
    """matrix.hpp keeps the reference's rounding conventions (reciprocal div, column-major linsolve, sign(0)=+1)."""
    src = tmp_path / "m.cpp"
    src.write_text(r'''
#include "matrix.hpp"
#include <stdio.h>
int main() {
  double a[3] = {1.0, 2.0, 3.0}, b[3] = {0.5, -1.0, 4.0}, r[3];
  add(r, a, b, 3); printf("%.17g %.17g %.17g\n", r[0], r[1], r[2]);
  sub(r, a, b, 3); printf("%.17g %.17g %.17g\n", r[0], r[1], r[2]);
  mul(r, a, 0.1, 3); printf("%.17g %.17g %.17g\n", r[0], r[1], r[2]);
  div(r, a, 3.0, 3); printf("%.17g %.17g %.17g\n", r[0], r[1], r[2]);
  printf("%.17g %.17g %.17g %.17g\n", norm(a, 3), dot(a, b, 3), sign(0.0), sign(-2.0));
  double m[9] = {2, 1, 0, 1, 3, 1, 0, 1, 4}, v[3] = {1, 2, 3};   // column-major, symmetric
  double mv[3]; mul(mv, m, v, 3, 3); printf("%.17g %.17g %.17g\n", mv[0], mv[1], mv[2]);
  linsolve(mv, m, 3); printf("%.15g %.15g %.15g\n", mv[0], mv[1], mv[2]);
  return 0;
}''')
    exe = tmp_path / "m"
    subprocess.run(["g++", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    rows = [[float(t) for t in line.split()] for line in out if line.strip()]
    assert rows[0] == [1.5, 1.0, 7.0] and rows[1] == [0.5, 3.0, -1.0]
    assert rows[2] == [1.0 * 0.1, 2.0 * 0.1, 3.0 * 0.1]
    inv = 1.0 / 3.0
    assert rows[3] == [1.0 * inv, 2.0 * inv, 3.0 * inv]      # multiply by the reciprocal, not a division
    assert rows[4] == [np.sqrt(14.0), 0.5 - 2.0 + 12.0, 1.0, -1.0]
    assert rows[5] == [4.0, 10.0, 14.0]
    np.testing.assert_allclose(rows[6], [1.0, 2.0, 3.0], rtol=1e-13)


def _read_traj(path):
    return np.array([[float(t) for t in line.split("\t")] for line in open(path) if line.strip()])


@pytest.mark.gpu
@pytest.mark.parametrize("which,fixture,ncol", [("pendulum", "pendulum_dv25_k5_tolref_f64", 3),
                                                ("msd", "msd_dv50_k5_tolref_f64", 6),
                                                ("semiactive", "semiactive_dv50_k5_tolref_f64", 3)])
def test_example_program_reproduces_reference_loop(tmp_path, which, fixture, ncol):
    """closed_loop <model> 100 through Cgmres<Model> (façade, batch of one on the GPU) against the reference's own
    closed loop (golden loop_u / loop_x, shipped sizes), in the reference's 6-decimal text format."""
    exe = os.path.join(BUILD, "closed_loop")
    if not os.path.exists(exe):
        _make("all")
    g = load_golden(os.path.join(GOLDEN_DIR, fixture + ".npz"))
    prefix = str(tmp_path / which)
    r = subprocess.run([exe, which, "100", prefix], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Elapsed time = " in r.stdout
    u = _read_traj(prefix + "_u.txt")
    x = _read_traj(prefix + "_x.txt")
    assert u.shape == (100, 1 + ncol)
    np.testing.assert_allclose(u[:, 0], 0.001 * np.arange(100), atol=1e-9)
    assert np.max(np.abs(u[:, 1:] - g["loop_u"][:100])) <= 1.5e-6   # 6 printed decimals
    assert np.max(np.abs(x[:, 1:] - g["loop_x"][:100])) <= 1.5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,devices,ticks", [(300, "0,0", 25), (67, "0,0,0", 12), (4100, "0,0", 11)])
def test_sharded_batch_in_cpp_equals_the_unsharded_batch(B, devices, ticks):
    """include/cgmres_batch.hpp: CgmresBatchSharded<Model> — one handle + stream per listed device, host vectors cut by
    cgmres_hip_shard_bounds, the closed loop resident on every shard, one gather at the end — against unsharded
    CgmresBatch runs of the same instances, bit for bit (controllers are independent).  Both shards sit on the one
    card of the test box; the uneven splits (150 + 150, 23 + 22 + 22, 2050 + 2050: wave and wg mappings) are on purpose."""
    exe = os.path.join(BUILD, "closed_loop")
    _make("all")
    r = subprocess.run([exe, "sharded", str(B), devices, str(ticks)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max |sharded - single| = 0;" in r.stdout, r.stdout
    assert r.stdout.count("shard ") >= len(devices.split(","))


@pytest.mark.gpu
def test_unmodified_reference_main_runs_on_gpu(tmp_path):
    """The reference's own arm_type_inverted_pendulum/main.cpp, compiled unchanged against the façade in the build
    container (oracle/_ref/mains/), run here: its output file must start with the reference's numbers."""
    exe = os.path.join(REF_MAINS, "arm_type_inverted_pendulum")
    if not os.path.exists(exe):
        pytest.skip("reference mains were not built (needs /root/reference at build time)")
    g = load_golden(os.path.join(GOLDEN_DIR, "pendulum_dv25_k5_tolref_f64.npz"))
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    u = _read_traj(tmp_path / "arm_type_inverted_pendulum_u.txt")
    assert u.shape[0] == 10001
    assert np.max(np.abs(u[:101, 1:] - g["loop_u"][:101])) <= 1.5e-6


def test_standalone_gmres_without_a_device_operator_fails_loudly(tmp_path):
    """The facade's `class Gmres` (reference include/gmres.hpp:8-129): a subclass with its own Ax_func compiles and
    links unchanged; calling the protected solver gmres(x, b) (gmres.hpp:28) without having named a device build of
    the operator (use_device_operator) ends the program with a message — there is no host solver behind it, and no
    silent CPU path.  (With an operator plugin the call runs on the GPU: tests/test_user_gmres.py.)"""
    inc = os.path.join(ROOT, "include")
    lib_dir = os.path.join(ROOT, "cgmres_cpp_amd", "lib")
    src = tmp_path / "calls.cpp"
    src.write_text('''
#include "gmres.hpp"
struct Mine : Gmres {
  Mine() : Gmres(4, 2, 1e-6) {}
  void Ax_func(double* Ax, const double* x) override { for (int i = 0; i < 4; ++i) Ax[i] = 2.0 * x[i]; }
  void go(double* x, const double* b) { gmres(x, b); }
};
int main(int argc, char**) { Mine m; double x[4] = {0}, b[4] = {1, 1, 1, 1}; if (argc > 1) m.go(x, b); return 0; }
''')
    exe = tmp_path / "calls"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", inc, str(src), f"-L{lib_dir}", f"-Wl,-rpath,{lib_dir}", "-lcgmres_hip",
                    "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0                       # deriving and constructing is fine
    r = subprocess.run([str(exe), "solve"], capture_output=True, text=True)
    assert r.returncode != 0 and "no device operator registered" in r.stderr and "no CPU fallback" in r.stderr, r.stderr
