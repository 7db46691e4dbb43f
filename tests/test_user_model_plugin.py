"""User-model plugin path: a Model header that is NOT one of the shipped three is compiled
for gfx950 (cgmres_cpp_amd/plugin.py), registered in the library (cgmres_hip_register_model) and driven through the
ordinary C ABI.  Checker: the oracle's generic controller instantiated for the same header on the CPU
(tests/user_models/vdp_oracle.cpp); its output is committed as tests/golden/user_vdp_closed_loop.txt."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import cgmres_cpp_amd as cg
from cgmres_cpp_amd import plugin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "tests", "user_models", "vdp_model.hpp")
GOLDEN = os.path.join(ROOT, "tests", "golden", "user_vdp_closed_loop.txt")
B, TICKS = 6, 25


def scenario():
    b = np.arange(B, dtype=np.float64)
    x0 = np.stack([1.0 + 0.1 * b, -0.5 + 0.05 * b], axis=1)
    p = np.stack([0.2 * b, 0.05 * (np.arange(B) % 3)], axis=1)
    u0 = np.tile(np.array([0.1, 1.9, 0.03]), (B, 1))
    return x0, u0, p


def load_fixture():
    rows = np.loadtxt(GOLDEN)
    assert rows.shape == (B * TICKS, 7)
    u = rows[:, 2:5].reshape(B, TICKS, 3)
    x = rows[:, 5:7].reshape(B, TICKS, 2)
    return u, x


@pytest.fixture(scope="module")
def vdp_plugin():
    if not os.path.exists(plugin._build.HIPCC):
        pytest.skip("hipcc not available")
    return plugin.build(HEADER, cls="VdpModel", name="vdp")


def test_fixture_is_what_the_oracle_prints(tmp_path):
    """The committed vector is the output of the generic oracle controller on this header (rebuilt here with g++)."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "vdp_oracle"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}",
                    os.path.join(ROOT, "tests", "user_models", "vdp_oracle.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), str(B), str(TICKS)], check=True, capture_output=True, text=True).stdout
    assert out == open(GOLDEN).read()


def test_plugin_builds_and_registers(vdp_plugin):
    """No GPU needed: the shared object exports the plugin entry points and the registry answers for its id."""
    syms = subprocess.run(["nm", "-D", "--defined-only", vdp_plugin], check=True, capture_output=True, text=True).stdout
    for s in ("cgmres_hip_plugin_abi", "cgmres_hip_plugin_info", "cgmres_hip_plugin_make", "cgmres_hip_plugin_probe",
              "cgmres_hip_plugin_last_error"):
        assert s in syms
    mid = plugin.register(vdp_plugin)
    assert mid >= 1000 and plugin.register(vdp_plugin) == mid  # idempotent
    mi = cg.model_info(mid)
    assert (mi["dim_x"], mi["dim_u"], mi["dim_p"], mi["dv"], mi["k_max"]) == (2, 3, 2, 30, 6)
    assert (mi["dt"], mi["h"], mi["zeta"], mi["Tf"], mi["alpha"], mi["tol"]) == (0.001, 0.002, 1000.0, 1.0, 0.5, 1e-6)
    with pytest.raises(cg.CgmresHipError):
        plugin.register(os.path.join(ROOT, "cgmres_cpp_amd", "lib", "libcgmres_hip.so"))  # not a plugin


@pytest.mark.gpu
def test_user_model_closed_loop_vs_oracle(vdp_plugin):
    mid = plugin.register(vdp_plugin)
    u_ref, x_ref = load_fixture()
    x0, u0, p = scenario()
    c = cg.CgmresBatch(mid, batch=B)
    assert c.variant == 1  # user models run on the lane mapping
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    x = x0.copy()
    for t in range(TICKS):
        assert np.max(np.abs(x - x_ref[:, t])) <= 1e-9, t
        u = c.control(x)
        assert np.max(np.abs(u - u_ref[:, t])) <= 1e-9, (t, u, u_ref[:, t])
        # plant = the model's own state equation (p is read by dxdt), explicit Euler like <example>/main.cpp:71-73
        f = np.stack([x[:, 1], (1.0 - x[:, 0] ** 2) * x[:, 1] - x[:, 0] + u[:, 0] + p[:, 1]], axis=1)
        x = x + f * 0.001
    c.close()


@pytest.mark.gpu
def test_user_model_probe_and_device_loop(vdp_plugin):
    """model_probe evaluates the DEVICE build of the user functions; closed_loop_device uses Model::dxdt as plant."""
    mid = plugin.register(vdp_plugin)
    x, u, p, l = np.array([0.7, -0.3]), np.array([0.4, 1.5, 0.02]), np.array([0.25, 0.1]), np.array([0.9, -1.1])
    out = np.concatenate(cg.model_probe(mid, x, u, p, l))
    f = [x[1], (1 - x[0] ** 2) * x[1] - x[0] + u[0] + p[1]]
    g = [(x[0] - p[0]) * 2.0, x[1]]
    hx = [(x[0] - p[0]) + l[1] * (-2 * x[0] * x[1] - 1), x[1] + l[0] + l[1] * (1 - x[0] ** 2)]
    hu = [u[0] + l[1] + 2 * u[2] * u[0], -0.1 + 2 * u[2] * u[1], u[0] ** 2 + u[1] ** 2 - 4.0]
    assert np.allclose(out, np.array(f + g + hx + hu), rtol=0, atol=1e-14)
    u_ref, x_ref = load_fixture()
    x0, u0, p0 = scenario()
    c = cg.CgmresBatch(mid, batch=B)
    c.set_ptau_repeat(p0)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p0, 10)
    xd, ud = c.device_buffer((B, 2)).upload(x0), c.device_buffer((B, 3)).upload(np.zeros((B, 3)))
    c.closed_loop_device(xd, ud, TICKS)
    # after TICKS ticks u holds the last tick's control and x the state after it
    assert np.max(np.abs(ud.download() - u_ref[:, TICKS - 1])) <= 1e-9
    c.close()


@pytest.mark.gpu
def test_facade_finds_the_plugin_for_a_user_main(vdp_plugin, tmp_path):
    """A main written against the reference's class surface (`Cgmres<Model>` with the user's own model.hpp) runs
    unchanged: include/cgmres.hpp identifies the Model among the plugins named in CGMRES_HIP_MODEL_PLUGINS."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    lib_dir = os.path.join(ROOT, "cgmres_cpp_amd", "lib")
    exe = tmp_path / "vdp_main"
    subprocess.run(["g++", "-O2", f"-I{ROOT}/include", f"-I{ROOT}/tests/user_models",
                    os.path.join(ROOT, "tests", "user_models", "vdp_main.cpp"), f"-L{lib_dir}",
                    f"-Wl,-rpath,{lib_dir}", "-lcgmres_hip", "-o", str(exe)], check=True)
    env = dict(os.environ, CGMRES_HIP_MODEL_PLUGINS=f"/nonexistent.so:{vdp_plugin}")
    out = subprocess.run([str(exe), "3", "12"], check=True, capture_output=True, text=True, env=env).stdout
    got = np.array([[float(v) for v in line.split()] for line in out.strip().split("\n")])
    u_ref, x_ref = load_fixture()
    assert got.shape == (12, 7)
    assert np.max(np.abs(got[:, 2:5] - u_ref[3, :12])) <= 1e-9
    assert np.max(np.abs(got[:, 5:7] - x_ref[3, :12])) <= 1e-9
    # without the plugin the facade refuses (no CPU fallback)
    r = subprocess.run([str(exe), "3", "2"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k != "CGMRES_HIP_MODEL_PLUGINS"})
    assert r.returncode != 0 and "not in the libcgmres_hip registry" in r.stderr
