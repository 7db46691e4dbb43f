"""CPU: libcgmres_hip.so loads without a GPU, exports every function include/cgmres_hip.h declares,
and fails loudly (no CPU fallback) when asked to create a controller without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

import cgmres_cpp_amd as cg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(cg.lib_path()):
        from cgmres_cpp_amd import build
        build.build()
    return cg.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cgmres_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgmres_hip_[a-z0-9_A-Z]+)\s*\(", src)))


def test_header_library_and_binding_agree(lib):
    decl = declared_symbols()
    assert decl == sorted(cg.SYMBOLS), set(decl) ^ set(cg.SYMBOLS)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/cgmres_hip.h but not exported"


def test_registry_without_gpu(lib):
    mi = cg.model_info("pendulum")
    assert (mi["dim_x"], mi["dim_u"], mi["dim_p"], mi["dv"], mi["k_max"]) == (4, 3, 2, 25, 5)
    assert (mi["dt"], mi["h"], mi["zeta"], mi["Tf"], mi["alpha"], mi["tol"]) == (0.001, 0.002, 1000.0, 0.5, 0.5, 1e-6)
    mi = cg.model_info("msd")
    assert (mi["dim_x"], mi["dim_u"], mi["dim_p"], mi["dv"], mi["k_max"], mi["Tf"]) == (4, 6, 2, 50, 5, 1.0)
    mi = cg.model_info("semiactive")
    assert (mi["dim_x"], mi["dim_u"], mi["dim_p"], mi["dv"], mi["k_max"]) == (2, 3, 0, 50, 5)
    with pytest.raises(cg.CgmresHipError):
        cg.model_info(17)


def test_config_struct_layout():
    # int32 x10 (ABI 2: + flags, reserved), double x6, pointer: the C struct of include/cgmres_hip.h
    assert ctypes.sizeof(cg.Config) == 10 * 4 + 6 * 8 + 8
    assert cg.Config.flags.offset == 32 and cg.Config.tol.offset == 40 and cg.Config.stream.offset == 88
    hdr = open(os.path.join(ROOT, "include", "cgmres_hip.h")).read()
    assert f"#define CGMRES_HIP_ABI_VERSION {cg.ABI_VERSION}" in hdr


def test_argument_validation_and_no_cpu_fallback(lib):
    cfg = cg.Config()
    assert lib.cgmres_hip_default_config(cg.PENDULUM, ctypes.byref(cfg)) == 0
    h = ctypes.c_void_p()
    cfg.abi_version = 99
    assert lib.cgmres_hip_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"ABI" in lib.cgmres_hip_last_error()
    cfg.abi_version = cg.ABI_VERSION
    cfg.dv = 0
    assert lib.cgmres_hip_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.dv = 20000  # dim_u*dv beyond the reference's int16 index range
    assert lib.cgmres_hip_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.dv = 25
    cfg.reserved = 7  # documented as 0: refused, so the field can be given a meaning later
    assert lib.cgmres_hip_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"reserved" in lib.cgmres_hip_last_error()
    cfg.reserved = 0
    cfg.flags = 1 << 20
    assert lib.cgmres_hip_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"unknown flags" in lib.cgmres_hip_last_error()
    cfg.flags = 0
    cfg.variant = 5
    assert lib.cgmres_hip_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"unknown variant" in lib.cgmres_hip_last_error()
    cfg.variant = 0
    if cg.device_count() == 0:
        with pytest.raises(cg.CgmresHipError, match="no HIP device|no CPU path"):
            cg.CgmresBatch("pendulum", batch=4)
    assert lib.cgmres_hip_control(None, None, None) == -1


def test_scenarios_match_checker_recipe(orc):
    """The product-side input generator (cgmres_cpp_amd/scenarios.py) and the checker's draw identical batches."""
    import numpy as np
    from cgmres_cpp_amd import scenarios
    for m in (0, 1, 2):
        for a, b in zip(orc.batch_scenario(m, 33), scenarios.batch(m, 33)):
            assert np.array_equal(a, b)


def test_dtype_spellings_and_buffer_checks():
    """Every fp32/fp64 spelling resolves; anything else raises (no silent fp64 default); typed buffers handed to the
    device entry points are checked for element type and count before their address crosses the ABI."""
    import torch
    for d in ("f64", "float64", np.float64, np.dtype("float64"), torch.float64, cg.F64):
        assert cg.parse_dtype(d) == cg.F64
    for d in ("f32", "float32", np.float32, np.dtype("float32"), torch.float32, cg.F32):
        assert cg.parse_dtype(d) == cg.F32
    for d in ("f16", np.int32, torch.bfloat16, 2, None, True):
        with pytest.raises(TypeError):
            cg.parse_dtype(d)
    t = torch.zeros(4, 3, dtype=torch.float32)
    assert cg._ptr(t, np.float32, 12) == t.data_ptr()
    with pytest.raises(TypeError):
        cg._ptr(t, np.float64, 12)
    with pytest.raises(ValueError):
        cg._ptr(t, np.float32, 16)
    with pytest.raises(ValueError):
        cg._ptr(t.t(), np.float32, 12)
    a = np.zeros((4, 3))
    assert cg._ptr(a, np.float64, 12) == a.ctypes.data
    assert cg._ptr(12345) == 12345 and cg._ptr(None) is None
