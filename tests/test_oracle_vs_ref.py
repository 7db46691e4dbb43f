"""CPU, only where /root/reference is mounted (skipped on the GPU box): the oracle against the live
compiled reference (oracle/_ref/libref.so) on long closed loops and seeded random states — bit-exact."""
import os

import numpy as np
import pytest

REF_PRESENT = os.path.isdir("/root/reference")
pytestmark = pytest.mark.skipif(not REF_PRESENT, reason="/root/reference not mounted (GPU box)")

CASES = [(0, 25, 5, -1.0), (0, 50, 10, -1.0), (0, 50, 10, 0.0), (0, 100, 20, -1.0),
         (1, 50, 5, -1.0), (1, 20, 5, -1.0), (1, 50, 10, -1.0),
         (2, 50, 5, -1.0), (2, 50, 10, -1.0), (2, 8, 3, -1.0)]


@pytest.fixture(scope="module")
def ref(orc):
    if not orc.have_ref():
        orc.build(ref=True)
    return orc


@pytest.mark.parametrize("model,dv,kmax,tol", CASES)
def test_closed_loop_bit_exact(ref, model, dv, kmax, tol):
    a = ref.Controller(model, dv, kmax, tol, which="ref")
    b = ref.Controller(model, dv, kmax, tol, which="oracle")
    x0, u0, p = ref.shipped_scenario(model)
    assert np.array_equal(ref.start_controller(a, x0, u0, p), ref.start_controller(b, x0, u0, p))
    n = 1500 if dv < 100 else 400
    ua, xa, ka, _ = ref.closed_loop(a, x0, n)
    ub, xb, kb, _ = ref.closed_loop(b, x0, n)
    assert np.array_equal(ua, ub) and np.array_equal(xa, xb) and np.array_equal(ka, kb)


@pytest.mark.parametrize("model,dv,kmax,tol", CASES[:3] + CASES[4:5] + CASES[7:8])
def test_random_state_records(ref, model, dv, kmax, tol):
    rng = np.random.default_rng(99 + model)
    a = ref.Controller(model, dv, kmax, tol, which="ref")
    b = ref.Controller(model, dv, kmax, tol, which="oracle")
    x0, u0, p = ref.shipped_scenario(model)
    for trial in range(10):
        U = np.tile(u0, dv) * (1 + 0.05 * rng.standard_normal(a.len))
        d = 0.1 * rng.standard_normal(a.len) if trial else np.zeros(a.len)
        t = float(rng.uniform(0, 3)) if trial else 0.0
        x = x0 + 0.1 * rng.standard_normal(a.dim_x)
        pt = np.tile(p, dv + 1) + (0.05 * rng.standard_normal(a.dim_p * (dv + 1)) if a.dim_p else 0)
        for c in (a, b):
            c.set_ptau(pt)
            c.set_state(t, U, d)
        assert np.array_equal(a.F(U, x, t), b.F(U, x, t))
        assert np.array_equal(a.prepare(x), b.prepare(x))
        v = rng.standard_normal(a.len)
        assert np.array_equal(a.Ax(v), b.Ax(v))
        assert np.array_equal(a.control(x), b.control(x))
        assert a.last_solve()[0] == b.last_solve()[0]
        for qa, qb in zip(a.get_state(), b.get_state()):
            assert np.array_equal(qa, qb)


def test_fp32_restatement_close_to_fp32_reference(ref):
    """fp32: not bit-exact by construction (double literals survive in the macro-converted reference)."""
    for model in (0, 1, 2):
        a = ref.Controller(model, 50, 10, -1.0, "f32", which="ref")
        b = ref.Controller(model, 50, 10, -1.0, "f32", which="oracle")
        x0, u0, p = ref.shipped_scenario(model)
        ref.start_controller(a, x0, u0, p)
        ref.start_controller(b, x0, u0, p)
        for tick in range(5):
            t, U, d = a.get_state()
            b.set_state(t, U, d)
            ua, ub = a.control(x0), b.control(x0)
            np.testing.assert_allclose(ua, ub, rtol=0, atol=1e-4)
