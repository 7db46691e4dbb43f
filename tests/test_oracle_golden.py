"""CPU: the oracle (oracle/cgmres_oracle.hpp) against the committed fixtures that oracle/gen_golden.py
produced from the unmodified reference.  fp64: bit-exact.  fp32: the restatement rounds every literal to
float while the `#define double float` reference keeps a few double literals, so a small tolerance."""
import numpy as np
import pytest

from conftest import golden_files, golden_ids, load_golden


def _tol(case):
    return dict(rtol=0, atol=0) if case["dtype"] == "f64" else dict(rtol=2e-4, atol=2e-4)


def _mk(orc, g):
    c = g["_case"]
    ctrl = orc.Controller(c["model"], c["dv"], c["kmax"], c["tol"], c["dtype"], which="oracle")
    if c["dim_p"]:
        ctrl.set_ptau(g["ptau"])
    return ctrl


@pytest.mark.parametrize("path", golden_files(), ids=golden_ids())
def test_teacher_forced_control(orc, path):
    g = load_golden(path)
    case = g["_case"]
    for tick in g["_ticks"]:
        p = f"tick{tick}_"
        ctrl = _mk(orc, g)
        ctrl.set_state(g[p + "t"][0], g[p + "U"], g[p + "dUdt"])
        u = ctrl.control(g[p + "x"])
        t1, U1, d1 = ctrl.get_state()
        n_ax, k_used, why = ctrl.last_solve()
        if case["dtype"] == "f64":
            assert np.array_equal(u, g[p + "u"]), (tick, u, g[p + "u"])
            assert np.array_equal(U1, g[p + "U1"])
            assert np.array_equal(d1, g[p + "dUdt1"])
            assert n_ax == int(g[p + "n_ax"][0])
            V, H, rho, gv = ctrl.krylov()
            k = n_ax
            assert np.array_equal(H[:k, :k + 1], g[p + "H"][:k, :k + 1])
            assert np.array_equal(gv[:k], g[p + "g"][:k])
        else:
            np.testing.assert_allclose(u, g[p + "u"], **_tol(case))
            np.testing.assert_allclose(U1, g[p + "U1"], **_tol(case))


@pytest.mark.parametrize("path", golden_files(), ids=golden_ids())
def test_F_Ax_gmres_records(orc, path):
    g = load_golden(path)
    case = g["_case"]
    for tick in g["_ticks"]:
        p = f"tick{tick}_"
        ctrl = _mk(orc, g)
        ctrl.set_state(g[p + "t"][0], g[p + "U"], g[p + "dUdt"])
        F0 = ctrl.F(g[p + "U"], g[p + "x"], g[p + "t"][0])
        b = ctrl.prepare(g[p + "x"])
        ax = ctrl.Ax(g[p + "Ax_v"])
        sol = ctrl.gmres(g[p + "dUdt"], g[p + "b"])
        if case["dtype"] == "f64":
            assert np.array_equal(F0, g[p + "F0"])
            assert np.array_equal(b, g[p + "b"])
            assert np.array_equal(ax, g[p + "Ax_out"])
            assert np.array_equal(sol, g[p + "gmres_x"])
            assert ctrl.last_solve()[0] == int(g[p + "gmres_nax"][0])
        else:
            np.testing.assert_allclose(F0, g[p + "F0"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("path", golden_files(), ids=golden_ids())
def test_closed_loop_and_batch(orc, path):
    g = load_golden(path)
    case = g["_case"]
    ctrl = orc.Controller(case["model"], case["dv"], case["kmax"], case["tol"], case["dtype"], which="oracle")
    un = orc.start_controller(ctrl, g["x0"], g["u0_guess"], g["p"])
    us, xs, ks, _ = orc.closed_loop(ctrl, g["x0"], len(g["loop_u"]))
    if case["dtype"] == "f64":
        assert np.array_equal(un, g["u0_newton"])
        assert np.array_equal(us, g["loop_u"]) and np.array_equal(xs, g["loop_x"])
        assert np.array_equal(ks, g["loop_k"])
    else:
        np.testing.assert_allclose(us[:20], g["loop_u"][:20], rtol=5e-3, atol=5e-3)
    # the seeded batch recipe itself (splitmix64) must reproduce the stored inputs exactly
    bx0, bu0, bp = orc.batch_scenario(case["model"], len(g["batch_x0"]))
    assert np.array_equal(bx0, g["batch_x0"]) and np.array_equal(bp, g["batch_p"])
    if case["dtype"] == "f64":
        for i in range(len(bx0)):
            ci = orc.Controller(case["model"], case["dv"], case["kmax"], case["tol"], "f64", which="oracle")
            assert np.array_equal(orc.start_controller(ci, bx0[i], bu0[i], bp[i]), g["batch_u0_newton"][i])
            u_i, x_i, k_i, _ = orc.closed_loop(ci, bx0[i], g["batch_u"].shape[1])
            assert np.array_equal(u_i, g["batch_u"][i]) and np.array_equal(k_i, g["batch_k"][i])
