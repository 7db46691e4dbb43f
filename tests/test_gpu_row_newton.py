"""GPU parity of the wg kernel with ROW-PARALLEL sweeps (csrc/tick_wg.hip.h: NWT — inside the Arnoldi loop every row of
16 lanes runs its instance's state recurrence of cgmres.hpp:132-140 as Newton's method on the whole trajectory, four
stages per lane, and the costate recurrence :145-153 as three in-row scans; the preamble's costate sweeps likewise; the
Hessenberg QR of a fixed-k solve after the loop, one column per lane) beyond what the variant-parametrised tests of
test_gpu_parity.py / test_gpu_closed_loop.py cover: which handles get it, horizon lengths at the edges of the
four-stages-per-lane ownership, the fresh-sin/cos path of the Newton iteration, the exits of gmres.hpp, the exported
Krylov arrays of both column schedules, and agreement with the serial-sweep kernel of the same library.
Checker: the oracle (oracle/liboracle.so), same seeded inputs.  Tolerances are SURVEY.md §8(c)'s."""
import numpy as np
import pytest

import cgmres_cpp_amd as cg
from test_gpu_wave import _refs, _teacher_forced, U_TOL

pytestmark = pytest.mark.gpu

NAME = "wg+row-newton"


def _batch(B, dv, km, tol, flags=0):
    c = cg.CgmresBatch("pendulum", batch=B, dv=dv, k_max=km, tol=tol, variant=2, flags=flags)
    return c


def test_which_handles_get_the_row_parallel_kernel():
    for kw, want in ((dict(dv=50, k_max=10), True), (dict(dv=43, k_max=5), True), (dict(dv=53, k_max=8), True),
                     (dict(dv=44, k_max=12), True), (dict(dv=42, k_max=10), True), (dict(dv=33, k_max=10), True),
                     (dict(dv=32, k_max=10), False), (dict(dv=25, k_max=5), False),  # short horizons: the serial sweep is cheaper
                     (dict(dv=54, k_max=10), False),                       # dim_u*dv > 160: the long-vector kernels
                     (dict(dv=50, k_max=10, dtype="f32"), False),
                     (dict(dv=50, k_max=10, flags=cg.FLAG_SERIAL_STATE_SWEEP), False),
                     (dict(dv=50, k_max=10, flags=cg.FLAG_SERIAL_COSTATE), False),
                     (dict(dv=50, k_max=10, variant=3), False)):
        args = dict(model="pendulum", batch=4096, variant=2)
        args.update(kw)
        c = cg.CgmresBatch(**args)
        assert (c.variant_name == NAME) == want, (kw, c.variant_name)
        c.close()
    c = cg.CgmresBatch("pendulum", batch=4096, dv=50, k_max=10)  # the headline handle, library's choice
    assert c.variant_name == NAME
    c.close()
    # the semi-active damper (state equation affine in x): scans only, "wg+row-scan"; the two-mass system: neither
    for kw, want in ((dict(model="semiactive", dv=50, k_max=10), "wg+row-scan"), (dict(model="semiactive", dv=7, k_max=3), "wg+row-scan"),
                     (dict(model="semiactive", dv=53, k_max=12), "wg+row-scan"),
                     (dict(model="semiactive", dv=50, k_max=10, dtype="f32"), "wg+parallel-costate"),
                     (dict(model="semiactive", dv=50, k_max=10, flags=cg.FLAG_SERIAL_STATE_SWEEP), "wg+parallel-costate"),
                     (dict(model="semiactive", dv=50, k_max=10, flags=cg.FLAG_SERIAL_COSTATE), "wg"),
                     (dict(model="msd", dv=26, k_max=10), None)):
        args = dict(batch=4096, variant=2)
        args.update(kw)
        c = cg.CgmresBatch(**args)
        assert c.variant_name == want if want else c.variant_name not in (NAME, "wg+row-scan"), (kw, c.variant_name)
        c.close()


@pytest.mark.parametrize("dv,km", [(33, 10), (34, 4), (35, 6), (36, 12), (39, 8), (42, 10),
                                    (43, 3), (44, 12), (46, 10), (45, 6), (47, 7), (48, 9), (49, 10), (51, 4), (52, 10), (53, 8)])
def test_horizon_lengths_at_the_edges_of_the_stage_ownership(orc, dv, km):
    """Lane r of a row owns stages 4r..4r+3 and the terminal state sits at "stage dv": horizons whose last stage is the
    first, second, third and fourth stage of its lane, from the shortest horizon the library gives this kernel (33: part
    of the base trajectory in an LDS region of its own) to the longest (53)."""
    B = 21
    x0, u0, p = orc.batch_scenario(0, B)
    c = _batch(B, dv, km, 1e-6)
    if c.variant_name != NAME:
        c.close()
        pytest.skip(f"LDS plan of dv = {dv}, k_max = {km}: {c.variant_name}")
    c.set_ptau_repeat(p), c.init_u0(u0), c.init_u0_newton(u0, x0, p, 10)
    refs = _refs(orc, 0, dv, km, 1e-6, x0, u0, p)
    for r in refs:
        _, U_o, d_o = r.get_state()
        r.set_state(0.7, U_o, d_o)
    _teacher_forced(orc, c, refs, x0, 3)
    c.close()


@pytest.mark.parametrize("dv,km", [(2, 2), (3, 3), (4, 4), (5, 5), (8, 4), (17, 6), (31, 10), (32, 12), (47, 7), (50, 10), (53, 12)])
def test_row_scan_kernel_of_the_semiactive_damper_at_every_ownership_edge(orc, dv, km):
    """"wg+row-scan" (tick_wg.hip.h: NWT = 2): the whole evaluation of F as two scans over the row — horizons from two
    stages (one lane) up, the terminal stage in every position of its lane, ragged batch, early exits."""
    B = 21
    x0, u0, p = orc.batch_scenario(2, B)
    c = cg.CgmresBatch("semiactive", batch=B, dv=dv, k_max=km, tol=1e-6, variant=2)
    if c.variant_name != "wg+row-scan":
        c.close()
        pytest.skip(f"LDS plan of dv = {dv}, k_max = {km}: {c.variant_name}")
    c.init_u0(u0), c.init_u0_newton(u0, x0, None, 10)
    refs = _refs(orc, 2, dv, km, 1e-6, x0, u0, p)
    for r in refs:
        _, U_o, d_o = r.get_state()
        r.set_state(0.7, U_o, d_o)
    _teacher_forced(orc, c, refs, x0, 3)
    c.close()


@pytest.mark.parametrize("tol", [1e-6, 0.0])
@pytest.mark.parametrize("flags,name", [(0, "rotation"), (cg.FLAG_WAVE_FRESH_TRIG, "fresh-trig")])
def test_both_trig_forms_of_the_newton_sweep_vs_oracle(orc, flags, name, tol):
    """The Newton iteration carries its sin/cos values by rotation; an angle increment beyond the rotation's range makes
    the wave evaluate them afresh (forced here by the flag).  Ragged batch (67 = 4 workgroups + 3 rows), early horizon
    and the horizon fully open, both column schedules (tol > 0: in place; tol = 0: QR after the loop)."""
    B, dv, km = 67, 50, 10
    x0, u0, p = orc.batch_scenario(0, B)
    c = _batch(B, dv, km, tol, flags)
    assert c.variant_name == NAME
    c.set_ptau_repeat(p), c.init_u0(u0), c.init_u0_newton(u0, x0, p, 10)
    refs = _refs(orc, 0, dv, km, tol, x0, u0, p)
    _teacher_forced(orc, c, refs, x0, 4)
    for r in refs:
        _, U_o, d_o = r.get_state()
        r.set_state(2.0, U_o, d_o)
    _teacher_forced(orc, c, refs, x0, 4)
    c.close()


def test_free_running_closed_loop_agrees_with_the_serial_sweep_kernel_and_the_oracle(orc):
    """60 device-resident ticks from the same start on the row-parallel kernel and on the serial-sweep kernel of the same
    library (FLAG_SERIAL_STATE_SWEEP): same Arnoldi counts, u within the parity bound of each other, and a sample of
    instances against the free-running oracle."""
    B, dv, km, n = 300, 50, 10, 60
    x0, u0, p = orc.batch_scenario(0, B)
    out = []
    for flags in (0, cg.FLAG_SERIAL_STATE_SWEEP):
        c = _batch(B, dv, km, 1e-6, flags)
        assert (c.variant_name == NAME) == (flags == 0)
        c.set_ptau_repeat(p), c.init_u0(u0), c.init_u0_newton(u0, x0, p, 10)
        xd, ud = c.device_buffer((B, 4)).upload(x0), c.device_buffer((B, 3))
        c.closed_loop_device(xd, ud, n)
        c.synchronize()
        out.append((xd.download(), ud.download(), c.get_status()))
        c.close()
    (xa, ua, (ka, ra)), (xb, ub, (kb, rb)) = out
    assert np.array_equal(ka, kb) and np.array_equal(ra, rb)
    assert np.max(np.abs(ua - ub)) <= 1e-8 and np.max(np.abs(xa - xb)) <= 1e-8
    for i in (0, 17, 150, 299):
        r = orc.Controller(0, dv, km, 1e-6)
        orc.start_controller(r, x0[i], u0[i], p[i])
        x = x0[i].copy()
        for _ in range(n):
            u = r.control(x)
            x = x + r.plant(x, u) * r.dt
        assert np.max(np.abs(u - ua[i])) <= 1e-7 and np.max(np.abs(x - xa[i])) <= 1e-7, i


def test_exit_paths_of_gmres(orc):
    """gmres.hpp:39-41 (||r0|| < tol) on the row-parallel kernel; :93-95 (convergence) is what every tol > 0 test of this
    file and of the variant-parametrised files exercises."""
    B, dv, km = 37, 44, 4
    x0, u0, p = orc.batch_scenario(0, B)

    def one(tol, u_init, xx, pp):
        c = _batch(B, dv, km, tol)
        assert c.variant_name == NAME
        c.set_ptau_repeat(pp)
        c.init_u0(u_init)
        _, U0, d0 = c.get_state()
        u = c.control(xx)
        n_ax, reason = c.get_status()
        _, U1, d1 = c.get_state()
        c.close()
        for i in range(B):
            r = orc.Controller(0, dv, km, tol)
            r.set_ptau_repeat(pp[i])
            r.init_u0(u_init[i])
            ur = r.control(xx[i])
            k_o, _, reason_o = r.last_solve()
            assert n_ax[i] == k_o and reason[i] == reason_o, (tol, i, n_ax[i], k_o, reason[i], reason_o)
            if np.all(np.isfinite(ur)):
                assert np.max(np.abs(u[i] - ur)) <= U_TOL * max(1.0, float(np.max(np.abs(ur)))), (tol, i)
        return U0, d0, U1, d1, n_ax, reason

    U0, d0, U1, d1, n_ax, reason = one(1e30, u0, x0, p)
    assert np.all(reason == cg.EXIT_SMALL_RESIDUAL) and np.all(n_ax == 0)
    assert np.array_equal(d0, d1) and np.allclose(U1, U0 + d0 * 1e-3)
    # (gmres.hpp:63-65, the breakdown exit, cannot be provoked on this kernel with finite numbers: a direction is absorbed
    # by the rounding of U + h*v only for |U| > 4e12, and controls of that size overflow the costate recurrence of the
    # pendulum within 20 of the >= 42 stages this kernel takes — in the reference as well.  The wave mapping, which
    # shares the rule, is tested for it at dv = 12: test_gpu_wave.py.)


@pytest.mark.parametrize("tol", [1e-6, 0.0])
def test_exported_krylov_arrays_vs_oracle(orc, tol):
    """get_krylov after one tick — basis, rotated Hessenberg, reflectors, residual vector — against the oracle's private
    members: tol > 0 rotates every column in its iteration (hess_column), tol = 0 factorises after the loop with one
    column per lane."""
    B, dv, km = 19, 50, 10
    x0, u0, p = orc.batch_scenario(0, B)
    c = _batch(B, dv, km, tol)
    assert c.variant_name == NAME
    c.set_ptau_repeat(p), c.init_u0(u0), c.init_u0_newton(u0, x0, p, 10)
    refs = _refs(orc, 0, dv, km, tol, x0, u0, p)
    for r in refs:
        _, U_o, d_o = r.get_state()
        r.set_state(0.4, U_o, d_o)
    t_o, U_o, d_o = zip(*[r.get_state() for r in refs])
    c.set_state(t_o[0], np.array(U_o), np.array(d_o))
    c.control(x0)
    n_ax, reason = c.get_status()
    V, H, rho, g = c.get_krylov(with_V=True)
    k1 = km + 1
    for i, r in enumerate(refs):
        r.control(x0[i])
        k_o, ks_o, _ = r.last_solve()
        assert n_ax[i] == k_o
        Vo, Ho, rhoo, go = r.krylov()
        Hd = np.asarray(H[i]).reshape(k1, k1)
        cols = min(k_o, 4)  # leading columns: later ones are built on a converged residual
        for col in range(cols):
            ref_col = Ho[col][: col + 1]
            scale = max(1.0, float(np.max(np.abs(ref_col))))
            assert np.max(np.abs(np.abs(Hd[col][: col + 1]) - np.abs(ref_col))) <= 1e-6 * scale, (i, col)
            assert np.max(np.abs(np.abs(np.asarray(g[i]).reshape(km, 3)[col]) - np.abs(go[col]))) <= 1e-6 * max(
                1.0, float(np.max(np.abs(go[col])))), (i, col)
        nv = min(cols + 1, k1)
        Vd = np.asarray(V[i]).reshape(k1, -1)
        assert np.max(np.abs(Vd[:nv] @ Vd[:nv].T - np.eye(nv))) < 1e-8
        assert np.max(np.abs(np.abs(Vd[:nv]) - np.abs(Vo[:nv]))) <= 1e-6
    c.close()
