"""GPU parity of the WAVE mapping (variant 4, csrc/tick_wave.hip.h: one wavefront per controller, the horizon
recurrences of cgmres.hpp:132-153 as DPP wave scans, Newton's method on the nonlinear part of the trajectory) beyond
what the variant-parametrised tests of test_gpu_parity.py / test_gpu_closed_loop.py cover: its own fall-back paths,
every exit of gmres.hpp on the pendulum, the exported Krylov arrays, horizon lengths at the edges of the lane mapping,
and how the library picks the mapping.  Checker: the oracle (oracle/liboracle.so), same seeded inputs.
Tolerances are SURVEY.md §8(c)'s (fp64 teacher-forced: |du| <= 1e-9, |dUdt| <= 1e-7 rel, Arnoldi counts equal)."""
import numpy as np
import pytest

import cgmres_cpp_amd as cg

pytestmark = pytest.mark.gpu

U_TOL, DUDT_REL = 1e-9, 1e-7
WAVE = 4


def _refs(orc, model, dv, km, tol, x0, u0, p):
    out = []
    for i in range(len(x0)):
        r = orc.Controller(model, dv, km, tol)
        orc.start_controller(r, x0[i], u0[i], p[i])
        out.append(r)
    return out


def _teacher_forced(orc, c, refs, x0, ticks, u_tol=U_TOL):
    """Every tick: controller state and x taken from the oracle, u / dUdt / counts / exit reasons compared."""
    x = x0.copy()
    worst = 0.0
    for tick in range(ticks):
        t_o, U_o, d_o = zip(*[r.get_state() for r in refs])
        c.set_state(t_o[0], np.array(U_o), np.array(d_o))
        u = c.control(x)
        n_ax, reason = c.get_status()
        _, U1, d1 = c.get_state()
        for i, r in enumerate(refs):
            ur = r.control(x[i])
            worst = max(worst, float(np.max(np.abs(u[i] - ur))))
            # u = U + dUdt*dt (cgmres.hpp:102-109): the bound on u that goes with SURVEY 8(c)'s bound on dUdt — relative to
            # |u| and to dt*|dUdt| where a time jump makes them large (|dUdt| ~ 1e5 on the two-mass system)
            d_ref = r.get_state()[2]
            bound = u_tol * max(1.0, float(np.max(np.abs(ur)))) + r.dt * DUDT_REL * max(1.0, float(np.max(np.abs(d_ref))))
            assert np.max(np.abs(u[i] - ur)) <= bound, (tick, i, u[i], ur)
            k_o, _, reason_o = r.last_solve()
            assert n_ax[i] == k_o and reason[i] == reason_o, (tick, i, n_ax[i], k_o, reason[i], reason_o)
            d_ref = r.get_state()[2]
            assert np.max(np.abs(d1[i] - d_ref)) <= DUDT_REL * max(1.0, float(np.max(np.abs(d_ref)))), (tick, i)
            x[i] = x[i] + r.plant(x[i], ur) * r.dt
    return worst


def test_the_library_takes_the_wave_mapping_for_batches_smaller_than_the_gpu():
    small = cg.CgmresBatch("pendulum", batch=300, dv=50, k_max=10)
    assert small.variant == WAVE and small.variant_name == "wave"
    for kw in (dict(flags=cg.FLAG_NO_WAVE), dict(batch=4096), dict(dtype="f32"), dict(dv=64), dict(k_max=12)):
        args = dict(model="pendulum", batch=300, dv=50, k_max=10)
        args.update(kw)
        c = cg.CgmresBatch(**args)
        assert c.variant in (2, 3), (kw, c.variant_name)
        c.close()
    c = cg.CgmresBatch("semiactive", batch=64, dv=50, k_max=10)   # affine in x: one 2 x 2 scan per sweep, no Newton
    assert c.variant == WAVE
    c.close()
    c = cg.CgmresBatch("msd", batch=64, dv=50, k_max=10)          # linear time-invariant: constant-matrix scans
    assert c.variant == WAVE
    c.close()
    with pytest.raises(cg.CgmresHipError, match="wave mapping"):
        cg.CgmresBatch("pendulum", batch=8, dv=64, k_max=10, variant=WAVE)
    with pytest.raises(cg.CgmresHipError, match="wave mapping"):
        cg.CgmresBatch("msd", batch=8, dv=20, k_max=12, variant=WAVE)
    small.close()


@pytest.mark.parametrize("tol", [1e-6, 0.0])
@pytest.mark.parametrize("flags,name", [(0, "newton+rotation"), (cg.FLAG_WAVE_FRESH_TRIG, "newton+fresh-trig"),
                                        (cg.FLAG_WAVE_SERIAL_SWEEPS, "serial-fallback")])
def test_every_form_of_the_mat_vec_sweep_vs_oracle(orc, flags, name, tol):
    """The perturbed state sweep of Ax_func (cgmres.hpp:164-175) has three forms on this mapping: Newton on the
    trajectory with trig values rotated from the base trajectory (the default), the same with fresh sin/cos (taken when
    a trajectory strays from the base by more than the rotation range) and the serial quad sweep (taken when Newton does
    not settle).  Each forced by a flag, 8 teacher-forced ticks of a ragged batch, late horizon included."""
    B, dv, km = 67, 50, 10
    x0, u0, p = orc.batch_scenario(0, B)
    c = cg.CgmresBatch("pendulum", batch=B, dv=dv, k_max=km, tol=tol, variant=WAVE, flags=flags)
    c.set_ptau_repeat(p), c.init_u0(u0), c.init_u0_newton(u0, x0, p, 10)
    refs = _refs(orc, 0, dv, km, tol, x0, u0, p)
    _teacher_forced(orc, c, refs, x0, 4)
    for r in refs:  # ... and with the horizon fully open (t = 2 s: dtau = 0.63 Tf / dv)
        _, U_o, d_o = r.get_state()
        r.set_state(2.0, U_o, d_o)
    _teacher_forced(orc, c, refs, x0, 4)
    c.close()


@pytest.mark.parametrize("model", [0, 2, 1])
@pytest.mark.parametrize("dv,km", [(2, 2), (3, 5), (15, 4), (16, 10), (17, 3), (31, 6), (32, 6), (33, 10),
                                    (47, 7), (48, 9), (49, 10), (62, 5), (63, 10)])
def test_horizon_lengths_at_the_row_boundaries_of_the_scans(orc, dv, km, model):
    """The scans cross DPP rows at lanes 16 / 32 / 48 and the terminal stage sits on lane dv: horizons that end on, just
    before and just after those lanes, the shortest ones, and the longest the mapping takes (dv = 63) — for the pendulum
    (Newton on the trajectory), the semi-active damper (one 2 x 2 affine scan per sweep) and the two-mass system
    (constant-matrix 4 x 4 scans from per-tick power tables)."""
    B = 5
    x0, u0, p = orc.batch_scenario(model, B)
    if model == 1 and 6 * dv > 320:
        pytest.skip("dim_u*dv beyond the wg mapping, which serves the white-box hooks of a wave handle")
    c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=km, tol=1e-6, variant=WAVE)
    if p.shape[1]:
        c.set_ptau_repeat(p)
    c.init_u0(u0), c.init_u0_newton(u0, x0, p if p.shape[1] else None, 10)
    refs = _refs(orc, model, dv, km, 1e-6, x0, u0, p)
    for r in refs:
        _, U_o, d_o = r.get_state()
        r.set_state(0.7, U_o, d_o)
    _teacher_forced(orc, c, refs, x0, 3)
    c.close()


def test_exit_paths_of_gmres_on_the_pendulum(orc):
    """gmres.hpp:39-41 (||r0|| < tol), :93-95 in the first column (k = 0: one mat-vec, 0 x 0 solve) and :63-65 (breakdown)
    on the wave mapping, instance by instance against the oracle."""
    B, dv, km = 37, 12, 4
    x0, u0, p = orc.batch_scenario(0, B)

    def both(tol, u_init=None, xx=x0, p=p):
        c = cg.CgmresBatch("pendulum", batch=B, dv=dv, k_max=km, tol=tol, variant=WAVE)
        c.set_ptau_repeat(p)
        ui = u0 if u_init is None else u_init
        c.init_u0(ui)
        refs = []
        for i in range(B):
            r = orc.Controller(0, dv, km, tol)
            r.set_ptau_repeat(p[i])
            r.init_u0(ui[i])
            refs.append(r)
        if u_init is None:
            c.init_u0_newton(u0, xx, p, 10)
            for i, r in enumerate(refs):
                r.init_u0_newton(u0[i], xx[i], p[i], 10)
        _, U0, d0 = c.get_state()
        u = c.control(xx)
        n_ax, reason = c.get_status()
        _, U1, d1 = c.get_state()
        c.close()
        for i, r in enumerate(refs):
            ur = r.control(xx[i])
            k_o, _, reason_o = r.last_solve()
            assert n_ax[i] == k_o and reason[i] == reason_o, (tol, i, n_ax[i], k_o, reason[i], reason_o)
            if np.all(np.isfinite(ur)):
                assert np.max(np.abs(u[i] - ur)) <= U_TOL * max(1.0, float(np.max(np.abs(ur)))), (tol, i)
        return U0, d0, U1, d1, n_ax, reason, refs

    U0, d0, U1, d1, n_ax, reason, _ = both(1e30)
    assert np.all(reason == cg.EXIT_SMALL_RESIDUAL) and np.all(n_ax == 0)
    assert np.array_equal(d0, d1) and np.allclose(U1, U0 + d0 * 1e-3)

    r0n, e1 = [], []
    for i in range(B):
        r = orc.Controller(0, dv, 1, 0.0)
        orc.start_controller(r, x0[i], u0[i], p[i])
        b = r.prepare(x0[i])
        r0n.append(float(np.linalg.norm(b - r.Ax(r.get_state()[2]))))
        r.control(x0[i])
        e1.append(abs(float(r.krylov()[2][1])))
    r0n, e1 = np.array(r0n), np.array(e1)
    assert np.all(e1 < r0n)
    tol = float(np.sqrt(np.median(r0n) * np.median(e1)))
    U0, d0, U1, d1, n_ax, reason, refs = both(tol)
    hit = (reason == cg.EXIT_CONVERGED) & (n_ax == 1)
    assert hit.sum() >= B // 3, (hit.sum(), tol)
    for i in np.nonzero(hit)[0]:
        assert refs[i].last_solve()[1] == 0
        assert np.array_equal(d0[i], d1[i])

    # breakdown: |U| = 1e17 absorbs h*v (ulp(1e17) = 16 > h), so F(U + h v0) == F(U) bit for bit; the plant rests at the
    # origin with the reference at the origin, so the costate is identically zero and the regrouped sum
    # (phi - Fh)/h + B^T lambda / h is exact as well (see test_status_exit_paths)
    big = np.full((B, 3), 1e17)
    U0, d0, U1, d1, n_ax, reason, _ = both(0.0, u_init=big, xx=np.zeros_like(x0), p=np.zeros_like(p))
    assert np.all(reason == cg.EXIT_BREAKDOWN) and np.all(n_ax == 1), (reason, n_ax)
    assert np.array_equal(d0, d1) and np.array_equal(U1, U0)


@pytest.mark.parametrize("tol", [1e-6, 0.0])
def test_exported_krylov_arrays_vs_oracle(orc, tol):
    """get_krylov after one tick: the basis (registers on this mapping), the rotated Hessenberg (kept by rows over the
    lanes), the reflectors (one per lane) and the residual vector against the oracle's private members."""
    B, dv, km = 9, 50, 10
    x0, u0, p = orc.batch_scenario(0, B)
    c = cg.CgmresBatch("pendulum", batch=B, dv=dv, k_max=km, tol=tol, variant=WAVE)
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
        Hd = np.asarray(H[i]).reshape(k1, k1)  # column-major ld k1 on both sides: [col][row]
        cols = min(k_o, 4)                     # leading columns: later ones are built on a converged residual
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


def test_abi_refuses_aliased_arguments_like_the_reference_debug_mode(orc):
    """cgmres.hpp:119-124 / matrix.hpp:76-81 (DEBUG_MODE): an output that is also an input ends the reference with
    exit(-1); the C ABI returns CGMRES_HIP_EINVAL."""
    import ctypes as C
    c = cg.CgmresBatch("pendulum", batch=2, dv=10, k_max=3)
    L = cg.load()
    buf = np.zeros((2, 30))
    ptr = buf.ctypes.data_as(C.c_void_p)
    assert L.cgmres_hip_control(c._h, ptr, ptr) == -1  # CGMRES_HIP_EINVAL
    assert b"same buffer" in L.cgmres_hip_last_error()
    assert L.cgmres_hip_F_func(c._h, ptr, ptr, ptr, C.c_double(0.0)) == -1
    assert L.cgmres_hip_Ax_func(c._h, ptr, ptr) == -1
    assert L.cgmres_hip_gmres(c._h, ptr, ptr) == -1
    c.close()
