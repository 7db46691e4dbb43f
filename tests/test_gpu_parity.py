"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(cgmres_cpp_amd -> libcgmres_hip.so), against
  * the committed golden fixtures produced by the unmodified reference (tests/golden/*.npz), and
  * the oracle (oracle/liboracle.so) on seeded inputs.
Tolerances are SURVEY.md §8(c)'s: fp64 teacher-forced single tick  |du|_inf <= 1e-9,
|dUdt|_inf <= 1e-7*max(1,|dUdt|_inf), Arnoldi count equal;  fp32: |du|_inf <= 1e-4 against the fp32 oracle."""
import numpy as np
import pytest

import cgmres_cpp_amd as cg
from conftest import golden_files, golden_ids, load_golden

pytestmark = pytest.mark.gpu

U_TOL = 1e-9
DUDT_REL = 1e-7
# 2 = "wg" (default mapping; in fp64 with row-parallel sweeps for the pendulum at 33 <= dv <= 53 — Newton on the trajectory —
#     and for the semi-active damper at dv <= 53 — scans only; tick_wg.hip.h: NWT),
# "2s" = the same with FLAG_SERIAL_STATE_SWEEP (the wg kernel with the serial quad sweep, where that differs),
# 1 = "lane" (reference statement order), 3 = "wg-lean" (two workgroups per CU),
# 4 = "wave" (one wavefront per controller: the latency mapping; pendulum fp64, dv <= 63, k_max <= 10)
SERIAL_STATE = "2s"
VARIANTS = [2, SERIAL_STATE, 1, 3, 4]


def dudt_close(a, b, rel=DUDT_REL):
    scale = max(1.0, float(np.max(np.abs(b))))
    return float(np.max(np.abs(a - b))) <= rel * scale


def new_batch(*a, **kw):
    """cg.CgmresBatch; a size / model the wave mapping does not serve skips the test case."""
    if kw.get("variant") == SERIAL_STATE:
        model = a[0] if a else kw.get("model")
        dv = kw.get("dv", 0)
        row_kernel = (model in (0, "pendulum") and 33 <= dv <= 53) or (model in (2, "semiactive") and dv <= 53)
        if not row_kernel or kw.get("dtype", "f64") != "f64":
            pytest.skip("FLAG_SERIAL_STATE_SWEEP only changes the fp64 kernels of the pendulum and the semi-active damper "
                        "at dim_u*dv <= 160")
        kw = dict(kw, variant=2, flags=kw.get("flags", 0) | cg.FLAG_SERIAL_STATE_SWEEP)
    try:
        return cg.CgmresBatch(*a, **kw)
    except cg.CgmresHipError as e:
        if kw.get("variant") == 4 and "wave mapping" in str(e):
            pytest.skip("the wave mapping does not cover this model / dtype / size")
        raise


def make_batch(case, batch, variant=0, tol=None):
    try:
        return new_batch(case["model"], batch=batch, dv=case["dv"], k_max=case["kmax"],
                         tol=case["tol"] if tol is None else tol, dtype=case["dtype"], variant=variant)
    except cg.CgmresHipError as e:
        if variant == 3 and "wg-lean mapping" in str(e):  # sizes beyond half a CU's LDS (e.g. fp64 with dim_u*dv = 300, k = 20)
            pytest.skip("lean LDS plan does not cover these sizes")
        raise


F64_FILES = [p for p in golden_files() if p.endswith("_f64.npz")]
F64_IDS = [i for i in golden_ids() if i.endswith("_f64")]
F32_FILES = [p for p in golden_files() if p.endswith("_f32.npz")]
F32_IDS = [i for i in golden_ids() if i.endswith("_f32")]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("path", F64_FILES, ids=F64_IDS)
def test_teacher_forced_control_vs_golden(path, variant):
    """(t, x, U, dUdt) of the reference's closed loop at recorded ticks -> u, U', dUdt', Arnoldi count."""
    g = load_golden(path)
    case = g["_case"]
    B = 3  # identical instances: also checks lanes do not interfere
    for tick in g["_ticks"]:
        p = f"tick{tick}_"
        c = make_batch(case, B, variant)
        c.set_ptau(g["ptau"])
        c.set_state(g[p + "t"][0], np.tile(g[p + "U"], (B, 1)), np.tile(g[p + "dUdt"], (B, 1)))
        u = c.control(np.tile(g[p + "x"], (B, 1)))
        t1, U1, d1 = c.get_state()
        n_ax, reason = c.get_status()
        for i in range(B):
            assert np.max(np.abs(u[i] - g[p + "u"])) <= U_TOL, (tick, u[i], g[p + "u"])
            assert np.max(np.abs(U1[i] - g[p + "U1"])) <= U_TOL
            assert dudt_close(d1[i], g[p + "dUdt1"]), (tick, np.max(np.abs(d1[i] - g[p + "dUdt1"])))
            assert n_ax[i] == int(g[p + "n_ax"][0]), (tick, n_ax, g[p + "n_ax"])
        assert abs(t1 - (g[p + "t"][0] + case_dt(c))) < 1e-15
        # Triangularised Hessenberg of the executed columns.  Only magnitudes are comparable: the reflector
        # sign is -sign(H(k,k)) (gmres.hpp:80) and H(k,k) can be a rounding-level number (seen: 2.8e-13 vs
        # 5.4e-13 around an exact zero), in which case the sign of the whole row of R flips harmlessly.
        # In fixed-k mode (tol = 0) the iterations that follow convergence orthogonalise rounding noise
        # (their H columns differ in the 2nd digit between ANY two builds while u/dUdt agree), so H is
        # only compared in the reference's early-exit mode, where every executed column has |residual| >= tol.
        if case["tol"] > 0:
            _, H, rho, gv = c.get_krylov()
            k = int(g[p + "n_ax"][0])
            scale = max(1.0, float(np.max(np.abs(g[p + "H"][:k, :k + 1]))))
            assert np.max(np.abs(np.abs(H[0][:k, :k + 1]) - np.abs(g[p + "H"][:k, :k + 1]))) <= 1e-6 * scale
            # reflectors (g0, g1, 2/(g0^2+g1^2)) of the executed columns, gmres.hpp:80-83 (magnitudes, same reason)
            gs = max(1.0, float(np.max(np.abs(g[p + "g"][:k]))))
            assert np.max(np.abs(np.abs(gv[0][:k]) - np.abs(g[p + "g"][:k]))) <= 1e-6 * gs, tick
            assert np.max(np.abs(np.abs(rho[0][:k + 1]) - np.abs(g[p + "rho"][:k + 1]))) <= \
                1e-6 * max(1.0, float(np.max(np.abs(g[p + "rho"][:k + 1])))), tick
        c.close()


def case_dt(c):
    return c.dt


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("path", F64_FILES, ids=F64_IDS)
def test_hooks_vs_golden(path, variant):
    """F_func, the pre-solve part of control (b), Ax_func and gmres as separate records."""
    g = load_golden(path)
    case = g["_case"]
    for tick in g["_ticks"]:
        p = f"tick{tick}_"
        c = make_batch(case, 2, variant)
        c.set_ptau(g["ptau"])
        U2, d2, x2 = np.tile(g[p + "U"], (2, 1)), np.tile(g[p + "dUdt"], (2, 1)), np.tile(g[p + "x"], (2, 1))
        c.set_state(g[p + "t"][0], U2, d2)
        F0 = c.F_func(U2, x2, g[p + "t"][0])
        fscale = max(1.0, float(np.max(np.abs(g[p + "F0"]))))
        assert np.max(np.abs(F0[1] - g[p + "F0"])) <= 1e-11 * fscale
        b = c.prepare(x2)
        bscale = max(1.0, float(np.max(np.abs(g[p + "b"]))))
        assert np.max(np.abs(b[0] - g[p + "b"])) <= 1e-7 * bscale  # (F - Fh)/h amplifies rounding by 1/h
        ax = c.Ax_func(np.tile(g[p + "Ax_v"], (2, 1)))
        ascale = max(1.0, float(np.max(np.abs(g[p + "Ax_out"]))))
        assert np.max(np.abs(ax[1] - g[p + "Ax_out"])) <= 1e-7 * ascale
        sol = c.gmres(d2, np.tile(g[p + "b"], (2, 1)))
        assert dudt_close(sol[0], g[p + "gmres_x"])
        assert c.get_status()[0][0] == int(g[p + "gmres_nax"][0])
        c.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("path", F64_FILES, ids=F64_IDS)
def test_closed_loop_batch_vs_golden(path, variant, orc):
    """The 8 seeded perturbed instances: Newton start + the first ticks of each closed loop.
    The plant step is taken with the golden u (teacher forcing on x) so chaos cannot accumulate."""
    g = load_golden(path)
    case = g["_case"]
    B = len(g["batch_x0"])
    c = make_batch(case, B, variant)
    c.set_ptau_repeat(g["batch_p"])
    c.init_u0(g["batch_u0_guess"])
    un = c.init_u0_newton(g["batch_u0_guess"], g["batch_x0"], g["batch_p"], 10)
    assert np.max(np.abs(un - g["batch_u0_newton"])) <= 1e-12
    x = g["batch_x0"].copy()
    n = min(g["batch_u"].shape[1], 20)
    for tick in range(n):
        u = c.control(x)
        assert np.max(np.abs(u - g["batch_u"][:, tick])) <= U_TOL * (1 + tick), tick
        assert np.array_equal(c.get_status()[0], g["batch_k"][:, tick]), tick
        x = g["batch_x"][:, tick].copy()
    c.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("path", F32_FILES, ids=F32_IDS)
def test_fp32_vs_fp32_reference(path, variant):
    """fp32 kernels against the fp32 build of the reference (`#define double float`), teacher-forced records:
    u, U', dUdt' and the Arnoldi count.  fp32 forward differences carry eps/h ~ 3e-5 relative noise, so dUdt' is
    compared at 5e-3 relative (a solve that runs all k_max iterations on the noise floor lands at 2e-4 .. 3.4e-3
    depending on the rounding of the build: seen at tick 0 of the dv = 100 case) and the count is not compared (the reference's own fp32-vs-fp64 counts
    differ on half the ticks)."""
    g = load_golden(path)
    case = g["_case"]
    for tick in g["_ticks"]:
        p = f"tick{tick}_"
        c = make_batch(case, 2, variant)
        c.set_ptau(g["ptau"])
        c.set_state(g[p + "t"][0], np.tile(g[p + "U"], (2, 1)), np.tile(g[p + "dUdt"], (2, 1)))
        u = c.control(np.tile(g[p + "x"], (2, 1)))
        _, U1, d1 = c.get_state()
        n_ax, _ = c.get_status()
        assert np.array_equal(u[0], u[1]) and n_ax[0] == n_ax[1]
        assert np.max(np.abs(u[0].astype(np.float64) - g[p + "u"])) <= 1e-4, (tick, u[0], g[p + "u"])
        assert np.max(np.abs(U1[0].astype(np.float64) - g[p + "U1"])) <= 1e-4, tick
        assert dudt_close(d1[0].astype(np.float64), g[p + "dUdt1"], rel=5e-3), \
            (tick, np.max(np.abs(d1[0] - g[p + "dUdt1"])), np.max(np.abs(g[p + "dUdt1"])))
        # the count itself is not comparable in fp32: once the residual estimate reaches the forward-difference noise
        # floor the exit test |rho_e| < 1e-6 is decided by rounding (seen: 8 here vs 20 in the fp32 reference at tick 0,
        # with u, U', dUdt' inside the bounds above)
        assert 1 <= int(n_ax[0]) <= case["kmax"]
        c.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("path", F32_FILES, ids=F32_IDS)
def test_fp32_seeded_batch_vs_fp32_reference(path, variant):
    """The 8 seeded perturbed instances in fp32: Newton start and the first closed-loop ticks of each, x teacher-forced
    from the fp32 reference's own trajectory."""
    g = load_golden(path)
    case = g["_case"]
    B = len(g["batch_x0"])
    c = make_batch(case, B, variant)
    c.set_ptau_repeat(g["batch_p"])
    c.init_u0(g["batch_u0_guess"])
    un = c.init_u0_newton(g["batch_u0_guess"], g["batch_x0"], g["batch_p"], 10)
    assert np.max(np.abs(un.astype(np.float64) - g["batch_u0_newton"])) <= 1e-5
    x = g["batch_x0"].copy()
    for tick in range(g["batch_u"].shape[1]):
        u = c.control(x)
        assert np.max(np.abs(u.astype(np.float64) - g["batch_u"][:, tick])) <= 1e-4 * (1 + tick), tick
        x = g["batch_x"][:, tick].copy()
    c.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dv,kmax", [(50, 10), (100, 20)])
def test_fp32_arnoldi_counts_where_fp32_noise_does_not_decide(orc, variant, dv, kmax):
    """fp32 Arnoldi counts cannot be compared one to one in general: once the residual estimate reaches the
    forward-difference noise floor (eps/h ~ 3e-5 relative) the exit test |rho_e[k+1]| < tol (gmres.hpp:93-95) is decided
    by rounding (seen at tick 0, where dtau = 0: 6 iterations here, 20 in the fp32 reference, with u / U' / dUdt' inside
    their bounds).  Where the noise does NOT decide they can: on every (tick, instance) for which the reference's own
    fp32 build, its fp64 build, and the fp32 build with a 10x looser and a 10x tighter tolerance all run the same number
    of iterations +-1 — all teacher-forced from the same state — the device's fp32 count must be within +-1 of it."""
    model, tol, B = 0, 1e-6, 16
    x0, u0, p = orc.batch_scenario(model, B)
    c = new_batch(model, batch=B, dv=dv, k_max=kmax, tol=tol, dtype="f32", variant=variant)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    mid = _oracle_batch(orc, model, dv, kmax, tol, x0, u0, p, "f32")
    others = [_oracle_batch(orc, model, dv, kmax, t2, x0, u0, p, dt2)
              for t2, dt2 in ((10 * tol, "f32"), (tol / 10, "f32"), (tol, "f64"))]
    x = x0.astype(np.float32).astype(np.float64)
    outside, decided = [], 0
    for tick in range(40):
        t_o, U_o, d_o = zip(*[r.get_state() for r in mid])
        c.set_state(t_o[0], np.array(U_o), np.array(d_o))
        c.control(x)
        n_ax, _ = c.get_status()
        for i, r in enumerate(mid):
            ks = []
            for grp in others:
                grp[i].set_state(t_o[i], U_o[i], d_o[i])
                grp[i].control(x[i])
                ks.append(grp[i].last_solve()[0])
            ur = r.control(x[i])
            k_ref = r.last_solve()[0]
            assert np.all(np.isfinite(ur)), (tick, i)
            if max(ks + [k_ref]) - min(ks + [k_ref]) <= 1:
                decided += 1
                if abs(int(n_ax[i]) - k_ref) > 1:
                    outside.append((tick, i, int(n_ax[i]), k_ref, ks))
            x[i] = (x[i] + r.plant(x[i], ur) * r.dt).astype(np.float32)
    c.close()
    assert not outside, outside[:8]
    # (the filter must leave something to check; at N = 100, k_max = 20 fp32 noise decides almost every exit of the first
    # 40 ticks — the fp64 build stops where the fp32 builds run on — and a handful of points are left)
    assert decided >= (30 if kmax == 10 else 1), decided


def _oracle_batch(orc, model, dv, kmax, tol, x0, u0, p, dtype="f64"):
    refs = []
    for i in range(len(x0)):
        r = orc.Controller(model, dv, kmax, tol, dtype)
        orc.start_controller(r, x0[i], u0[i], p[i])
        refs.append(r)
    return refs


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("model,dv,kmax,tol,B,ticks", [
    (0, 50, 10, 1e-6, 256, 12),   # BASELINE configs[1]: pendulum batch 256
    (0, 50, 10, 0.0, 130, 6),     # fixed-k mode, ragged batch (not a multiple of 64)
    (2, 50, 10, 1e-6, 192, 8),    # configs[2] shape, reduced batch
    (1, 50, 10, 1e-6, 96, 6),     # configs[3] Model1
    (1, 20, 5, 1e-6, 1, 30),      # configs[0]: single MSD controller
    (0, 25, 5, 1e-6, 65, 30),     # shipped pendulum sizes
])
def test_seeded_batch_vs_oracle(orc, model, dv, kmax, tol, B, ticks, variant):
    """Closed loop of a seeded perturbed batch; every tick is compared instance by instance with the
    oracle, then the oracle's x is adopted (teacher forcing) so both sides always see identical inputs."""
    x0, u0, p = orc.batch_scenario(model, B)
    c = new_batch(model, batch=B, dv=dv, k_max=kmax, tol=tol, variant=variant)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    refs = _oracle_batch(orc, model, dv, kmax, tol, x0, u0, p)
    x = x0.copy()
    for tick in range(ticks):
        t_o, U_o, d_o = zip(*[r.get_state() for r in refs])
        c.set_state(t_o[0], np.array(U_o), np.array(d_o))  # teacher forcing of the controller state too
        u = c.control(x)
        n_ax, reason = c.get_status()
        _, U1, d1 = c.get_state()
        for i, r in enumerate(refs):
            ur = r.control(x[i])
            assert np.max(np.abs(u[i] - ur)) <= U_TOL, (tick, i)
            assert n_ax[i] == r.last_solve()[0], (tick, i, n_ax[i], r.last_solve())
            assert reason[i] == r.last_solve()[2]
            assert dudt_close(d1[i], r.get_state()[2]), (tick, i)
            x[i] = x[i] + r.plant(x[i], ur) * r.dt
    c.close()


PAR_CASES = [
    # model, dv, kmax, tol, B, dtype
    (0, 50, 10, 1e-6, 40, "f64"),  # headline sizes: chunks of 12 stages, the direct chunk 14
    (0, 49, 10, 0.0, 33, "f64"),   # dv % 4 = 1
    (0, 47, 8, 1e-6, 17, "f64"),   # dv % 4 = 3
    (0, 64, 8, 1e-6, 23, "f64"),   # pipeline chunks 52 + 12: the longest two-chunk horizon (one-stage-per-lane coefficient tail)
    (0, 25, 6, 0.0, 21, "f64"),    # pipeline chunks 14 + 11: odd last chunk (its last pair has a first stage only)
    (0, 24, 6, 1e-6, 19, "f64"),   # pipeline chunks 12 + 12: both take the one-stage-per-lane path
    (0, 16, 5, 1e-6, 20, "f64"),   # dv % 4 = 0, chunks of 4 stages
    (0, 7, 3, 1e-6, 18, "f64"),    # chunks of ONE stage (the look-ahead of the stage loop runs past them)
    (0, 4, 3, 0.0, 5, "f64"),      # the shortest horizon the LDS-scratch form takes (the two-pass form wants dv >= 6)
    (2, 50, 10, 1e-6, 48, "f64"),  # semiactive: dim_x = 2 -> 32 transfer-matrix lanes per chunk, 8-scalar boundary records
    (2, 37, 6, 0.0, 19, "f64"),
    (1, 26, 6, 1e-6, 21, "f64"),   # MSD with a short horizon: constant Jacobian, NBW_LIN = 0 (no coefficient fetch at all)
    (1, 9, 4, 0.0, 16, "f64"),
    (1, 50, 10, 1e-6, 35, "f64"),  # MSD at BASELINE size: L = 300 -> the long-vector kernels (MAXM = 20, fh_hbm plan)
    (0, 100, 20, 1e-6, 19, "f64"),  # pendulum N = 100: long vectors, four pipeline chunks in the state sweep
    (0, 100, 20, 1e-6, 33, "f32"),  # BASELINE configs[4] shape
    (0, 53, 12, 0.0, 16, "f64"),   # dim_u*dv = 159, k = 12: no room for the LDS-scratch form -> two-pass by default
]
# how the parallel form is asked for -> (variant, flags, the name the handle must resolve to; None = whatever fits)
PAR_FORMS = {
    "lds-scratch": (2, "SERIAL_STATE", None),  # (with the serial state sweep: the row-Newton kernel has no costate sweep of its own in the loop)
    "two-pass": (2, "TWO_PASS", "wg+two-pass-costate"),
    "lean-two-pass": (3, 0, "wg-lean+two-pass-costate"),
}


@pytest.mark.parametrize("form", list(PAR_FORMS))
@pytest.mark.parametrize("model,dv,kmax,tol,B,dtype", PAR_CASES)
def test_chunk_parallel_costate_vs_serial_and_oracle(orc, model, dv, kmax, tol, B, dtype, form):
    """The chunk-parallel costate sweeps — the form with per-stage LDS scratch (full plan, short
    vectors), the two-pass form on the full plans (short and long vectors, F(U,x+hf,t+h) in LDS or in HBM) and on the
    lean plan — against the oracle AND against the serial sweep of the same library (flags=FLAG_SERIAL_COSTATE picks
    the serial kernel instantiation of the same plan) at horizon lengths around the chunking's edge cases.
    Teacher-forced ticks, early exits included (tol > 0)."""
    variant, flags, want_name = PAR_FORMS[form]
    flags = {"TWO_PASS": cg.FLAG_TWO_PASS_COSTATE, "SERIAL_STATE": cg.FLAG_SERIAL_STATE_SWEEP}.get(flags, flags)
    f32 = dtype == "f32"
    x0, u0, p = orc.batch_scenario(model, B)
    try:
        par = cg.CgmresBatch(model, batch=B, dv=dv, k_max=kmax, tol=tol, dtype=dtype, variant=variant, flags=flags)
    except cg.CgmresHipError as e:
        assert variant == 3 and "wg-lean mapping" in str(e)
        pytest.skip("lean LDS plan does not cover these sizes")
    ser = cg.CgmresBatch(model, batch=B, dv=dv, k_max=kmax, tol=tol, dtype=dtype, variant=variant,
                         flags=cg.FLAG_SERIAL_COSTATE)
    assert ser.variant_name in ("wg", "wg-lean")
    if par.variant_name == ser.variant_name:
        par.close(), ser.close()
        pytest.skip(f"no parallel form for these sizes on this plan ({par.variant_name})")
    if form == "lds-scratch":
        # L <= 160 with room for the scratch: the LDS-scratch form; otherwise the library's own fallback, two-pass
        assert par.variant_name in ("wg+parallel-costate", "wg+two-pass-costate")
    else:
        assert par.variant_name == want_name
    refs = _oracle_batch(orc, model, dv, kmax, tol, x0, u0, p, dtype) if not f32 else None
    for c in (par, ser):
        c.set_ptau_repeat(p)
        c.init_u0(u0)
        c.init_u0_newton(u0, x0, p, 10)
    x = x0.copy()
    # fp32: the solve runs on the forward-difference noise floor (eps/h ~ 3e-5 relative): two association orders of the
    # same sweep land up to ~1e-4 apart on u — each within the fp32 parity bar of the oracle (test_gpu_closed_loop.py)
    tol_ps = 3e-4 if f32 else 1e-11
    for tick in range(6):
        if f32:
            t_s, U_s0, d_s0 = ser.get_state()  # fp32: both forms from the SERIAL form's state (no fp32 oracle loop here)
            state = (t_s, U_s0, d_s0)
        else:
            t_o, U_o, d_o = zip(*[r.get_state() for r in refs])
            state = (t_o[0], np.array(U_o), np.array(d_o))
        out = []
        for c in (par, ser):
            c.set_state(*state)
            u = c.control(x)
            out.append((u, c.get_status(), c.get_state()))
        (u_p, (k_p, why_p), (_, U_p, d_p)), (u_s, (k_s, why_s), (_, U_s, d_s)) = out
        assert np.max(np.abs(u_p.astype(float) - u_s)) <= tol_ps, (tick, np.max(np.abs(u_p.astype(float) - u_s)))
        assert np.max(np.abs(U_p.astype(float) - U_s)) <= tol_ps, tick
        if f32:  # (fp32: the exit test sits in the forward-difference noise: the Arnoldi counts of two association
            # orders are not comparable — seen 11 vs 15 with u equal to 1e-4, SURVEY.md §7.3 — only u and U are)
            x = x + np.array([orc.Controller(model, dv, kmax, tol).plant(x[i], u_s[i].astype(float)) for i in range(B)]) * 1e-3
            continue
        assert np.array_equal(k_p, k_s) and np.array_equal(why_p, why_s), tick
        for i, r in enumerate(refs):
            ur = r.control(x[i])
            assert np.max(np.abs(u_p[i] - ur)) <= U_TOL, (tick, i)
            assert k_p[i] == r.last_solve()[0] and why_p[i] == r.last_solve()[2], (tick, i)
            assert dudt_close(d_p[i], r.get_state()[2]), (tick, i)
            x[i] = x[i] + r.plant(x[i], ur) * r.dt
    par.close(), ser.close()


def test_eight_instances_per_workgroup_on_request(orc):
    """flags = FLAG_IPW8: the 8-instance workgroups (otherwise taken only when 16 instances do not fit the LDS) on the
    headline sizes, closed loop with early exits against the oracle."""
    model, dv, km, B, n = 0, 50, 10, 100, 12
    x0, u0, p = orc.batch_scenario(model, B)
    c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=km, tol=1e-6, variant=2, flags=cg.FLAG_IPW8)
    assert c.variant_name == "wg"
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    refs = _oracle_batch(orc, model, dv, km, 1e-6, x0, u0, p)
    x = x0.copy()
    for tick in range(n):
        u = c.control(x)
        n_ax, reason = c.get_status()
        for i, r in enumerate(refs):
            ur = r.control(x[i])
            assert np.max(np.abs(u[i] - ur)) <= U_TOL, (tick, i)
            assert n_ax[i] == r.last_solve()[0] and reason[i] == r.last_solve()[2], (tick, i)
            x[i] = x[i] + r.plant(x[i], ur) * r.dt
    c.close()


def test_chunk_parallel_costate_form_follows_the_lds_budget():
    """Which costate sweep a handle gets: the LDS-scratch form where its 23.5 KB fit (headline), the two-pass form with
    4 chunks where only boundary records fit (dim_u*dv = 159, k = 12; long vectors; the lean plans of the pendulum),
    the serial sweep on request or when not even those fit."""
    for kw, want in ((dict(model=0, dv=50, k_max=10, variant=2), "wg+row-newton"),
                     (dict(model=0, dv=50, k_max=10, variant=2, flags=cg.FLAG_SERIAL_STATE_SWEEP), "wg+parallel-costate"),
                     (dict(model=0, dv=40, k_max=10, variant=2), "wg+row-newton"),
                     (dict(model=0, dv=54, k_max=10, variant=2), "wg+two-pass-costate"),  # (dim_u*dv > 160: the long-vector kernels)
                     (dict(model=0, dv=50, k_max=10, variant=2, dtype="f32"), "wg+parallel-costate"),
                     (dict(model=0, dv=53, k_max=12, variant=2), "wg+row-newton"),
                     (dict(model=0, dv=53, k_max=12, variant=2, flags=cg.FLAG_SERIAL_STATE_SWEEP), "wg+two-pass-costate"),
                     (dict(model=0, dv=30, k_max=10, variant=2), "wg+parallel-costate"),  # (short horizons: the serial sweep is cheaper)
                     (dict(model=1, dv=50, k_max=10, variant=2), "wg+two-pass-costate"),
                     (dict(model=0, dv=50, k_max=10, variant=3), "wg-lean+two-pass-costate"),
                     (dict(model=0, dv=100, k_max=20, variant=3, dtype="f32"), "wg-lean+two-pass-costate"),
                     (dict(model=1, dv=50, k_max=10, variant=3), "wg-lean"),   # 182 scalars of LDS left: the records of three chunks need 416
                     (dict(model=0, dv=50, k_max=10, variant=2, flags=cg.FLAG_SERIAL_COSTATE), "wg"),
                     (dict(model=0, dv=5, k_max=3, variant=3), "wg-lean")):
        c = cg.CgmresBatch(batch=16, tol=0.0, **kw)
        assert c.variant_name == want, (kw, c.variant_name)
        c.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_large_angles_take_the_library_trig_path(orc, variant):
    """Angles beyond the fast range of the device trig kernel (|arg| >= 1e5) make the wg sweep redo the affected
    chunk of stages with the library sin/cos.  A batch that mixes such instances with ordinary ones (same
    workgroup, so the redo also covers the ordinary ones) must still match the oracle instance by instance."""
    model, dv, kmax, tol, B = 0, 50, 10, 1e-6, 24
    x0, u0, p = orc.batch_scenario(model, B)
    two_pi = 2.0 * np.pi
    x0[1, 0] += 20000 * two_pi   # x0 - x1 out of range from the first stage on
    x0[5, 1] -= 30000 * two_pi   # both angles out of range
    x0[17, 0] += 15915 * two_pi  # just below 1e5: crosses the cut-over only if the sweep moves it
    x0[17, 1] += 15915 * two_pi
    c = new_batch(model, batch=B, dv=dv, k_max=kmax, tol=tol, variant=variant)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    refs = _oracle_batch(orc, model, dv, kmax, tol, x0, u0, p)
    x = x0.copy()
    for tick in range(3):
        t_o, U_o, d_o = zip(*[r.get_state() for r in refs])
        c.set_state(t_o[0], np.array(U_o), np.array(d_o))
        u = c.control(x)
        assert np.all(np.isfinite(u))
        _, U1, d1 = c.get_state()
        for i, r in enumerate(refs):
            ur = r.control(x[i])
            if i in (1, 5, 17):
                # |angle| ~ 1e5 has an absolute resolution of 1.5e-11 rad and these instances sit ~1e5 rad from
                # their target (|u| ~ 1e3): compare at 1e-7 relative — the point is that the library path is
                # taken and agrees, not the conditioning of an absurd scenario
                assert np.max(np.abs(u[i] - ur)) <= 1e-7 * max(1.0, np.max(np.abs(ur))), (tick, i, u[i], ur)
            else:
                assert np.max(np.abs(u[i] - ur)) <= U_TOL, (tick, i, u[i], ur)
                assert dudt_close(d1[i], r.get_state()[2]), (tick, i)
            x[i] = x[i] + r.plant(x[i], ur) * r.dt
    c.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_fast_angular_rates_leave_the_rotation_range(orc, variant):
    """The wg state sweep advances sin/cos by rotating the previous stage's values through the angle increment; an
    increment beyond its range (|d| > 0.04 rad per stage) makes the sweep redo that chunk of stages with fresh
    evaluations.  A late horizon (t = 1 s: dtau = 3.9 ms) and angular rates of tens of rad/s force that path for some
    instances of a workgroup (the redo covers the whole workgroup) — every instance must still match the oracle."""
    model, dv, kmax, tol, B = 0, 50, 10, 1e-6, 40
    x0, u0, p = orc.batch_scenario(model, B)
    x0[2, 2], x0[2, 3] = 30.0, -20.0     # |d(x0-x1)| = dtau*50 = 0.2 per stage
    x0[9, 3] = 25.0                      # x1 alone moves 0.1 per stage
    x0[21, 2], x0[21, 3] = -12.0, 9.0    # 0.08: beyond the range as well
    x0[33, 2] = 8.0                      # 0.03: stays inside
    c = new_batch(model, batch=B, dv=dv, k_max=kmax, tol=tol, variant=variant)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    refs = _oracle_batch(orc, model, dv, kmax, tol, x0, u0, p)
    for r in refs:                       # a late horizon from the first tick on
        _, U_o, d_o = r.get_state()
        r.set_state(1.0, U_o, d_o)
    x = x0.copy()
    for tick in range(3):
        t_o, U_o, d_o = zip(*[r.get_state() for r in refs])
        c.set_state(t_o[0], np.array(U_o), np.array(d_o))
        u = c.control(x)
        assert np.all(np.isfinite(u))
        n_ax, _ = c.get_status()
        _, U1, d1 = c.get_state()
        for i, r in enumerate(refs):
            ur = r.control(x[i])
            scale = max(1.0, float(np.max(np.abs(ur))))
            assert np.max(np.abs(u[i] - ur)) <= U_TOL * scale, (tick, i, u[i], ur)
            assert dudt_close(d1[i], r.get_state()[2]), (tick, i)
            assert n_ax[i] == r.last_solve()[0], (tick, i)
            x[i] = x[i] + r.plant(x[i], ur) * r.dt
    c.close()


def test_closed_loop_device_matches_host_loop(orc):
    """closed_loop_device (plant on the GPU, device pointers) == host-driven loop with the same plant rule."""
    B, dv, km, n = 70, 50, 10, 8
    x0, u0, p = orc.batch_scenario(0, B)

    def start():
        c = cg.CgmresBatch("pendulum", batch=B, dv=dv, k_max=km)
        c.set_ptau_repeat(p)
        c.init_u0(u0)
        c.init_u0_newton(u0, x0, p, 10)
        return c
    a, b = start(), start()
    xd = a.device_buffer((B, 4)).upload(x0)
    ud = a.device_buffer((B, 3))
    a.closed_loop_device(xd, ud, n)
    a.synchronize()
    xa, ua = xd.download(), ud.download()
    refs = [orc.Controller(0, dv, km) for _ in range(B)]
    x = x0.copy()
    for tick in range(n):
        u = b.control(x)
        for i in range(B):
            x[i] = x[i] + refs[i].plant(x[i], u[i]) * b.dt
    assert np.max(np.abs(ua - u)) <= 1e-8 and np.max(np.abs(xa - x)) <= 1e-10
    assert abs(a.t - b.t) < 1e-15 and abs(a.t - n * a.dt) < 1e-12
    xd.free(), ud.free(), a.close(), b.close()


@pytest.mark.parametrize("model,name", [(0, "pendulum"), (1, "msd"), (2, "semiactive")])
def test_full_size_properties(orc, model, name):
    """BASELINE's sizes (B = 4096, dv = 50, k_max = 10; MSD runs the fh_hbm plan of the long-vector kernel): properties
    that need no full oracle run.
      * a sample of 48 instances spread over the batch agrees with the oracle;
      * instances with identical inputs give bit-identical outputs wherever they sit in the batch;
      * with tol=0 every instance runs exactly k_max Arnoldi iterations; the leading Krylov vectors are orthonormal."""
    B, dv, km = 4096, 50, 10
    x0, u0, p = orc.batch_scenario(model, B)
    x0[1000], p[1000] = x0[3], p[3]      # duplicates far apart: different waves / workgroups
    x0[4095], p[4095] = x0[3], p[3]
    c = cg.CgmresBatch(name, batch=B, dv=dv, k_max=km, tol=0.0)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    sample = list(range(0, B, 89))[:46] + [1000, 4095]
    refs = {i: orc.Controller(model, dv, km, 0.0) for i in sample}
    for i, r in refs.items():
        orc.start_controller(r, x0[i], u0[i], p[i])
    x = x0.copy()
    for tick in range(3):
        u = c.control(x)
        n_ax, reason = c.get_status()
        assert np.all(n_ax == km) and np.all(reason == cg.EXIT_NATURAL)
        assert np.array_equal(u[3], u[1000]) and np.array_equal(u[3], u[4095])
        for i, r in refs.items():
            assert np.max(np.abs(u[i] - r.control(x[i]))) <= U_TOL * (1 + 10 * tick), (tick, i)
        assert np.all(np.isfinite(u))
        x = x + 0.0  # plant frozen: the controller still advances t and U
    V, H, rho, g = c.get_krylov(with_V=True)
    for i in (0, 777, 4095):
        G = V[i][:4] @ V[i][:4].T  # leading columns only: later ones are built on a converged residual
        assert np.max(np.abs(G - np.eye(4))) < 1e-8
    c.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_status_exit_paths(orc, variant):
    """Edge cases of gmres.hpp on the GPU, instance by instance against the oracle:
      * ||r0|| < tol (:39-41): dUdt untouched, no Arnoldi step;
      * convergence in the FIRST column (:93-95 with k = 0): the k loop breaks without incrementing k, so the
        triangular solve has size 0 and dUdt is untouched as well although one mat-vec ran;
      * breakdown |h(k+1,k)| < DBL_EPSILON (:63-65): dUdt untouched."""
    B, dv, km = 37, 8, 3
    x0, u0, p = orc.batch_scenario(2, B)

    def both(tol, u_init=None, x0=x0):
        c = new_batch("semiactive", batch=B, dv=dv, k_max=km, tol=tol, variant=variant)
        ui = u0 if u_init is None else u_init
        c.init_u0(ui)
        refs = []
        for i in range(B):
            r = orc.Controller(2, dv, km, tol)
            r.init_u0(ui[i])
            refs.append(r)
        if u_init is None:
            c.init_u0_newton(u0, x0, None, 10)
            for i, r in enumerate(refs):
                r.init_u0_newton(u0[i], x0[i], p[i], 10)
        _, U0, d0 = c.get_state()
        u = c.control(x0)
        n_ax, reason = c.get_status()
        _, U1, d1 = c.get_state()
        c.close()
        for i, r in enumerate(refs):
            ur = r.control(x0[i])
            k_o, ks_o, reason_o = r.last_solve()
            assert n_ax[i] == k_o and reason[i] == reason_o, (tol, i, n_ax[i], k_o, reason[i], reason_o)
            if np.all(np.isfinite(ur)):
                assert np.max(np.abs(u[i] - ur)) <= U_TOL * max(1.0, float(np.max(np.abs(ur)))), (tol, i)
        return U0, d0, U1, d1, n_ax, reason, refs

    # (1) huge tol: every instance leaves at the residual test
    U0, d0, U1, d1, n_ax, reason, _ = both(1e30)
    assert np.all(reason == cg.EXIT_SMALL_RESIDUAL) and np.all(n_ax == 0)
    assert np.array_equal(d0, d1) and np.allclose(U1, U0 + d0 * 1e-3)

    # (2) tol between ||r0|| and |rho_e[1]|: picked per batch from the oracle's own numbers
    r0n, e1 = [], []
    for i in range(B):
        r = orc.Controller(2, dv, 1, 0.0)
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
    assert hit.sum() >= B // 2, (hit.sum(), tol)
    for i in np.nonzero(hit)[0]:
        assert refs[i].last_solve()[1] == 0            # the oracle solved a 0 x 0 system
        assert np.array_equal(d0[i], d1[i])            # dUdt untouched
    assert np.allclose(U1[hit], U0[hit] + d0[hit] * 1e-3)

    # (3) breakdown: with |U| = 1e17 the perturbation h*v (|v| <= 1) is absorbed by the rounding of U + h*v
    # (ulp(1e17) = 16 > h), so F(U + h v0) == F(U) bit for bit and A*v0 = (F(U+hv0) - F(U))/h = 0, h(1,0) = 0 <
    # DBL_EPSILON.  The plant rests at x = 0 so the costate is identically zero: the wg mapping's regrouped sum
    # (phi - Fh)/h + B^T lambda / h is then exact as well (with lambda != 0 it leaves |F| eps / h of rounding there,
    # which is the documented association difference, not a breakdown).
    big = np.full((B, 3), 1e17)
    U0, d0, U1, d1, n_ax, reason, _ = both(0.0, u_init=big, x0=np.zeros_like(x0))
    assert np.all(reason == cg.EXIT_BREAKDOWN) and np.all(n_ax == 1)
    assert np.array_equal(d0, d1) and np.array_equal(d1, np.zeros_like(d1))
    assert np.array_equal(U1, U0)


@pytest.mark.parametrize("variant", VARIANTS)
def test_nonfinite_state_is_flagged_per_instance(orc, variant):
    """CGMRES_HIP_EXIT_NONFINITE: a NaN / Inf plant state makes ||r0|| non-finite.  The reference has no test for it —
    every comparison with a NaN is false (gmres.hpp:39-41, 63-65, 93-95 fall through) and NaNs end up in dUdt, U and u;
    the device ends with the same NaNs, stops that instance at once and SAYS so in the status.  The other instances of
    the same workgroup are untouched (instances never exchange data)."""
    model, dv, km, B = 0, 50, 10, 40
    x0, u0, p = orc.batch_scenario(model, B)
    bad = {3: np.nan, 20: np.inf, 37: -np.inf}
    x = x0.copy()
    for i, v in bad.items():
        x[i, i % 2] = v
    c = new_batch(model, batch=B, dv=dv, k_max=km, tol=1e-6, variant=variant)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    for tick in range(2):
        u = c.control(x)
        n_ax, reason = c.get_status()
        _, U, d = c.get_state()
        for i in range(B):
            r = orc.Controller(model, dv, km, 1e-6)
            orc.start_controller(r, x0[i], u0[i], p[i])
            for _ in range(tick + 1):
                ur = r.control(x[i])
            if i in bad:
                assert reason[i] == cg.EXIT_NONFINITE and n_ax[i] == 0, (tick, i, reason[i], n_ax[i])
                assert np.all(np.isnan(ur))                       # the reference's fall-through: NaN everywhere
                assert np.all(np.isnan(d[i])) and np.all(np.isnan(U[i])) and np.all(np.isnan(u[i]))
            else:
                assert reason[i] == r.last_solve()[2] and n_ax[i] == r.last_solve()[0], (tick, i)
                assert np.max(np.abs(u[i] - ur)) <= U_TOL, (tick, i)
    c.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("model,dv,kmax", [(0, 50, 10), (1, 50, 10), (2, 50, 10)])
def test_ax_func_after_control(orc, model, dv, kmax, variant):
    """The facade's Ax_func (cgmres.hpp:164-175) after control(): x_dxh and F_dxh_h are the ones control() left
    behind (cgmres.hpp:85,88), U and t the advanced ones — exactly the members the reference would read.  Both
    mappings must agree with the oracle (MSD at dv = 50 runs the fh_hbm plan of the wg mapping)."""
    B = 19
    x0, u0, p = orc.batch_scenario(model, B)
    c = new_batch(model, batch=B, dv=dv, k_max=kmax, variant=variant)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    refs = _oracle_batch(orc, model, dv, kmax, 1e-6, x0, u0, p)
    x = x0.copy()
    rng = np.random.default_rng(3)
    for tick in range(3):
        u = c.control(x)
        v = rng.standard_normal((B, c.len))
        ax = c.Ax_func(v)
        for i, r in enumerate(refs):
            ur = r.control(x[i])
            ar = r.Ax(v[i])
            assert np.max(np.abs(u[i] - ur)) <= U_TOL * (1 + tick)
            assert np.max(np.abs(ax[i] - ar)) <= 1e-7 * max(1.0, float(np.max(np.abs(ar)))), (tick, i)
            x[i] = x[i] + r.plant(x[i], ur) * r.dt
    c.close()


def test_device_sincos_accuracy():
    """The fp64 sin/cos of the horizon sweeps against the host libm: <= 2 ulp of the result on the ranges the
    models visit (angles around pi, arguments up to the 1e5 cut-over to the library path) and beyond it."""
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(-10, 10, 20000), rng.uniform(-1e5, 1e5, 20000), rng.uniform(-1e9, 1e9, 2000),
                        np.pi * np.arange(-8, 9) / 2, [0.0, 1e-300, -1e-20, 3.14159265358979, 0.785398163397448]])
    s, c = cg.selftest_sincos(a)
    for got, ref in ((s, np.sin(a)), (c, np.cos(a))):
        err = np.abs(got - ref)
        # 2 ulp of the result, plus the absolute error of the two-constant argument reduction
        # (|pi/2 - (PIO2_HI + PIO2_LO)| ~ 6.5e-27 per multiple of pi/2), which only shows next to a zero
        bound = 2.0 * np.spacing(np.abs(ref)) + 1e-26 * np.maximum(1.0, np.abs(a))
        assert np.all(err <= bound), float(np.max(err / bound))


def test_variant_resolution():
    a = cg.CgmresBatch("pendulum", batch=4, dv=50, k_max=10)
    assert a.variant == 4          # up to two controllers per SIMD: the wave mapping (tests/test_gpu_wave.py)
    w = cg.CgmresBatch("pendulum", batch=4096, dv=50, k_max=10)
    assert w.variant == 2          # the headline batch: one 16-instance workgroup per CU
    w.close()
    # more 16-instance workgroups than CUs: the default becomes the lean LDS plan (two workgroups per CU) ...
    big = cg.CgmresBatch("pendulum", batch=8192, dv=100, k_max=20, dtype="f32")
    assert big.variant == 3
    # ... an explicit 2 / 3 is honoured at any size the plan supports
    e2 = cg.CgmresBatch("pendulum", batch=8192, dv=100, k_max=20, dtype="f32", variant=2)
    e3 = cg.CgmresBatch("pendulum", batch=4, dv=50, k_max=10, variant=3)
    assert e2.variant == 2 and e3.variant == 3
    big.close(), e2.close(), e3.close()
    b = cg.CgmresBatch("pendulum", batch=4, dv=50, k_max=10, variant=1)
    assert b.variant == 1
    c = cg.CgmresBatch("msd", batch=4, dv=200, k_max=5)  # dim_u*dv = 1200: beyond the wg mapping -> lane
    assert c.variant == 1
    with pytest.raises(cg.CgmresHipError):
        cg.CgmresBatch("msd", batch=4, dv=200, k_max=5, variant=2)
    a.close(), b.close(), c.close()


def test_model_probe_matches_oracle_models(orc):
    """Registry fingerprint: device dxdt/dPhidx/dHdx/dHdu at a probe point == the restated models (via F machinery
    is overkill: compare with finite evaluations through the oracle's plant and a one-stage horizon)."""
    rng = np.random.default_rng(5)
    for model in (0, 1, 2):
        mi = cg.model_info(model)
        x, u = rng.standard_normal(mi["dim_x"]), rng.standard_normal(mi["dim_u"])
        p, l = rng.standard_normal(max(mi["dim_p"], 1))[:mi["dim_p"]], rng.standard_normal(mi["dim_x"])
        f, gphi, hx, hu = cg.model_probe(model, x, u, p, l)
        r = orc.Controller(model, 8, 3)
        assert np.max(np.abs(f - r.plant(x, u))) <= 1e-12
        assert np.all(np.isfinite(np.concatenate([gphi, hx, hu])))


def test_multiple_controller_mixed_batch(orc):
    """BASELINE configs[3] shape: a Model1 (MSD) batch and a Model2 (pendulum) batch stepped in the same loop on
    separate streams (cgmres_cpp_amd.multi.MultipleController, multiple_controller/main.cpp:104-110), each against
    the oracle, teacher-forced."""
    from cgmres_cpp_amd.multi import MultipleController
    specs = [dict(model="msd", batch=40, dv=50, k_max=10), dict(model="pendulum", batch=56, dv=50, k_max=10)]
    mc = MultipleController(specs)
    refs, xs = [], []
    for m, model in zip(mc.members, (1, 0)):
        x0, u0, p = orc.batch_scenario(model, m.batch)
        m.set_ptau_repeat(p)
        m.init_u0(u0)
        m.init_u0_newton(u0, x0, p, 10)
        refs.append(_oracle_batch(orc, model, 50, 10, 1e-6, x0, u0, p))
        xs.append(x0.copy())
    for tick in range(5):
        us = mc.control(xs)
        for k, (m, rs) in enumerate(zip(mc.members, refs)):
            n_ax, _ = m.get_status()
            for i, r in enumerate(rs):
                ur = r.control(xs[k][i])
                assert np.max(np.abs(us[k][i] - ur)) <= U_TOL * (1 + tick), (tick, k, i)
                assert n_ax[i] == r.last_solve()[0]
                xs[k][i] = xs[k][i] + r.plant(xs[k][i], ur) * r.dt
    mc.close()  # (the device-pointer path on two streams: tests/test_gpu_closed_loop.py)
