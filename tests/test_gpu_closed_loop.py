"""GPU parity of the BENCHMARKED entry point: cgmres_hip_closed_loop_device on the wg mapping — several control
ticks fused per launch (CGMRES_HIP_TICKS_PER_LAUNCH), the controller state carried on chip between them and handed
over through HBM between launches — against the oracle's free-running closed loop with the example's Euler plant
(<example>/main.cpp:63-73), from the same seeded start.

n in {10, 11, 25, 60} = exactly one full fuse, one launch boundary + a 1-tick tail, two boundaries + a partial
tail, five boundaries.  Sizes are BASELINE.json's (B = 4096 / 8192); the oracle runs on a spread sample.
Tolerances: SURVEY.md §8(c) closed loop, first 100 ticks: |du|_inf <= 1e-9 (fp64); fp32 against the fp32 oracle 1e-4.
"""
import numpy as np
import pytest

import cgmres_cpp_amd as cg
from cgmres_cpp_amd import scenarios

pytestmark = pytest.mark.gpu

N_TICKS = (10, 11, 25, 60)

# id -> (model, dv, kmax, tol, dtype, B, variant)
CASES = {
    "pendulum_B4096_fixedk": (0, 50, 10, 0.0, "f64", 4096, 2),      # the bench.py headline mode
    "pendulum_B4096_tolref": (0, 50, 10, 1e-6, "f64", 4096, 2),     # bench.py's reference_mode leg
    "msd_B4096_fh_hbm": (1, 50, 10, 1e-6, "f64", 4096, 2),          # MAXM = 20 kernel, F(U,x+hf,t+h) kept in HBM
    "msd_B4096_fh_hbm_fixedk": (1, 50, 10, 0.0, "f64", 4096, 2),
    "semiactive_B4096": (2, 50, 10, 1e-6, "f64", 4096, 2),          # BASELINE configs[2]
    "semiactive_B4096_fixedk": (2, 50, 10, 0.0, "f64", 4096, 2),
    "pendulum_f32_N100_k20_B8192": (0, 100, 20, 1e-6, "f32", 8192, 2),  # configs[4] per-GPU share, one workgroup per CU
    "pendulum_f32_N100_k20_B8192_lean": (0, 100, 20, 1e-6, "f32", 8192, 3),  # ... and as the library runs it by default
    "pendulum_B4096_lean": (0, 50, 10, 1e-6, "f64", 4096, 3),       # the lean LDS plan (two workgroups per CU): configs[3]
    "pendulum_B4096_lean_fixedk": (0, 50, 10, 0.0, "f64", 4096, 3),
    "msd_B4096_lean": (1, 50, 10, 1e-6, "f64", 4096, 3),            # members, MultipleController's mapping
    "semiactive_B8192_lean": (2, 50, 10, 1e-6, "f64", 8192, 3),
    "pendulum_B256_wave": (0, 50, 10, 1e-6, "f64", 256, 4),         # BASELINE configs[1] on the latency mapping (one wave per controller)
    "pendulum_B512_wave_fixedk": (0, 50, 10, 0.0, "f64", 512, 4),   # the per-GPU shard of the headline batch at 8 GPUs
    "pendulum_B4096_wave": (0, 50, 10, 1e-6, "f64", 4096, 4),       # the wave mapping oversubscribed (several waves per SIMD / rounds)
    "pendulum_dv25_k5_B67_wave": (0, 25, 5, 1e-6, "f64", 67, 4),    # shipped sizes, ragged batch
    "semiactive_B512_wave": (2, 50, 10, 1e-6, "f64", 512, 4),       # the wave mapping for a model that is affine in x
    "semiactive_B300_wave_fixedk": (2, 50, 10, 0.0, "f64", 300, 4),
    "msd_B512_wave": (1, 50, 10, 1e-6, "f64", 512, 4),              # linear time-invariant: constant-matrix scans (L = 300: 6 elements per lane)
    "msd_dv20_k5_B33_wave": (1, 20, 5, 0.0, "f64", 33, 4),          # BASELINE configs[0] sizes, ragged batch
    "pendulum_B200_lane": (0, 50, 10, 1e-6, "f64", 200, 1),         # the lane mapping: one tick per launch
    "msd_dv20_k5_B33": (1, 20, 5, 1e-6, "f64", 33, 2),              # ragged batch, shipped-size MSD (IPW rows unused)
}


def sample_of(B, n=46):
    step = max(1, B // n)
    s = list(range(0, B, step))[:n]
    for extra in (B - 1, B // 2 + 1, 15, 16):  # last instance, a mid one, both sides of a workgroup edge
        if 0 <= extra < B and extra not in s:
            s.append(extra)
    return sorted(s)


class OracleLoop:
    """Free-running closed loops of the sampled instances on the oracle, with snapshots at the requested ticks."""

    def __init__(self, orc, model, dv, kmax, tol, dtype, x0, u0, p, sample, snaps):
        self.f32 = dtype == "f32"
        self.snap = {}
        npdt = np.float32 if self.f32 else np.float64
        ctrls, xs = [], []
        for i in sample:
            c = orc.Controller(model, dv, kmax, tol, dtype)
            orc.start_controller(c, x0[i], u0[i], p[i])
            ctrls.append(c)
            xs.append(np.array(x0[i], dtype=npdt))
        u_last = [None] * len(sample)
        for tick in range(1, max(snaps) + 1):
            for j, c in enumerate(ctrls):
                u = c.control(xs[j])
                f = c.plant(xs[j], u)
                # plant step in the controller's precision (the device does x + f*dt in T)
                xs[j] = (xs[j] + f.astype(npdt) * npdt(c.dt)).astype(npdt)
                u_last[j] = u
            if tick in snaps:
                self.snap[tick] = dict(
                    x=np.array(xs, dtype=np.float64), u=np.array(u_last),
                    state=[c.get_state() for c in ctrls], solve=[c.last_solve() for c in ctrls])


_oracle_cache = {}


def oracle_for(orc, name):
    if name not in _oracle_cache:
        model, dv, kmax, tol, dtype, B, _ = CASES[name]
        x0, u0, p = orc.batch_scenario(model, B)
        sample = sample_of(B)
        _oracle_cache[name] = (x0, u0, p, sample,
                               OracleLoop(orc, model, dv, kmax, tol, dtype, x0, u0, p, sample, set(N_TICKS)))
    return _oracle_cache[name]


@pytest.mark.parametrize("n", N_TICKS)
@pytest.mark.parametrize("name", list(CASES))
def test_closed_loop_device_vs_oracle(orc, name, n):
    model, dv, kmax, tol, dtype, B, variant = CASES[name]
    x0, u0, p, sample, ol = oracle_for(orc, name)
    f32 = dtype == "f32"
    npdt = np.float32 if f32 else np.float64
    c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=kmax, tol=tol, dtype=dtype, variant=variant)
    assert c.variant == variant
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    xd = c.device_buffer((B, c.dim_x)).upload(x0.astype(npdt))
    ud = c.device_buffer((B, c.dim_u))
    c.closed_loop_device(xd, ud, n)
    c.synchronize()
    x, u = xd.download().astype(np.float64), ud.download().astype(np.float64)
    t, U, d = c.get_state()
    n_ax, reason = c.get_status()
    xd.free(), ud.free(), c.close()
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(u)) and np.all(np.isfinite(U)) and np.all(np.isfinite(d))
    if tol == 0.0:
        assert np.all(n_ax == kmax) and np.all(reason == cg.EXIT_NATURAL)
    s = ol.snap[n]
    u_tol, x_tol, d_rel = (1e-4, 1e-4, 2e-3) if f32 else (1e-9, 1e-9, 1e-7)
    flips = 0
    for j, i in enumerate(sample):
        t_o, U_o, d_o = s["state"][j]
        assert abs(t - t_o) <= (1e-6 if f32 else 1e-12), (t, t_o)
        assert np.max(np.abs(u[i] - s["u"][j])) <= u_tol, (name, n, i, u[i], s["u"][j])
        assert np.max(np.abs(x[i] - s["x"][j])) <= x_tol, (name, n, i, x[i], s["x"][j])
        assert np.max(np.abs(U[i].astype(np.float64) - U_o)) <= u_tol, (name, n, i)
        scale = max(1.0, float(np.max(np.abs(d_o))))
        assert np.max(np.abs(d[i].astype(np.float64) - d_o)) <= d_rel * scale, (name, n, i)
        k_o, _, reason_o = s["solve"][j]
        if f32:
            flips += int(n_ax[i] != k_o)  # fp32: the exit test sits in the rounding noise on many ticks
        else:
            assert n_ax[i] == k_o and reason[i] == reason_o, (name, n, i, n_ax[i], k_o, reason[i], reason_o)
    if f32:
        assert flips <= len(sample) // 2, flips


def test_closed_loop_device_resumes_across_calls(orc):
    """Two calls (7 + 18 ticks) == one call of 25 ticks, bit for bit: the hand-over of (U, dUdt, x, t) through HBM at
    a launch boundary does not depend on where the caller cuts the loop."""
    model, dv, kmax, B = 0, 50, 10, 300
    x0, u0, p = orc.batch_scenario(model, B)

    def run(parts):
        c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=kmax)
        c.set_ptau_repeat(p)
        c.init_u0(u0)
        c.init_u0_newton(u0, x0, p, 10)
        xd = c.device_buffer((B, 4)).upload(x0)
        ud = c.device_buffer((B, 3))
        for m in parts:
            c.closed_loop_device(xd, ud, m)
        c.synchronize()
        out = (xd.download(), ud.download(), c.get_state(), c.get_status())
        xd.free(), ud.free(), c.close()
        return out
    a, b = run([25]), run([7, 18])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2][0] == b[2][0] and np.array_equal(a[2][1], b[2][1]) and np.array_equal(a[2][2], b[2][2])
    assert np.array_equal(a[3][0], b[3][0])


@pytest.mark.parametrize("variant", [0, 3])
def test_multiple_controller_device_loop_vs_oracle(orc, variant):
    """BASELINE configs[3] shape on one GPU: MultipleController.closed_loop_device — a Model1 (MSD) batch and a Model2
    (pendulum) batch, each on its own stream, launches interleaved (multiple_controller/main.cpp:104-118) — for 25
    ticks, every member against the oracle's free-running loops."""
    from cgmres_cpp_amd.multi import MultipleController
    n, dv, km = 25, 50, 10
    # variant 3 = the lean mapping MultipleController picks by itself once its members have to share CUs
    specs = [dict(model="msd", batch=72, dv=dv, k_max=km, variant=variant),
             dict(model="pendulum", batch=88, dv=dv, k_max=km, variant=variant)]
    mc = MultipleController(specs)
    # (library's choice: such small members both take the wave mapping)
    assert [m.variant for m in mc.members] == ([variant, variant] if variant else [4, 4])
    xs, us, want = [], [], []
    for m, model in zip(mc.members, (1, 0)):
        x0, u0, p = orc.batch_scenario(model, m.batch)
        m.set_ptau_repeat(p)
        m.init_u0(u0)
        m.init_u0_newton(u0, x0, p, 10)
        xs.append(m.device_buffer((m.batch, m.dim_x)).upload(x0))
        us.append(m.device_buffer((m.batch, m.dim_u)))
        sample = sample_of(m.batch, 12)
        want.append((sample, OracleLoop(orc, model, dv, km, 1e-6, "f64", x0, u0, p, sample, {n})))
    mc.closed_loop_device(xs, us, n)
    mc.synchronize()
    for m, xd, ud, (sample, ol) in zip(mc.members, xs, us, want):
        x, u = xd.download(), ud.download()
        n_ax, _ = m.get_status()
        s = ol.snap[n]
        for j, i in enumerate(sample):
            assert np.max(np.abs(u[i] - s["u"][j])) <= 1e-9, (m.model, i)
            assert np.max(np.abs(x[i] - s["x"][j])) <= 1e-9, (m.model, i)
            assert n_ax[i] == s["solve"][j][0]
        assert abs(m.t - n * m.dt) < 1e-12
        xd.free(), ud.free()
    mc.close()


@pytest.mark.parametrize("per_instance", [True, False])
@pytest.mark.parametrize("model,variant", [(0, 2), (1, 2), (0, 1), (0, 3), (1, 3), (0, 4), (1, 4)])
def test_closed_loop_device_with_moving_reference(orc, model, variant, per_instance):
    """Time-varying reference inside the fused device loop (cgmres_hip_closed_loop_device_ptau): a new parameter
    horizon before every tick == the reference's `set_ptau` (cgmres.hpp:36-39) called before every `control()`.
    23 ticks = two launch boundaries + a partial tail; the target ramps along the horizon AND from tick to tick."""
    B, dv, km, n = 83, 50, 10, 23
    x0, u0, p = orc.batch_scenario(model, B)
    c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=km, variant=variant)
    c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    npar = c.dim_p * (dv + 1)
    stage = np.arange(dv + 1)
    nb = B if per_instance else 1
    seq = np.empty((n, nb, dv + 1, c.dim_p))
    for k in range(n):
        for i in range(nb):
            # reference component 0 ramps with the tick and along the horizon; component 1 stays
            seq[k, i, :, 0] = p[i, 0] * (1.0 + 0.004 * k) + 0.0007 * stage * (1 + 0.1 * (i % 5))
            seq[k, i, :, 1] = p[i, 1]
    seq = seq.reshape(n, nb, npar) if per_instance else seq.reshape(n, npar)
    sd = c.device_buffer(seq.shape).upload(seq)
    xd = c.device_buffer((B, c.dim_x)).upload(x0)
    ud = c.device_buffer((B, c.dim_u))
    c.closed_loop_device(xd, ud, n, sd, per_instance)
    c.synchronize()
    x, u = xd.download(), ud.download()
    n_ax, _ = c.get_status()
    # the handle keeps the last tick's horizon: one more ordinary tick must use it
    u_next = c.control(x)
    sample = sample_of(B, 14)
    for i in sample:
        r = orc.Controller(model, dv, km)
        orc.start_controller(r, x0[i], u0[i], p[i])
        xi = x0[i].copy()
        for k in range(n):
            r.set_ptau(seq[k, i] if per_instance else seq[k])
            ui = r.control(xi)
            xi = xi + r.plant(xi, ui) * r.dt
        assert np.max(np.abs(u[i] - ui)) <= 1e-9 and np.max(np.abs(x[i] - xi)) <= 1e-9, (i, u[i], ui)
        assert n_ax[i] == r.last_solve()[0]
        assert np.max(np.abs(u_next[i] - r.control(xi))) <= 1e-9, i
    sd.free(), xd.free(), ud.free(), c.close()


def test_binned_placement_is_bit_identical(orc):
    """Early-exit mode on a batch that needs more workgroups than the GPU holds at once: before every fused launch the
    instances are re-placed by the Arnoldi count of their last tick (WgParams::perm, bin_by_count_kernel: a stable
    counting sort).  Instances never exchange data, and in this slow phase of the scenario (the first 35 ticks: the
    rotation form of the trig update never leaves its range, so no wave-wide choice of form is ever taken) every
    instance comes out bit for bit as in caller order (flags = FLAG_NO_BINNING) — three re-placements and a partial
    tail.  (Fast motion: test_binned_placement_in_fast_motion_is_reproducible.)"""
    model, dv, km, B, n = 0, 50, 10, 8300, 35   # 519 sixteen-instance workgroups > 2 x 256 resident (lean plan)
    x0, u0, p = orc.batch_scenario(model, B)
    outs = []
    for flags in (0, cg.FLAG_NO_BINNING):
        c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=km, tol=1e-6, flags=flags)
        assert c.variant == 3
        c.set_ptau_repeat(p)
        c.init_u0(u0)
        c.init_u0_newton(u0, x0, p, 10)
        xd = c.device_buffer((B, 4)).upload(x0)
        ud = c.device_buffer((B, 3))
        c.closed_loop_device(xd, ud, n)
        c.synchronize()
        outs.append((xd.download(), ud.download(), c.get_state(), c.get_status()))
        xd.free(), ud.free(), c.close()
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2][1], b[2][1]) and np.array_equal(a[2][2], b[2][2])
    assert np.array_equal(a[3][0], b[3][0]) and np.array_equal(a[3][1], b[3][1])
    assert np.all(np.isfinite(a[1]))
    # ... and a spread sample against the oracle's free-running loop
    sample = sample_of(B, 10)
    ol = OracleLoop(orc, model, dv, km, 1e-6, "f64", x0, u0, p, sample, {n})
    for j, i in enumerate(sample):
        assert np.max(np.abs(a[1][i] - ol.snap[n]["u"][j])) <= 1e-9 and np.max(np.abs(a[0][i] - ol.snap[n]["x"][j])) <= 1e-9
        assert a[3][0][i] == ol.snap[n]["solve"][j][0]


def test_binned_placement_in_fast_motion_is_reproducible():
    """Tick ~5400 of the seeded scenario: the swing-up reaches 10 rad/s, the rotation form of the quad sweep's trig update
    leaves its range and the kernel falls back to fresh evaluations — a choice the wg mapping takes per WAVE (16
    instances), and the two forms round differently.  There an instance's bits depend on its workgroup mates, so binned
    and caller-order runs agree to rounding only; what must hold: (i) the placement is a stable sort, so two binned runs
    from the same state are bit-identical, (ii) binned vs caller order: the same Arnoldi counts on >= 97 % of the
    instances after two launches and controls within the free-running bound of bench.py's gate (1e-6)."""
    model, dv, km, B, warm, n = 0, 50, 10, 8300, 5400, 20
    x0, u0, p = scenarios.batch(model, B)
    w = cg.CgmresBatch(model, batch=B, dv=dv, k_max=km, tol=1e-6, flags=cg.FLAG_NO_BINNING)
    assert w.variant == 3
    w.set_ptau_repeat(p), w.init_u0(u0), w.init_u0_newton(u0, x0, p, 10)
    xd, ud = w.device_buffer((B, 4)).upload(x0), w.device_buffer((B, 3))
    w.closed_loop_device(xd, ud, warm)
    w.synchronize()
    x_w, (t_w, U_w, d_w) = xd.download(), w.get_state()
    xd.free(), ud.free(), w.close()
    ok = np.all(np.isfinite(x_w), axis=1) & np.all(np.isfinite(U_w), axis=1) & np.all(np.isfinite(d_w), axis=1)
    assert ok.sum() >= B - 8                      # (instance 3599 of the seeded batch diverges in the reference too)
    assert np.max(np.abs(x_w[ok][:, 2:])) > 3.0   # fast motion: angular rates of several rad/s
    x_w[~ok], U_w[~ok], d_w[~ok] = x0[~ok], 0.0, 0.0

    def run(flags):
        c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=km, tol=1e-6, flags=flags)
        c.set_ptau_repeat(p)
        c.set_state(t_w, U_w, d_w)
        xd, ud = c.device_buffer((B, 4)).upload(x_w), c.device_buffer((B, 3))
        c.closed_loop_device(xd, ud, n)
        c.synchronize()
        out = (xd.download(), ud.download(), c.get_status()[0])
        xd.free(), ud.free(), c.close()
        return out
    a, a2, b = run(0), run(0), run(cg.FLAG_NO_BINNING)
    assert np.array_equal(a[0], a2[0], equal_nan=True) and np.array_equal(a[1], a2[1], equal_nan=True)
    assert np.array_equal(a[2], a2[2])
    fin = np.all(np.isfinite(a[1]), axis=1) & np.all(np.isfinite(b[1]), axis=1)
    assert fin.sum() >= B - 16
    assert np.mean(a[2][fin] == b[2][fin]) >= 0.97
    assert np.max(np.abs(a[1][fin] - b[1][fin])) <= 1e-6, np.max(np.abs(a[1][fin] - b[1][fin]))


def test_multiple_controller_at_baseline_size_vs_oracle(orc):
    """BASELINE configs[3] as tools/bench_configs.py --config 4 runs it on one GPU: 4096 Model1 (MSD) + 4096 Model2
    (pendulum) controllers, MultipleController picks the lean mapping by itself (512 workgroups for 256 CUs), the two
    members' kernels are co-resident on the CUs (two workgroups of DIFFERENT kernels per CU).  25 ticks of the joint
    device loop, a spread 24-instance sample of EACH member against the oracle's free-running loops
    (multiple_controller/main.cpp:104-118)."""
    from cgmres_cpp_amd.multi import MultipleController
    n, dv, km, B = 25, 50, 10, 4096
    mc = MultipleController([dict(model="msd", batch=B, dv=dv, k_max=km), dict(model="pendulum", batch=B, dv=dv, k_max=km)])
    assert [m.variant for m in mc.members] == [3, 3]
    xs, us, want = [], [], []
    for m, model in zip(mc.members, (1, 0)):
        x0, u0, p = orc.batch_scenario(model, B)
        m.set_ptau_repeat(p)
        m.init_u0(u0)
        m.init_u0_newton(u0, x0, p, 10)
        xs.append(m.device_buffer((B, m.dim_x)).upload(x0))
        us.append(m.device_buffer((B, m.dim_u)))
        sample = sample_of(B, 20)
        want.append((sample, OracleLoop(orc, model, dv, km, 1e-6, "f64", x0, u0, p, sample, {n})))
    mc.closed_loop_device(xs, us, n)
    mc.synchronize()
    for m, xd, ud, (sample, ol) in zip(mc.members, xs, us, want):
        x, u = xd.download(), ud.download()
        n_ax, reason = m.get_status()
        assert np.all(np.isfinite(x)) and np.all(np.isfinite(u))
        s = ol.snap[n]
        assert len(sample) >= 24
        for j, i in enumerate(sample):
            assert np.max(np.abs(u[i] - s["u"][j])) <= 1e-9, (m.model, i)
            assert np.max(np.abs(x[i] - s["x"][j])) <= 1e-9, (m.model, i)
            assert n_ax[i] == s["solve"][j][0] and reason[i] == s["solve"][j][2]
        xd.free(), ud.free()
    mc.close()
