"""ORACLE — TEST INFRASTRUCTURE ONLY (imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker; never by the product package cgmres_cpp_amd).

ctypes front end for the two checker libraries that share oracle/orc_api.h:
  * ``liboracle.so``     the repo's CPU restatement (oracle/cgmres_oracle.hpp)
  * ``_ref/libref.so``   the unmodified reference compiled in place (only where /root/reference exists)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref.so")
REFERENCE_ROOT = "/root/reference"

PENDULUM, MSD, SEMIACTIVE = 0, 1, 2
MODEL_NAMES = {PENDULUM: "pendulum", MSD: "msd", SEMIACTIVE: "semiactive"}
EXIT_NATURAL, EXIT_CONVERGED, EXIT_SMALL_RESIDUAL, EXIT_BREAKDOWN = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(ref=True, quiet=True):
    """Compile liboracle.so (always) and _ref/libref.so (when the reference tree is mounted)."""
    targets = ["oracle"]
    if ref and os.path.isdir(REFERENCE_ROOT):
        targets.append("ref")
    subprocess.run(["make", "-C", HERE, "-j4"] + targets, check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load(path):
    lib = C.CDLL(path)
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    lib.orc_destroy.argtypes = [C.c_void_p]
    lib.orc_dims.argtypes = [C.c_void_p, _ip]
    lib.orc_tuning.argtypes = [C.c_void_p, _dp]
    lib.orc_set_ptau.argtypes = [C.c_void_p, _dp]
    lib.orc_init_u0.argtypes = [C.c_void_p, _dp]
    lib.orc_init_u0_newton.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int]
    lib.orc_control.argtypes = [C.c_void_p, _dp, _dp]
    lib.orc_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.orc_set_state.argtypes = [C.c_void_p, C.c_double, _dp, _dp]
    lib.orc_F.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double]
    lib.orc_prepare.argtypes = [C.c_void_p, _dp, _dp]
    lib.orc_Ax.argtypes = [C.c_void_p, _dp, _dp]
    lib.orc_gmres.argtypes = [C.c_void_p, _dp, _dp]
    lib.orc_get_krylov.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    lib.orc_last_solve.argtypes = [C.c_void_p, _ip]
    lib.orc_plant.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.orc_run_closed_loop.restype = C.c_double
    lib.orc_run_closed_loop.argtypes = [C.POINTER(C.c_void_p), C.c_int, _dp, _dp, C.c_int, C.c_int]
    return lib


_libs = {}


def lib(which="oracle"):
    if which not in _libs:
        path = ORACLE_SO if which == "oracle" else REF_SO
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built (python -c 'import oracle.orc as o; o.build()')")
        _libs[which] = _load(path)
    return _libs[which]


def have_ref():
    return os.path.exists(REF_SO)


def _p(a):
    return a.ctypes.data_as(_dp)


def _arr(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


class Controller:
    """One controller instance of either checker library; method names follow the reference
    (include/cgmres.hpp: set_ptau / init_u0 / init_u0_newton / control)."""

    def __init__(self, model, dv, kmax, tol=-1.0, dtype="f64", which="oracle"):
        self._lib = lib(which)
        self._h = self._lib.orc_create(model, dv, kmax, float(tol), 1 if dtype == "f32" else 0)
        if not self._h:
            raise ValueError(f"{which}: combination not built: model={model} dv={dv} kmax={kmax} tol={tol}")
        d = (C.c_int * 7)()
        self._lib.orc_dims(self._h, d)
        self.dim_x, self.dim_u, self.dim_p, self.dv, self.kmax, self.len, _ = list(d)
        t = (C.c_double * 5)()
        self._lib.orc_tuning(self._h, t)
        self.dt, self.h, self.zeta, self.Tf, self.alpha = list(t)
        self.model = model

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.orc_destroy(self._h)
            self._h = None

    def set_ptau(self, ptau):
        a = _arr(ptau, self.dim_p * (self.dv + 1))
        if a.size:
            self._lib.orc_set_ptau(self._h, _p(a))

    def set_ptau_repeat(self, p):
        self.set_ptau(np.tile(_arr(p, self.dim_p), self.dv + 1))

    def init_u0(self, u0):
        self._lib.orc_init_u0(self._h, _p(_arr(u0, self.dim_u)))

    def init_u0_newton(self, u0, x0, p0, n_loop=10):
        u = _arr(u0, self.dim_u).copy()
        p = _arr(p0) if self.dim_p else np.zeros(1)
        self._lib.orc_init_u0_newton(self._h, _p(u), _p(_arr(x0, self.dim_x)), _p(p), n_loop)
        return u

    def control(self, x):
        u = np.empty(self.dim_u)
        self._lib.orc_control(self._h, _p(u), _p(_arr(x, self.dim_x)))
        return u

    def get_state(self):
        t = C.c_double()
        U = np.empty(self.len)
        d = np.empty(self.len)
        self._lib.orc_get_state(self._h, C.byref(t), _p(U), _p(d))
        return t.value, U, d

    def set_state(self, t, U, dUdt):
        self._lib.orc_set_state(self._h, float(t), _p(_arr(U, self.len)), _p(_arr(dUdt, self.len)))

    def F(self, U, x, t):
        r = np.empty(self.len)
        self._lib.orc_F(self._h, _p(r), _p(_arr(U, self.len)), _p(_arr(x, self.dim_x)), float(t))
        return r

    def prepare(self, x):
        b = np.empty(self.len)
        self._lib.orc_prepare(self._h, _p(b), _p(_arr(x, self.dim_x)))
        return b

    def Ax(self, v):
        o = np.empty(self.len)
        self._lib.orc_Ax(self._h, _p(o), _p(_arr(v, self.len)))
        return o

    def gmres(self, x0, b):
        x = _arr(x0, self.len).copy()
        self._lib.orc_gmres(self._h, _p(x), _p(_arr(b, self.len)))
        return x

    def krylov(self):
        k1 = self.kmax + 1
        V = np.empty(self.len * k1)
        H = np.empty(k1 * k1)
        rho = np.empty(k1)
        g = np.empty(3 * self.kmax)
        self._lib.orc_get_krylov(self._h, _p(V), _p(H), _p(rho), _p(g))
        return V.reshape(k1, self.len), H.reshape(k1, k1), rho, g.reshape(self.kmax, 3)

    def last_solve(self):
        o = (C.c_int * 3)()
        self._lib.orc_last_solve(self._h, o)
        return tuple(o)

    def plant(self, x, u):
        f = np.empty(self.dim_x)
        self._lib.orc_plant(self._h, _p(f), _p(_arr(x, self.dim_x)), _p(_arr(u, self.dim_u)))
        return f


# ---------------------------------------------------------------------------------------------
# Example scenarios: initial state / reference / initial guess of each example main
# (arm_type_inverted_pendulum/main.cpp:35-52, mass_spring_damper/main.cpp:35-55,
#  semiactive_damper/main.cpp:35-40), plus the seeded per-instance perturbation of SURVEY.md §8(d).
# ---------------------------------------------------------------------------------------------
PI_TILDE = 3.14159265358979


def splitmix64(seed):
    """Generator of u01 = (z >> 11) * 2**-53."""
    mask = (1 << 64) - 1
    state = seed & mask
    while True:
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        yield (z >> 11) * (2.0 ** -53)


def shipped_scenario(model):
    """(x0, u0_guess, p) of the example main, unperturbed."""
    if model == PENDULUM:
        return (np.array([PI_TILDE, PI_TILDE, 0.0, 0.0]), np.array([0.0, 3.0, 0.01]),
                np.array([PI_TILDE / 4.0, 0.0]))
    if model == MSD:
        return (np.array([2.0, 2.0, 0.0, 0.0]), np.array([0.0, 0.0, 10.0, 10.0, 5e-4, 5e-4]),
                np.array([1.0, -1.0]))
    if model == SEMIACTIVE:
        return (np.array([2.0, 0.0]), np.array([0.028393761456740, 0.166095020295846, 0.030103250483332]),
                np.zeros(0))
    raise ValueError(model)


def batch_scenario(model, batch, seed=12345):
    """Per-instance perturbed (x0[B,dim_x], u0[B,dim_u], p[B,dim_p]) exactly as SURVEY.md §8(d) lists:
    values drawn per instance, in order r1, r2, ..."""
    rng = splitmix64(seed)
    x0s, u0s, ps = [], [], []
    x0, u0, p = shipped_scenario(model)
    for _ in range(batch):
        if model == PENDULUM:
            r = [next(rng) for _ in range(5)]
            x = np.array([PI_TILDE + 0.2 * (r[0] - 0.5), PI_TILDE + 0.2 * (r[1] - 0.5),
                          0.2 * (r[2] - 0.5), 0.2 * (r[3] - 0.5)])
            pp = np.array([(PI_TILDE / 4.0) * (0.5 + r[4]), 0.0])
        elif model == MSD:
            r = [next(rng) for _ in range(2)]
            x = np.array([2.0 + 0.4 * (r[0] - 0.5), 2.0 + 0.4 * (r[1] - 0.5), 0.0, 0.0])
            pp = p.copy()
        else:
            r = [next(rng)]
            x = np.array([2.0 + 0.4 * (r[0] - 0.5), 0.0])
            pp = p.copy()
        x0s.append(x)
        u0s.append(u0.copy())
        ps.append(pp)
    return np.array(x0s), np.array(u0s), np.array(ps).reshape(batch, -1)


def start_controller(ctrl, x0, u0, p, n_newton=10):
    """The setup block of each example main (*/main.cpp:54-57): set_ptau, init_u0, init_u0_newton."""
    if ctrl.dim_p:
        ctrl.set_ptau_repeat(p)
    ctrl.init_u0(u0)
    return ctrl.init_u0_newton(u0, x0, p, n_newton)


def closed_loop(ctrl, x0, n_ticks, record_state=False):
    """The example main loop (*/main.cpp:63-73): control, then forward-Euler plant with the
    reference's mul-then-add rounding (dxdt*dt, then x + that)."""
    x = np.array(x0, dtype=np.float64)
    us, xs, states, ks = [], [], [], []
    for _ in range(n_ticks):
        if record_state:
            states.append((ctrl.get_state(), x.copy()))
        u = ctrl.control(x)
        ks.append(ctrl.last_solve()[0])
        f = ctrl.plant(x, u)
        x = x + f * ctrl.dt
        us.append(u)
        xs.append(x.copy())
    return np.array(us), np.array(xs), np.array(ks), states


def run_closed_loop(ctrls, x, ticks, nthreads):
    """CPU-baseline driver: `ticks` closed-loop ticks of every controller in `ctrls` (same library),
    instances statically partitioned over `nthreads` native threads.  x [n, dim_x] is advanced in place.
    Returns (wall seconds, u of the last tick)."""
    n = len(ctrls)
    hs = (C.c_void_p * n)(*[c._h for c in ctrls])
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.zeros((n, ctrls[0].dim_u))
    secs = ctrls[0]._lib.orc_run_closed_loop(hs, n, _p(x), _p(u), int(ticks), int(nthreads))
    return secs, u, x
