"""cgmres_cpp_amd — batched C/GMRES control ticks on MI355X (gfx950).

Python mirror of the reference's controller interface (include/cgmres.hpp of blockahead/CGMRES_cpp:
``set_ptau`` / ``set_ptau_repeat`` / ``init_u0`` / ``init_u0_newton`` / ``control``) over the C ABI of
``libcgmres_hip.so`` (include/cgmres_hip.h), batched over independent controller instances.

There is no CPU implementation behind this package: loading fails loudly when the HIP library is
missing, and creating a controller fails when no gfx950 device is usable.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

PENDULUM, MSD, SEMIACTIVE = 0, 1, 2
MODEL_IDS = {"pendulum": PENDULUM, "arm_type_inverted_pendulum": PENDULUM, "msd": MSD,
             "mass_spring_damper": MSD, "semiactive": SEMIACTIVE, "semiactive_damper": SEMIACTIVE}
F64, F32 = 0, 1
EXIT_NATURAL, EXIT_CONVERGED, EXIT_SMALL_RESIDUAL, EXIT_BREAKDOWN, EXIT_NONFINITE = 0, 1, 2, 3, 4
FLAG_SERIAL_COSTATE, FLAG_IPW8, FLAG_NO_BINNING, FLAG_TWO_PASS_COSTATE = 1, 2, 4, 8
FLAG_NO_WAVE, FLAG_WAVE_FRESH_TRIG, FLAG_WAVE_SERIAL_SWEEPS, FLAG_SERIAL_STATE_SWEEP = 16, 32, 64, 128
ABI_VERSION = 2
TICKS_PER_LAUNCH = 10  # CGMRES_HIP_TICKS_PER_LAUNCH: closed_loop_device fuses this many ticks per launch (wg mapping)

# every symbol include/cgmres_hip.h declares (tests/test_capi_symbols.py checks header == this == library)
SYMBOLS = [
    "cgmres_hip_model_info", "cgmres_hip_default_config", "cgmres_hip_model_probe", "cgmres_hip_register_model",
    "cgmres_hip_selftest_sincos", "cgmres_hip_register_operator", "cgmres_hip_operator_info", "cgmres_hip_gmres_user",
    "cgmres_hip_last_error",
    "cgmres_hip_device_count", "cgmres_hip_create", "cgmres_hip_destroy", "cgmres_hip_get_config", "cgmres_hip_variant_name",
    "cgmres_hip_set_ptau", "cgmres_hip_set_ptau_repeat", "cgmres_hip_init_u0", "cgmres_hip_init_u0_newton",
    "cgmres_hip_control", "cgmres_hip_control_device", "cgmres_hip_closed_loop_device",
    "cgmres_hip_closed_loop_device_ptau", "cgmres_hip_synchronize", "cgmres_hip_shard_bounds",
    "cgmres_hip_get_time", "cgmres_hip_get_state", "cgmres_hip_set_state", "cgmres_hip_get_status",
    "cgmres_hip_get_krylov", "cgmres_hip_F_func", "cgmres_hip_prepare", "cgmres_hip_Ax_func", "cgmres_hip_gmres",
    "cgmres_hip_timer_start", "cgmres_hip_timer_stop", "cgmres_hip_malloc", "cgmres_hip_free",
    "cgmres_hip_memcpy_h2d", "cgmres_hip_memcpy_d2h",
]


class Config(C.Structure):
    """struct cgmres_hip_config (include/cgmres_hip.h)."""
    _fields_ = [("abi_version", C.c_int32), ("model_id", C.c_int32), ("dtype", C.c_int32), ("batch", C.c_int32),
                ("dv", C.c_int32), ("k_max", C.c_int32), ("device", C.c_int32), ("variant", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32), ("tol", C.c_double), ("dt", C.c_double), ("h", C.c_double), ("zeta", C.c_double),
                ("Tf", C.c_double), ("alpha", C.c_double), ("stream", C.c_void_p)]


class CgmresHipError(RuntimeError):
    pass


_lib = None


def lib_path():
    """The in-tree library; CGMRES_HIP_LIB overrides it (A/B builds of the kernels, tools/ab_build.py)."""
    return os.environ.get("CGMRES_HIP_LIB") or _build.LIB_PATH


def load():
    """Loads libcgmres_hip.so (never builds implicitly; see cgmres_cpp_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise CgmresHipError(f"{path} is missing: build it with `python -m cgmres_cpp_amd.build` "
                             "(there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int32
    lib.cgmres_hip_last_error.restype = C.c_char_p
    lib.cgmres_hip_model_info.argtypes = [i32, C.POINTER(i32), C.POINTER(C.c_double)]
    lib.cgmres_hip_default_config.argtypes = [i32, C.POINTER(Config)]
    lib.cgmres_hip_model_probe.argtypes = [i32, i32] + [C.POINTER(C.c_double)] * 5
    lib.cgmres_hip_selftest_sincos.argtypes = [i32, C.POINTER(C.c_double), i32, C.POINTER(C.c_double),
                                               C.POINTER(C.c_double)]
    lib.cgmres_hip_register_operator.argtypes = [C.c_char_p, C.POINTER(i32)]
    lib.cgmres_hip_operator_info.argtypes = [i32, C.POINTER(i32)]
    lib.cgmres_hip_gmres_user.argtypes = [i32, i32, i32, i32, C.c_double, vp, vp, vp, vp, vp]
    lib.cgmres_hip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.cgmres_hip_destroy.argtypes = [vp]
    lib.cgmres_hip_get_config.argtypes = [vp, C.POINTER(Config)]
    lib.cgmres_hip_variant_name.argtypes = [vp]
    lib.cgmres_hip_variant_name.restype = C.c_char_p
    lib.cgmres_hip_set_ptau.argtypes = [vp, vp, C.c_int]
    lib.cgmres_hip_set_ptau_repeat.argtypes = [vp, vp, C.c_int]
    lib.cgmres_hip_init_u0.argtypes = [vp, vp, C.c_int]
    lib.cgmres_hip_init_u0_newton.argtypes = [vp, vp, vp, vp, i32]
    lib.cgmres_hip_control.argtypes = [vp, vp, vp]
    lib.cgmres_hip_control_device.argtypes = [vp, vp, vp]
    lib.cgmres_hip_closed_loop_device.argtypes = [vp, vp, vp, i32]
    lib.cgmres_hip_closed_loop_device_ptau.argtypes = [vp, vp, vp, i32, vp, C.c_int]
    lib.cgmres_hip_synchronize.argtypes = [vp]
    lib.cgmres_hip_shard_bounds.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.cgmres_hip_get_time.argtypes = [vp, C.POINTER(C.c_double)]
    lib.cgmres_hip_get_state.argtypes = [vp, C.POINTER(C.c_double), vp, vp]
    lib.cgmres_hip_set_state.argtypes = [vp, C.c_double, vp, vp]
    lib.cgmres_hip_get_status.argtypes = [vp, vp, vp]
    lib.cgmres_hip_get_krylov.argtypes = [vp, vp, vp, vp, vp]
    lib.cgmres_hip_F_func.argtypes = [vp, vp, vp, vp, C.c_double]
    lib.cgmres_hip_prepare.argtypes = [vp, vp, vp]
    lib.cgmres_hip_Ax_func.argtypes = [vp, vp, vp]
    lib.cgmres_hip_gmres.argtypes = [vp, vp, vp]
    lib.cgmres_hip_timer_start.argtypes = [vp]
    lib.cgmres_hip_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.cgmres_hip_malloc.argtypes = [vp, C.POINTER(vp), C.c_uint64]
    lib.cgmres_hip_free.argtypes = [vp, vp]
    lib.cgmres_hip_memcpy_h2d.argtypes = [vp, vp, vp, C.c_uint64]
    lib.cgmres_hip_memcpy_d2h.argtypes = [vp, vp, vp, C.c_uint64]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise CgmresHipError(f"cgmres_hip error {rc}: {load().cgmres_hip_last_error().decode()}")


def device_count():
    return load().cgmres_hip_device_count()


def model_info(model):
    """dict(dim_x, dim_u, dim_p, dv, k_max, dt, h, zeta, Tf, alpha, tol): the Model constants of the registry."""
    model = MODEL_IDS.get(model, model)
    d = (C.c_int32 * 5)()
    t = (C.c_double * 6)()
    _check(load().cgmres_hip_model_info(model, d, t))
    return dict(zip(("dim_x", "dim_u", "dim_p", "dv", "k_max"), list(d)),
                **dict(zip(("dt", "h", "zeta", "Tf", "alpha", "tol"), list(t))))


def model_probe(model, x, u, p, lmd, device=0):
    """Device evaluation of [dxdt | dPhidx | dHdx | dHdu] at one point (registry fingerprint)."""
    model = MODEL_IDS.get(model, model)
    mi = model_info(model)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, u, p if mi["dim_p"] else np.zeros(1), lmd)]
    out = np.empty(3 * mi["dim_x"] + mi["dim_u"])
    dp = C.POINTER(C.c_double)
    _check(load().cgmres_hip_model_probe(model, device, *[a.ctypes.data_as(dp) for a in arrs],
                                         out.ctypes.data_as(dp)))
    nx = mi["dim_x"]
    return out[:nx], out[nx:2 * nx], out[2 * nx:3 * nx], out[3 * nx:]


def selftest_sincos(a, device=0):
    """(sin, cos) of the fp64 arguments `a` as computed by the device routine of the horizon sweeps."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    s, c = np.empty_like(a), np.empty_like(a)
    dp = C.POINTER(C.c_double)
    _check(load().cgmres_hip_selftest_sincos(device, a.ctypes.data_as(dp), a.size, s.ctypes.data_as(dp),
                                             c.ctypes.data_as(dp)))
    return s, c


def gmres_user(op_id, x0, b, k_max, tol, params=None, device=0):
    """Gmres::gmres(x, b) (reference include/gmres.hpp:28-112) with a registered device operator (plugin.register_operator)
    for a batch of independent systems: x0, b [batch, len]; params [batch, n_params].  Returns (x, n_ax, reason)."""
    d = (C.c_int32 * 2)()
    _check(load().cgmres_hip_operator_info(op_id, d))
    L, npar = d[0], d[1]
    x = np.array(np.asarray(x0, dtype=np.float64).reshape(-1, L))
    bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(-1, L))
    B = len(x)
    if len(bb) != B:
        raise ValueError("x0 and b must have the same batch")
    pp = None
    if npar:
        pp = np.ascontiguousarray(np.broadcast_to(np.asarray(params, dtype=np.float64).reshape(-1, npar), (B, npar)))
    n_ax = np.empty(B, dtype=np.int32)
    why = np.empty(B, dtype=np.int32)
    _check(load().cgmres_hip_gmres_user(op_id, device, B, int(k_max), float(tol), pp.ctypes.data if npar else None,
                                        x.ctypes.data, bb.ctypes.data, n_ax.ctypes.data, why.ctypes.data))
    return x, n_ax, why


class DeviceBuffer:
    """A raw HBM allocation owned through the C ABI (used where no torch tensor is at hand)."""

    def __init__(self, ctrl, shape, dtype):
        self._ctrl = ctrl
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _check(load().cgmres_hip_malloc(ctrl._h, C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        _check(load().cgmres_hip_memcpy_h2d(self._ctrl._h, self.ptr, a.ctypes.data, self.nbytes))
        return self

    def download(self):
        a = np.empty(self.shape, dtype=self.dtype)
        _check(load().cgmres_hip_memcpy_d2h(self._ctrl._h, a.ctypes.data, self.ptr, self.nbytes))
        return a

    def free(self):
        if self.ptr and self._ctrl._h:
            _check(load().cgmres_hip_free(self._ctrl._h, self.ptr))
        self.ptr = None


def parse_dtype(dtype):
    """F64 / F32 from 'f64' | 'float64' | np.float64 | torch.float64 | F64 (and the fp32 spellings); anything else is a
    TypeError — a silent fp64 default would make the kernels read 8-byte elements from a caller's 4-byte buffers."""
    if isinstance(dtype, (int, np.integer)) and not isinstance(dtype, bool) and int(dtype) in (F64, F32):
        return int(dtype)
    name = dtype if isinstance(dtype, str) else None
    if name is None and dtype is not None:
        s = str(dtype)
        if s.startswith("torch."):
            name = s[6:]
        else:
            try:
                name = np.dtype(dtype).name
            except TypeError:
                name = None
    table = {"f64": F64, "float64": F64, "double": F64, "f32": F32, "float32": F32, "float": F32, "single": F32}
    if name not in table:
        raise TypeError(f"dtype must be fp64 or fp32, got {dtype!r}")
    return table[name]


def _ptr(a, dtype=None, numel=None):
    """Device/host address of a numpy array, a DeviceBuffer, a torch tensor or a raw int.  With dtype/numel the
    element type and element count of typed containers are checked (raw ints cannot be)."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    have_dt, have_n = None, None
    if isinstance(a, DeviceBuffer):
        ptr, have_dt, have_n = a.ptr, a.dtype, int(np.prod(a.shape))
    elif isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        ptr, have_dt, have_n = a.ctypes.data, a.dtype, a.size
    elif hasattr(a, "data_ptr"):
        if hasattr(a, "is_contiguous") and not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        ptr, have_n = a.data_ptr(), int(a.numel())
        have_dt = np.dtype(str(a.dtype)[6:]) if str(a.dtype).startswith("torch.") else None
    else:
        raise TypeError(type(a))
    if dtype is not None and have_dt is not None and np.dtype(have_dt) != np.dtype(dtype):
        raise TypeError(f"buffer holds {np.dtype(have_dt).name}, the controller computes in {np.dtype(dtype).name}")
    if numel is not None and have_n is not None and have_n != numel:
        raise ValueError(f"buffer holds {have_n} elements, expected {numel}")
    return ptr


class CgmresBatch:
    """`batch` reference controllers (``Cgmres<Model>``, include/cgmres.hpp:9) advancing in lock-step on one GPU.

    Method names, argument meaning and defaults follow the reference class; every vector gains a leading
    batch axis.  ``dv`` / ``k_max`` / ``tol`` and the tuning constants default to the Model's shipped
    ``static constexpr`` values (e.g. arm_type_inverted_pendulum/model.hpp:21-35).
    """

    def __init__(self, model, batch=1, dv=None, k_max=None, tol=None, dtype="f64", device=0, stream=None,
                 variant=0, flags=0, **tuning):
        lib = load()
        self.model = MODEL_IDS.get(model, model)
        cfg = Config()
        _check(lib.cgmres_hip_default_config(self.model, C.byref(cfg)))
        cfg.batch = int(batch)
        cfg.dtype = parse_dtype(dtype)
        if dv is not None:
            cfg.dv = int(dv)
        if k_max is not None:
            cfg.k_max = int(k_max)
        if tol is not None and tol >= 0:
            cfg.tol = float(tol)
        for k, v in tuning.items():
            if k not in ("dt", "h", "zeta", "Tf", "alpha"):
                raise TypeError(f"unknown tuning constant {k}")
            setattr(cfg, k, float(v))
        cfg.device = int(device)
        cfg.variant = int(variant)
        cfg.flags = int(flags)
        cfg.stream = stream
        self._h = None
        h = C.c_void_p()
        _check(lib.cgmres_hip_create(C.byref(cfg), C.byref(h)))
        self._h = h.value
        _check(lib.cgmres_hip_get_config(self._h, C.byref(cfg)))  # resolved variant
        self.cfg = cfg
        self.variant = cfg.variant
        self.variant_name = lib.cgmres_hip_variant_name(self._h).decode()  # "lane" | "wg" | "wg-lean" | "wg+parallel-costate"
        mi = model_info(self.model)
        self.dim_x, self.dim_u, self.dim_p = mi["dim_x"], mi["dim_u"], mi["dim_p"]
        self.batch, self.dv, self.k_max = cfg.batch, cfg.dv, cfg.k_max
        self.len = self.dim_u * self.dv
        self.dt, self.h, self.zeta, self.Tf, self.alpha, self.tol = cfg.dt, cfg.h, cfg.zeta, cfg.Tf, cfg.alpha, cfg.tol
        self.np_dtype = np.float32 if cfg.dtype == F32 else np.float64

    def close(self):
        if self._h:
            load().cgmres_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ----------------------------------------------------------------------------------
    def _host(self, a, shape, allow_broadcast=False):
        a = np.ascontiguousarray(a, dtype=self.np_dtype)
        if allow_broadcast and a.size == int(np.prod(shape[1:])):
            return a.reshape(shape[1:]), 0
        if a.size != int(np.prod(shape)):
            raise ValueError(f"expected {shape}, got {a.shape}")
        return a.reshape(shape), 1

    def device_buffer(self, shape, dtype=None):
        return DeviceBuffer(self, shape, dtype or self.np_dtype)

    # -- setup, cgmres.hpp:36-76 ------------------------------------------------------------------
    def set_ptau(self, ptau):
        if self.dim_p == 0:
            return
        a, per = self._host(ptau, (self.batch, self.dim_p * (self.dv + 1)), True)
        _check(load().cgmres_hip_set_ptau(self._h, a.ctypes.data, per))

    def set_ptau_repeat(self, p):
        if self.dim_p == 0:
            return
        a, per = self._host(p, (self.batch, self.dim_p), True)
        _check(load().cgmres_hip_set_ptau_repeat(self._h, a.ctypes.data, per))

    def init_u0(self, u0):
        a, per = self._host(u0, (self.batch, self.dim_u), True)
        _check(load().cgmres_hip_init_u0(self._h, a.ctypes.data, per))

    def init_u0_newton(self, u0, x0, p0=None, n_loop=10):
        """Returns the refined u0 [batch, dim_u] (the reference updates the caller's u0 in place)."""
        u = np.array(np.broadcast_to(np.asarray(u0, dtype=self.np_dtype).reshape(-1, self.dim_u),
                                     (self.batch, self.dim_u)))
        x = np.array(np.broadcast_to(np.asarray(x0, dtype=self.np_dtype).reshape(-1, self.dim_x),
                                     (self.batch, self.dim_x)))
        if self.dim_p:
            p = np.array(np.broadcast_to(np.asarray(p0, dtype=self.np_dtype).reshape(-1, self.dim_p),
                                         (self.batch, self.dim_p)))
            pp = p.ctypes.data
        else:
            pp = None
        _check(load().cgmres_hip_init_u0_newton(self._h, u.ctypes.data, x.ctypes.data, pp, int(n_loop)))
        return u

    # -- the hot path, cgmres.hpp:78-110 ------------------------------------------------------------
    def control(self, x):
        """One tick. x [batch, dim_x] host array -> u [batch, dim_u]."""
        a, _ = self._host(x, (self.batch, self.dim_x))
        u = np.empty((self.batch, self.dim_u), dtype=self.np_dtype)
        _check(load().cgmres_hip_control(self._h, u.ctypes.data, a.ctypes.data))
        return u

    def control_device(self, u_dev, x_dev):
        _check(load().cgmres_hip_control_device(self._h, _ptr(u_dev, self.np_dtype, self.batch * self.dim_u),
                                                _ptr(x_dev, self.np_dtype, self.batch * self.dim_x)))

    def closed_loop_device(self, x_dev, u_dev, n_ticks, ptau_seq_dev=None, per_instance=True):
        """n_ticks of the example main loop on the device.  ptau_seq_dev: optional device array of per-tick parameter
        horizons, [n_ticks, batch, dim_p*(dv+1)] (per_instance) or [n_ticks, dim_p*(dv+1)] — `set_ptau` before every tick."""
        xp = _ptr(x_dev, self.np_dtype, self.batch * self.dim_x)
        up = _ptr(u_dev, self.np_dtype, self.batch * self.dim_u)
        if ptau_seq_dev is None or self.dim_p == 0:
            _check(load().cgmres_hip_closed_loop_device(self._h, xp, up, int(n_ticks)))
        else:
            n = int(n_ticks) * (self.batch if per_instance else 1) * self.dim_p * (self.dv + 1)
            _check(load().cgmres_hip_closed_loop_device_ptau(self._h, xp, up, int(n_ticks),
                                                             _ptr(ptau_seq_dev, self.np_dtype, n), 1 if per_instance else 0))

    def synchronize(self):
        _check(load().cgmres_hip_synchronize(self._h))

    # -- state ------------------------------------------------------------------------------------
    @property
    def t(self):
        t = C.c_double()
        _check(load().cgmres_hip_get_time(self._h, C.byref(t)))
        return t.value

    def get_state(self):
        t = C.c_double()
        U = np.empty((self.batch, self.len), dtype=self.np_dtype)
        d = np.empty((self.batch, self.len), dtype=self.np_dtype)
        _check(load().cgmres_hip_get_state(self._h, C.byref(t), U.ctypes.data, d.ctypes.data))
        return t.value, U, d

    def set_state(self, t, U=None, dUdt=None):
        Ua = self._host(U, (self.batch, self.len))[0] if U is not None else None
        da = self._host(dUdt, (self.batch, self.len))[0] if dUdt is not None else None
        _check(load().cgmres_hip_set_state(self._h, float(t), _ptr(Ua), _ptr(da)))

    def get_status(self):
        n = np.empty(self.batch, dtype=np.int32)
        r = np.empty(self.batch, dtype=np.int32)
        _check(load().cgmres_hip_get_status(self._h, n.ctypes.data, r.ctypes.data))
        return n, r

    def get_krylov(self, with_V=False):
        k1 = self.k_max + 1
        H = np.empty((self.batch, k1, k1), dtype=self.np_dtype)
        rho = np.empty((self.batch, k1), dtype=self.np_dtype)
        g = np.empty((self.batch, self.k_max, 3), dtype=self.np_dtype)
        V = np.empty((self.batch, k1, self.len), dtype=self.np_dtype) if with_V else None
        _check(load().cgmres_hip_get_krylov(self._h, _ptr(V), H.ctypes.data, rho.ctypes.data, g.ctypes.data))
        return V, H, rho, g

    # -- white-box hooks ----------------------------------------------------------------------------
    def F_func(self, U, x, t):
        Ua = self._host(U, (self.batch, self.len))[0]
        xa = self._host(x, (self.batch, self.dim_x))[0]
        r = np.empty((self.batch, self.len), dtype=self.np_dtype)
        _check(load().cgmres_hip_F_func(self._h, r.ctypes.data, Ua.ctypes.data, xa.ctypes.data, float(t)))
        return r

    def prepare(self, x):
        xa = self._host(x, (self.batch, self.dim_x))[0]
        b = np.empty((self.batch, self.len), dtype=self.np_dtype)
        _check(load().cgmres_hip_prepare(self._h, b.ctypes.data, xa.ctypes.data))
        return b

    def Ax_func(self, v):
        va = self._host(v, (self.batch, self.len))[0]
        o = np.empty((self.batch, self.len), dtype=self.np_dtype)
        _check(load().cgmres_hip_Ax_func(self._h, o.ctypes.data, va.ctypes.data))
        return o

    def gmres(self, x0, b):
        xa = self._host(x0, (self.batch, self.len))[0].copy()
        ba = self._host(b, (self.batch, self.len))[0]
        _check(load().cgmres_hip_gmres(self._h, xa.ctypes.data, ba.ctypes.data))
        return xa

    # -- measurement --------------------------------------------------------------------------------
    def timer_start(self):
        _check(load().cgmres_hip_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        _check(load().cgmres_hip_timer_stop(self._h, C.byref(ms)))
        return ms.value
