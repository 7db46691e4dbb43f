"""TEST INFRASTRUCTURE — the oracle gate that every published number passes before it is printed.

Used by bench.py and tools/bench_configs.py (and by the GPU tests) as the CHECKER of a run that has already been
timed: a spread sample of the instances a process has just advanced on the GPU is replayed on the CPU oracle
(oracle/liboracle.so — the CPU restatement of the reference, pinned bit-exact to the compiled reference by
tests/test_oracle_vs_ref.py) and compared.  Nothing here is ever inside a timed region, and nothing in the product
package imports this module.

Tolerances: fp64 one teacher-forced tick 1e-9 on u and x; fp32 1e-4 against the fp32 oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL_IDS = {"pendulum": 0, "msd": 1, "semiactive": 2, 0: 0, 1: 1, 2: 2}


class ParityError(RuntimeError):
    pass


def sample_of(B, n):
    """n instances spread over [0, B) plus the last one and both sides of a workgroup edge (15 | 16)."""
    step = max(1, B // n)
    s = list(range(0, B, step))[:n]
    for extra in (B - 1, 15, 16):
        if 0 <= extra < B and extra not in s:
            s.append(extra)
    return sorted(s)


class OracleSample:
    """Oracle controllers for a spread sample of one batch (one model, one size, one scalar type)."""

    def __init__(self, model, dv, kmax, tol, x0, u0, p, n, dtype="f64"):
        from oracle import orc
        if not os.path.exists(orc.ORACLE_SO):
            orc.build(ref=False)
        self.orc = orc
        self.dtype = dtype
        self.idx = sample_of(len(x0), n)
        self.x = [np.array(x0[i], dtype=np.float64) for i in self.idx]
        self.ctrls = []
        for i in self.idx:
            c = orc.Controller(MODEL_IDS[model], dv, kmax, tol, dtype=dtype)
            orc.start_controller(c, x0[i], u0[i], p[i] if c.dim_p else np.zeros(0))
            self.ctrls.append(c)
        self.u = [None] * len(self.idx)
        self.flips = 0

    def _euler(self, c, x, u):
        # the example main's plant step (<example>/main.cpp:71-73); the fp32 job rounds like the device does
        f = c.plant(x, u)
        if self.dtype == "f32":  # the device does x + f*dt in the controller's precision
            return (x.astype(np.float32) + f.astype(np.float32) * np.float32(c.dt)).astype(np.float64)
        return x + f * c.dt

    def advance(self, ticks):
        for j, c in enumerate(self.ctrls):
            for _ in range(ticks):
                u = c.control(self.x[j])
                self.x[j] = self._euler(c, self.x[j], u)
                self.u[j] = u

    def adopt(self, t, U, dUdt, x):
        """Teacher forcing: take over the device's controller and plant state."""
        for j, i in enumerate(self.idx):
            self.ctrls[j].set_state(t, U[i], dUdt[i])
            self.x[j] = np.array(x[i], dtype=np.float64)

    def compare(self, what, x, u, n_ax, tol_u, strict_counts):
        """max error of u and x over the sample; Arnoldi counts must be equal when `strict_counts` (in early-exit
        mode a differing count is counted in .flips instead: the exit test can sit within rounding of tol, §8c)"""
        worst = 0.0
        self.flips = 0
        for j, i in enumerate(self.idx):
            if not (np.all(np.isfinite(u[i])) and np.all(np.isfinite(x[i]))) and not np.all(np.isfinite(self.u[j])):
                continue  # diverged on BOTH sides (the caller decides what a non-finite instance means)
            du = float(np.max(np.abs(np.asarray(u[i], dtype=np.float64) - self.u[j])))
            dx = float(np.max(np.abs(np.asarray(x[i], dtype=np.float64) - self.x[j])))
            worst = max(worst, du, dx)
            if not (du <= tol_u and dx <= tol_u):
                raise ParityError(f"{what}: instance {i}: |du| = {du:.3e}, |dx| = {dx:.3e} > {tol_u:g}")
            k_o = self.ctrls[j].last_solve()[0]
            if n_ax[i] != k_o:
                if strict_counts:
                    raise ParityError(f"{what}: instance {i}: Arnoldi count {n_ax[i]} vs oracle {k_o}")
                self.flips += 1
        return worst


def gate_continuation(ctrl, chk, x_dev, u_dev, sync, tol_1tick, tol_free, strict_counts, free_ticks=10):
    """From the controller/plant state a timed region left behind, continued on both sides (teacher-forced from the
    device's own state): ONE tick at the single-tick tolerance, then `free_ticks` free-running fused ticks across a
    launch boundary.  x_dev/u_dev: torch tensors the handle's closed loop advances in place; sync(): joins the stream.
    Returns a dict of the observed errors; raises ParityError on mismatch."""
    out = {}
    t_dev, U_dev, d_dev = ctrl.get_state()
    chk.adopt(t_dev, U_dev, d_dev, x_dev.cpu().numpy())
    ctrl.closed_loop_device(x_dev, u_dev, 1)
    sync()
    chk.advance(1)
    out["continuation_1tick_max_err"] = chk.compare("one teacher-forced tick from the timed state", x_dev.cpu().numpy(),
                                                    u_dev.cpu().numpy(), ctrl.get_status()[0], tol_1tick, strict_counts)
    flips = chk.flips
    if free_ticks:
        ctrl.closed_loop_device(x_dev, u_dev, free_ticks)
        sync()
        chk.advance(free_ticks)
        out[f"continuation_{free_ticks}tick_fused_max_err"] = chk.compare(
            f"{free_ticks} fused ticks after the timed state", x_dev.cpu().numpy(), u_dev.cpu().numpy(),
            ctrl.get_status()[0], tol_free, False)
    out["arnoldi_count_flips"] = flips + chk.flips
    out["instances_checked"] = len(chk.idx)
    return out
