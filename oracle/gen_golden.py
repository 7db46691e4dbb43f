#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz from the UNMODIFIED reference
compiled into oracle/_ref/libref.so (oracle/Makefile `ref`).  Runs only where /root/reference is
mounted; the fixtures (pure data: inputs and the reference's outputs) are committed.

    python oracle/gen_golden.py            # rewrites every fixture

Record kinds, per case = (model, dv, k_max, tol mode, dtype)  [SURVEY.md §8(c)]:
  tick<N>_*   teacher-forced control record of the shipped closed-loop scenario at tick N:
              inputs  t, x, U, dUdt            (controller state before the tick, plant state)
              outputs u, U1, dUdt1, n_ax, H, rho, g, b, Fh, xh, F0  (b = GMRES rhs, Fh = F_dxh_h,
              F0 = F(U,x,t) as a plain F_func record, Ax_v/Ax_out = one Ax_func record on a seeded v)
  loop_u / loop_x / loop_k   the first ticks of the closed loop (u, plant x after the tick, Arnoldi count)
  batch_*     8 seeded perturbed instances (splitmix64 seed 12345): x0, u0 (after Newton), p, and
              u / k for the first ticks of each instance's own closed loop
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# (model, dv, kmax, tol, dtype, record ticks, closed-loop length stored, batch ticks)
CASES = [
    (orc.PENDULUM, 25, 5, -1.0, "f64", (0, 1, 2, 10, 100, 1000), 101, 20),
    (orc.PENDULUM, 50, 10, -1.0, "f64", (0, 1, 2, 10, 100, 1000, 3437), 101, 30),
    (orc.PENDULUM, 50, 10, 0.0, "f64", (0, 1, 10, 100), 101, 20),
    (orc.PENDULUM, 100, 20, -1.0, "f64", (0, 1, 10, 100), 51, 10),
    (orc.PENDULUM, 8, 3, -1.0, "f64", (0, 1, 2, 10, 100), 101, 20),
    (orc.MSD, 50, 5, -1.0, "f64", (0, 1, 2, 10, 100, 1000), 101, 20),
    (orc.MSD, 20, 5, -1.0, "f64", (0, 1, 2, 10, 100, 1000), 101, 20),
    (orc.MSD, 50, 10, -1.0, "f64", (0, 1, 10, 100), 101, 20),
    (orc.MSD, 8, 3, -1.0, "f64", (0, 1, 2, 10, 100), 101, 20),
    (orc.SEMIACTIVE, 50, 5, -1.0, "f64", (0, 1, 2, 10, 100, 1000), 101, 20),
    (orc.SEMIACTIVE, 50, 10, -1.0, "f64", (0, 1, 10, 100, 1000), 101, 20),
    (orc.SEMIACTIVE, 50, 10, 0.0, "f64", (0, 1, 10, 100), 51, 10),
    (orc.SEMIACTIVE, 8, 3, -1.0, "f64", (0, 1, 2, 10, 100), 101, 20),
    # fp32 reference (`#define double float` build of the same headers)
    (orc.PENDULUM, 100, 20, -1.0, "f32", (0, 1, 10, 100), 51, 10),
    (orc.PENDULUM, 50, 10, -1.0, "f32", (0, 1, 10, 100), 51, 10),
    (orc.MSD, 50, 10, -1.0, "f32", (0, 1, 10, 100), 51, 10),
    (orc.SEMIACTIVE, 50, 10, -1.0, "f32", (0, 1, 10, 100), 51, 10),
]


def case_name(model, dv, kmax, tol, dtype):
    return f"{orc.MODEL_NAMES[model]}_dv{dv}_k{kmax}_{'tol0' if tol == 0.0 else 'tolref'}_{dtype}"


def one_case(model, dv, kmax, tol, dtype, ticks, n_loop, n_batch_ticks):
    rec = {}
    c = orc.Controller(model, dv, kmax, tol, dtype, which="ref")
    x0, u0, p = orc.shipped_scenario(model)
    un = orc.start_controller(c, x0, u0, p)
    rec["meta"] = np.array([model, dv, kmax, c.dim_x, c.dim_u, c.dim_p, 1 if dtype == "f32" else 0], dtype=np.int64)
    rec["tol"] = np.array([1e-6 if tol < 0 else tol])
    rec["tuning"] = np.array([c.dt, c.h, c.zeta, c.Tf, c.alpha])
    rec["x0"], rec["u0_guess"], rec["u0_newton"], rec["p"] = x0, u0, un, p
    rec["ptau"] = np.tile(p, dv + 1)
    rng = np.random.default_rng(20251003 + 7 * model + dv)
    x = x0.copy()
    n_total = max(max(ticks) + 1, n_loop)
    us, xs, ks = [], [], []
    for tick in range(n_total):
        if tick in ticks:
            t, U, d = c.get_state()
            pre = f"tick{tick}_"
            rec[pre + "t"], rec[pre + "x"], rec[pre + "U"], rec[pre + "dUdt"] = np.array([t]), x.copy(), U, d
            # white-box records on a twin controller in the same state (keeps the main loop untouched)
            w = orc.Controller(model, dv, kmax, tol, dtype, which="ref")
            if c.dim_p:
                w.set_ptau_repeat(p)
            w.set_state(t, U, d)
            rec[pre + "F0"] = w.F(U, x, t)
            rec[pre + "b"] = w.prepare(x)
            v = rng.standard_normal(c.len)
            rec[pre + "Ax_v"], rec[pre + "Ax_out"] = v, w.Ax(v)
            sol = w.gmres(d, rec[pre + "b"])
            rec[pre + "gmres_x"] = sol
            rec[pre + "gmres_nax"] = np.array([w.last_solve()[0]])
        u = c.control(x)
        k = c.last_solve()[0]
        if tick in ticks:
            t1, U1, d1 = c.get_state()
            V, H, rho, g = c.krylov()
            rec[pre + "u"], rec[pre + "U1"], rec[pre + "dUdt1"] = u, U1, d1
            rec[pre + "n_ax"] = np.array([k])
            rec[pre + "H"], rec[pre + "rho"], rec[pre + "g"] = H, rho, g
            assert np.array_equal(sol, d1) and k == rec[pre + "gmres_nax"][0]
        x = x + c.plant(x, u) * c.dt
        if tick < n_loop:
            us.append(u), xs.append(x.copy()), ks.append(k)
    rec["loop_u"], rec["loop_x"], rec["loop_k"] = np.array(us), np.array(xs), np.array(ks)

    # seeded perturbed batch (SURVEY.md §8d recipe)
    nb = 8
    bx0, bu0, bp = orc.batch_scenario(model, nb)
    bun, bus, bks, bxs = [], [], [], []
    for i in range(nb):
        ci = orc.Controller(model, dv, kmax, tol, dtype, which="ref")
        bun.append(orc.start_controller(ci, bx0[i], bu0[i], bp[i]))
        u_i, x_i, k_i, _ = orc.closed_loop(ci, bx0[i], n_batch_ticks)
        bus.append(u_i), bks.append(k_i), bxs.append(x_i)
    rec["batch_x0"], rec["batch_u0_guess"], rec["batch_p"] = bx0, bu0, bp
    rec["batch_u0_newton"] = np.array(bun)
    rec["batch_u"], rec["batch_k"], rec["batch_x"] = np.array(bus), np.array(bks), np.array(bxs)
    return rec


def main():
    if not orc.have_ref():
        orc.build()
    if not orc.have_ref():
        sys.exit("oracle/_ref/libref.so unavailable: /root/reference is not mounted here")
    os.makedirs(OUT, exist_ok=True)
    total = 0
    for case in CASES:
        name = case_name(*case[:5])
        rec = one_case(*case)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **rec)
        sz = os.path.getsize(path)
        total += sz
        print(f"{name:40s} {sz/1024:8.1f} KiB  k-hist(loop)={np.bincount(rec['loop_k'], minlength=case[2]+1).tolist()}")
    print(f"total {total/1024:.1f} KiB in {OUT}")


if __name__ == "__main__":
    main()
