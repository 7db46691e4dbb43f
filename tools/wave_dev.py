#!/usr/bin/env python3
"""Development check of a kernel mapping against the oracle + closed-loop timing at several batch sizes.

    python tools/wave_dev.py [--variant 4] [--batches 256,512,1024] [--ticks 200] [--tol 0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cgmres_cpp_amd as cg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=int, default=4)
    ap.add_argument("--batches", default="256,512,1024")
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--warm", type=int, default=50)
    ap.add_argument("--tol", type=float, default=0.0)
    ap.add_argument("--dv", type=int, default=50)
    ap.add_argument("--kmax", type=int, default=10)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--model", type=int, default=0, help="0 pendulum, 1 msd, 2 semiactive")
    args = ap.parse_args()
    from oracle import orc
    model = args.model
    for B in [int(s) for s in args.batches.split(",")]:
        x0, u0, p = orc.batch_scenario(model, B)
        c = cg.CgmresBatch(model, batch=B, dv=args.dv, k_max=args.kmax, tol=args.tol, variant=args.variant,
                           flags=args.flags)
        c.set_ptau_repeat(p)
        c.init_u0(u0)
        c.init_u0_newton(u0, x0, p, 10)
        xd = c.device_buffer((B, c.dim_x))
        ud = c.device_buffer((B, c.dim_u))
        xd.upload(x0)
        out = {"variant": c.variant, "name": c.variant_name, "B": B, "tol": args.tol}
        if not args.no_check:
            # free-running closed loop of a sample against the oracle
            idx = sorted(set(np.linspace(0, B - 1, 12).astype(int).tolist()))
            n = 25
            c.closed_loop_device(xd, ud, n)
            c.synchronize()
            xg, ug = xd.download(), ud.download()
            n_ax, reason = c.get_status()
            t, Ug, dg = c.get_state()
            err = 0.0
            errd = 0.0
            bad_k = 0
            for i in idx:
                r = orc.Controller(model, args.dv, args.kmax, args.tol)
                orc.start_controller(r, x0[i], u0[i], p[i])
                x = x0[i].copy()
                for _ in range(n):
                    u = r.control(x)
                    x = x + r.plant(x, u) * r.dt
                err = max(err, float(np.max(np.abs(u - ug[i]))), float(np.max(np.abs(x - xg[i]))))
                errd = max(errd, float(np.max(np.abs(r.get_state()[2] - dg[i])) / max(1.0, np.max(np.abs(dg[i])))))
                bad_k += int(n_ax[i] != r.last_solve()[0])
            out.update(err_u_x=err, err_dUdt_rel=errd, n_ax_mismatch=bad_k, n_ax_hist=np.bincount(n_ax).tolist())
        c.closed_loop_device(xd, ud, args.warm)
        c.synchronize()
        best = 1e9
        for rep in range(3):
            c.timer_start()
            c.closed_loop_device(xd, ud, args.ticks)
            ms = c.timer_stop()
            best = min(best, ms)
        out.update(us_per_tick=1e3 * best / args.ticks, Msteps_s=B * args.ticks / best / 1e3)
        print(json.dumps(out), flush=True)
        c.close()


if __name__ == "__main__":
    main()
