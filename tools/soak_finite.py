#!/usr/bin/env python3
"""Dev helper: long closed loop of the headline batch; reports the first tick block with a non-finite u/x and the
Arnoldi-count histogram every 500 ticks.   python tools/soak_finite.py [ticks] [tol] [batch] [flags]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cgmres_cpp_amd as cg
from cgmres_cpp_amd import scenarios
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
FLAGS = int(sys.argv[4]) if len(sys.argv) > 4 else 0
x0, u0, p = scenarios.batch("pendulum", B)
c = cg.CgmresBatch("pendulum", batch=B, dv=50, k_max=10, tol=tol, flags=FLAGS)
print(c.variant_name, "B", B, "tol", tol, "flags", FLAGS, flush=True)
c.set_ptau_repeat(p); c.init_u0(u0); c.init_u0_newton(u0, x0, p, 10)
xd = c.device_buffer((B, 4)).upload(x0); ud = c.device_buffer((B, 3))
for t in range(0, N, 100):
    c.closed_loop_device(xd, ud, 100); c.synchronize()
    x, u = xd.download(), ud.download()
    bad = ~(np.isfinite(x).all(axis=1) & np.isfinite(u).all(axis=1))
    if t % 500 == 400 or bad.any():
        print(t + 100, "bad", int(bad.sum()), np.nonzero(bad)[0][:8], "k", np.bincount(c.get_status()[0], minlength=11), "reason", np.bincount(c.get_status()[1], minlength=5),
              "max|x2|,|x3|", float(np.nanmax(np.abs(x[:, 2]))), float(np.nanmax(np.abs(x[:, 3]))), flush=True)
    if bad.any():
        break
