#!/usr/bin/env python3
"""Phase timing helper (run under rocprofv3 --kernel-trace --stats): calls the white-box hooks of the
C ABI at the headline size so each phase shows up as its own kernel in the trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cgmres_cpp_amd as cg
from cgmres_cpp_amd import scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x0, u0, p = scenarios.batch("pendulum", B)
c = cg.CgmresBatch("pendulum", batch=B, dv=50, k_max=10, tol=0.0, variant=variant)
c.set_ptau_repeat(p); c.init_u0(u0); c.init_u0_newton(u0, x0, p, 10)
x = x0.copy()
for _ in range(20):
    u = c.control(x)
t, U, d = c.get_state()
for rep in range(5):
    F = c.F_func(U, x, t)
    b = c.prepare(x)
    ax = c.Ax_func(d)
    sol = c.gmres(d, b)
    u = c.control(x)
print("done", float(np.abs(u).max()))
