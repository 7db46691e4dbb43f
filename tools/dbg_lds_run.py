import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import cgmres_cpp_amd as cg
from cgmres_cpp_amd import scenarios
for (model, B, dv, km) in [("pendulum", 40, 50, 10), ("pendulum", 20, 25, 5), ("pendulum", 20, 8, 3), ("msd", 20, 20, 5), ("semiactive", 20, 50, 10), ("msd", 20, 50, 10)]:
    x0, u0, p = scenarios.batch(model, B)
    c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=km, tol=1e-6)
    if p.shape[1]: c.set_ptau_repeat(p)
    c.init_u0(u0); c.init_u0_newton(u0, x0, p, 10)
    x = x0.copy()
    for t in range(3):
        u = c.control(x)
    print(model, dv, km, "ok", np.isfinite(u).all(), flush=True)
    c.close()
