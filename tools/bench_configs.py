#!/usr/bin/env python3
"""Secondary measurements (not bench lines): the other BASELINE.json configurations on one GPU, closed loop on the
device with the example plants, fixed-k mode (tol = 0) and reference mode (tol = 1e-6).
    python tools/bench_configs.py > gpurun_out/configs.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cgmres_cpp_amd as cg
from cgmres_cpp_amd import scenarios

NAMES = {0: "pendulum", 1: "msd", 2: "semiactive"}
DIMS = {0: (4, 3, 2), 1: (4, 6, 2), 2: (2, 3, 0)}


def alg_bytes(model, dv, k, scalar):
    nx, nu, npar = DIMS[model]
    L = nu * dv
    return scalar * (L * (8 + 6 * k + k * (k - 1) // 2) + (3 + k) * (npar * (dv + 1) + nx) + nu)


def run(model, B, dv, kmax, dtype, tol, steps=100, warmup=50):
    x0, u0, p = scenarios.batch(NAMES[model], B)
    npdt = np.float64 if dtype == "f64" else np.float32
    c = cg.CgmresBatch(model, batch=B, dv=dv, k_max=kmax, tol=tol, dtype=dtype)
    if DIMS[model][2]:
        c.set_ptau_repeat(p)
    c.init_u0(u0)
    c.init_u0_newton(u0, x0, p, 10)
    dev = torch.device("cuda", 0)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    x = torch.from_numpy(x0.astype(npdt)).to(dev)
    u = torch.zeros(B, DIMS[model][1], dtype=tdt, device=dev)
    c.closed_loop_device(x, u, warmup)
    torch.cuda.synchronize()
    c.timer_start()
    c.closed_loop_device(x, u, steps)
    ms = c.timer_stop() / steps
    torch.cuda.synchronize()
    n_ax, _ = c.get_status()
    ok = bool(torch.isfinite(u).all().item())
    var = c.variant
    c.close()
    k_eff = [int(k) for k in n_ax]
    by = float(sum(alg_bytes(model, dv, k, 8 if dtype == "f64" else 4) for k in k_eff)) if tol > 0 else \
        float(B * alg_bytes(model, dv, kmax, 8 if dtype == "f64" else 4))
    return {"model": NAMES[model], "batch": B, "dv": dv, "kmax": kmax, "dtype": dtype, "tol": tol, "variant": var,
            "ms_per_tick": ms, "steps_per_s": B / ms * 1e3, "mean_arnoldi_last_tick": float(np.mean(n_ax)),
            "algorithmic_GBps": by / ms / 1e6, "frac_of_8TBps": by / ms / 1e6 / 8000.0, "finite": ok}


out = []
for args in [(2, 4096, 50, 10, "f64"), (1, 4096, 50, 10, "f64"), (0, 4096, 50, 10, "f64"),
             (0, 8192, 100, 20, "f32"), (0, 256, 50, 10, "f64"), (1, 1, 20, 5, "f64")]:
    for tol in (0.0, 1e-6):
        out.append(run(*args, tol))
        print(json.dumps(out[-1]), flush=True)


def run_mixed(B_each=4096, dv=50, kmax=10, tol=0.0, steps=100, warmup=50):
    """BASELINE configs[4] on ONE GPU: Model1 (MSD) and Model2 (pendulum) batches stepped in the same loop on two
    streams (multiple_controller/main.cpp:104-110), host wall clock around both."""
    from cgmres_cpp_amd.multi import MultipleController
    dev = torch.device("cuda", 0)
    mc = MultipleController([dict(model=1, batch=B_each, dv=dv, k_max=kmax, tol=tol),
                             dict(model=0, batch=B_each, dv=dv, k_max=kmax, tol=tol)])
    xs, us = [], []
    for m, mid in zip(mc.members, (1, 0)):
        x0, u0, p = scenarios.batch(NAMES[mid], B_each)
        if DIMS[mid][2]:
            m.set_ptau_repeat(p)
        m.init_u0(u0)
        m.init_u0_newton(u0, x0, p, 10)
        xs.append(torch.from_numpy(x0).to(dev))
        us.append(torch.zeros(B_each, DIMS[mid][1], dtype=torch.float64, device=dev))
    mc.closed_loop_device(xs, us, warmup)
    mc.synchronize()
    t0 = time.perf_counter()
    mc.closed_loop_device(xs, us, steps)
    mc.synchronize()
    dt = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(u).all().item()) for u in us)
    by = B_each * (alg_bytes(1, dv, kmax, 8) + alg_bytes(0, dv, kmax, 8))
    mc.close()
    return {"model": "msd+pendulum (two streams)", "batch": 2 * B_each, "dv": dv, "kmax": kmax, "dtype": "f64",
            "tol": tol, "variant": 2, "ms_per_tick": dt / steps * 1e3, "steps_per_s": 2 * B_each * steps / dt,
            "mean_arnoldi_last_tick": float(kmax), "algorithmic_GBps": by / (dt / steps) / 1e9,
            "frac_of_8TBps": by / (dt / steps) / 1e9 / 8000.0, "finite": ok}


print(json.dumps(run_mixed()), flush=True)
