#!/usr/bin/env python3
"""Diagnostic (never part of the product build): rebuilds the library with -DCGM_STAMPS into gpurun_out/diag/,
runs closed-loop ticks at the headline size and prints where block 0 spends its shader cycles."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cgmres_cpp_amd import build as b

diag = os.path.join(ROOT, "_diag")  # git-ignored, but travels with gpurun (gpurun_out/ does not)
os.makedirs(diag, exist_ok=True)
MODEL = "msd" if "--model=msd" in sys.argv else ("pendulum32" if "--model=pendulum32" in sys.argv else "pendulum")
MODEL_CODE = {"pendulum": 0, "msd": 1, "pendulum32": 2}[MODEL]
def opt(name, default):
    for a in sys.argv:
        if a.startswith(f"--{name}="):
            return int(a.split("=")[1])
    return default
lib = os.path.join(diag, f"libcgmres_hip_stamps_{MODEL}.so")
for a in sys.argv:
    if a.startswith("--lib="):  # a stamp build made earlier (e.g. an A/B pair copied aside)
        lib = os.path.abspath(a[6:])
if "--build" in sys.argv or "--build-only" in sys.argv or not os.path.exists(lib):
    srcs, _ = b.sources()
    from concurrent.futures import ThreadPoolExecutor
    def cc(s):
        o = os.path.join(diag, os.path.basename(s)[:-4] + f"_{MODEL}.o")
        subprocess.run([b.HIPCC] + b.CFLAGS + [a for a in sys.argv if a.startswith("-D")] + ["-DCGM_STAMPS", f"-DCGM_STAMPS_MODEL={MODEL_CODE}", "-c", "-o", o, s], check=True)
        return o
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", lib] + objs, check=True)
    if "--build-only" in sys.argv:
        sys.exit(0)
b.LIB_PATH = lib
import cgmres_cpp_amd as cg
from cgmres_cpp_amd import scenarios
B, DV, KM = opt("batch", 4096), opt("dv", 50), opt("kmax", 10)
NAME = "pendulum" if MODEL == "pendulum32" else MODEL
DT = "f32" if MODEL == "pendulum32" else "f64"
x0, u0, p = scenarios.batch(NAME, B)
c = cg.CgmresBatch(NAME, batch=B, dv=DV, k_max=KM, tol=0.0, dtype=DT, variant=opt("variant", 0), flags=opt("flags", 0))
print("variant", c.variant, c.variant_name, "B", B, "dv", DV, "kmax", KM, DT)
x0, u0 = x0.astype(c.np_dtype), u0.astype(c.np_dtype)
c.set_ptau_repeat(p); c.init_u0(u0); c.init_u0_newton(u0, x0, p, 10)
xd = c.device_buffer(x0.shape).upload(x0); ud = c.device_buffer(u0.shape)
c.closed_loop_device(xd, ud, 50); c.synchronize()
L = cg.load()
out = (ctypes.c_longlong * 64)()
L.cgmres_hip_debug_stamps(out)
N = 100
c.closed_loop_device(xd, ud, N); c.synchronize()
L.cgmres_hip_debug_stamps(out)
names = {14: "loop top: before barrier_or", 15: "barrier_or (drains V row store)", 0: "prologue loads", 1: "preamble (2 sweeps)", 3: "ring preload issue", 4: "sweep phase 1 (state)",
         5: "sweep phase 2 (coeffs)", 18: "costate A: prologue (terminal costate, first fetches)", 19: "costate A: stage loop", 20: "costate A: tail stages",
         16: "costate A: four chunks side by side (record store when 18-20 are stamped)", 17: "costate: barrier",
         6: "sweep phase 3 (costate; B: boundaries + combine if chunk-parallel)", 7: "MGS rounds", 8: "norm+normalise+store",
         21: "row Newton: x0/x2 scans", 22: "row Newton: iterations (visits = iterations)", 23: "row Newton: costate scans + out", 24: "row Newton: operands + stage coefficients",
         9: "Hessenberg scalar", 10: "loop exit barrier", 11: "back-subst", 12: "x update (V*y)", 13: "epilogue"}
if c.variant == 4:  # the wave mapping's own stamp ids (tick_wave.hip.h)
    names = {11: "tick top / epilogue tail", 0: "x+hf, control rows", 1: "serial state sweeps (3 quads)", 2: "preamble: 3 x costate scans",
             3: "ax: x0/x2 scans", 4: "ax: Newton iterations (visits = iterations)", 12: "ax: after Newton (first-order fix)", 5: "ax: costate scans + dH/du",
             9: "r0", 6: "MGS rounds", 7: "norm+normalise+store", 8: "Hessenberg scalar", 13: "loop exit",
             10: "back-subst + x update"}
tot = out[29]; wall = out[28]
print(f"ticks {N}: shader cycles/tick {tot/N:.0f}, wall {wall/N/100:.1f} us/tick -> clock {tot/wall*100/1e3:.2f} GHz")
for k, n in names.items():
    print(f"  {n:34s} {out[k]/N:10.0f} cyc/tick  {100*out[k]/tot:5.1f}%   ({out[32+k]/N:.0f} visits)")
print("  accounted", sum(out[k] for k in names) / tot)
