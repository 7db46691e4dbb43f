#!/usr/bin/env python3
"""Dev helper: build A/B variants of the library with extra -D flags into gpurun_out/ab/<name>/ and (on the GPU
box) time bench.py against each through CGMRES_HIP_LIB.
    python tools/ab_build.py build name1:-DFLAG1 name2:-DFLAG2,-DFLAG3 ...
    python tools/ab_build.py bench name1 name2 ...      (prints ms_per_step per variant, interleaved rounds)"""
import json, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cgmres_cpp_amd import build as b
AB = os.path.join(ROOT, "_ab")  # git-ignored, but travels with gpurun (gpurun_out/ does not)

def build(name, flags, only=("capi.hip", "inst_pendulum_f64.hip")):
    d = os.path.join(AB, name); os.makedirs(d, exist_ok=True)
    srcs, _ = b.sources()
    if os.environ.get("AB_SRC_ROOT"):  # another checkout of the sources (e.g. `git --work-tree=/tmp/old checkout HEAD -- cgmres_cpp_amd/csrc include`)
        root = os.environ["AB_SRC_ROOT"]
        srcs = [os.path.join(root, os.path.relpath(s, ROOT)) for s in srcs]
        flags = flags + ["-I" + os.path.join(root, "include")]
    def cc(s):
        o = os.path.join(d, os.path.basename(s)[:-4] + ".o")
        subprocess.run([b.HIPCC] + b.CFLAGS + flags + ["-c", "-o", o, s], check=True)
        return o
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", os.path.join(d, "lib.so")] + objs, check=True)

if sys.argv[1] == "build":
    for spec in sys.argv[2:]:
        name, _, fl = spec.partition(":")
        build(name, [f for f in fl.split(",") if f])
        print("built", name)
else:
    names = sys.argv[2:]
    res = {n: [] for n in names}
    for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
        for n in names:
            env = dict(os.environ)
            lib, _, sw = n.partition("@")  # name@serial: bench.py --flags 1 (CGMRES_HIP_FLAG_SERIAL_COSTATE)
            extra = ["--flags", str(int(sw[2:], 0))] if sw.startswith("f=") else {"serial": ["--flags", "1"], "twopass": ["--flags", "8"], "": []}[sw]  # CGMRES_HIP_FLAG_*
            if lib != "base":
                env["CGMRES_HIP_LIB"] = os.path.join(AB, lib, "lib.so")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-ref-mode", "--steps", "150", "--warmup", "30"] + extra + os.environ.get("AB_BENCH_ARGS", "").split(),
                               env=env, capture_output=True, text=True)
            res[n].append(json.loads(r.stdout.strip().split("\n")[-1])["ms_per_step"])
    for n in names:
        print(f"{n:20s} ms/step min {min(res[n]):.4f}  all {['%.4f' % v for v in res[n]]}")
