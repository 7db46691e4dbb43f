"""Builds libcgmres_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m cgmres_cpp_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcgmres_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function"]


def sources():
    deps = [os.path.join(INCLUDE, "cgmres_hip.h")]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            deps.append(os.path.join(CSRC, f))
    return [os.path.join(CSRC, "capi.hip")], deps


def up_to_date():
    if not os.path.exists(LIB_PATH):
        return False
    t = os.path.getmtime(LIB_PATH)
    return all(os.path.getmtime(d) <= t for d in sources()[1])


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs, _ = sources()
    cmd = [HIPCC] + FLAGS + ["-o", LIB_PATH] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libcgmres_hip.so")
    if verbose and r.stderr.strip():
        sys.stderr.write(r.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
