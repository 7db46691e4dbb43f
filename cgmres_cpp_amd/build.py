"""Builds libcgmres_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m cgmres_cpp_amd.build [--force]

Every .hip file under csrc/ is one translation unit (the kernel instantiations are split per model and
scalar type so they compile in parallel); objects go to csrc/_obj/, the library to lib/.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcgmres_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def sources():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return srcs, hdrs + [os.path.join(INCLUDE, "cgmres_hip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def up_to_date():
    srcs, hdrs = sources()
    return not _stale(LIB_PATH, srcs + hdrs)


def _compile(src, hdrs, force, verbose):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    if not force and not _stale(obj, [src] + hdrs):
        return obj, None
    cmd = [HIPCC] + CFLAGS + ["-c", "-o", obj, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    return obj, (r if r.returncode != 0 or r.stderr.strip() else None)


def build(force=False, verbose=False, jobs=None):
    if not force and up_to_date():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs, hdrs = sources()
    jobs = jobs or min(len(srcs), max(1, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, hdrs, force, verbose), srcs))
    failed = False
    for obj, r in results:
        if r is not None:
            sys.stderr.write(r.stdout + r.stderr)
            failed = failed or r.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed building libcgmres_hip.so")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB_PATH] + [o for o, _ in results]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link of libcgmres_hip.so failed")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
