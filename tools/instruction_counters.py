#!/usr/bin/env python3
"""Summary of a rocprofv3 --pmc pass with SQ instruction counters (tools/profile_round4.sh: SQ_INSTS_VALU, SQ_INSTS_SALU,
SQ_INSTS_LDS, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAVES in one pass of their own, no trace domains) for the tick kernel
of the profiled command -> profiles/<name>.json.

    python tools/instruction_counters.py <pass dir under gpurun_out/> <kernel substring> <batch> <waves per controller> <name>

Per launch averages, then per CU and tick / per controller and tick, and the share of a SIMD's VALU issue slots the
kernel uses: VALU instructions per SIMD x 4 cycles (a wave64 VALU instruction occupies its SIMD for 4 cycles) / cycles
of a tick, the latter from SQ_WAVE_CYCLES (counted in units of 4 cycles per resident wave) / SQ_WAVES."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pass_dir, kernel, batch, waves_per_ctrl, name = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
ticks_per_launch, n_cu = 10, 256
files = glob.glob(os.path.join(root, "gpurun_out", pass_dir, "**", "*counter_collection.csv"), recursive=True)
f = max(files, key=os.path.getmtime)
acc = collections.defaultdict(lambda: [0, 0.0])
kname = None
for r in csv.DictReader(open(f)):
    if kernel not in r["Kernel_Name"]:
        continue
    kname = r["Kernel_Name"]
    a = acc[r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
out = {k: v / n for k, (n, v) in acc.items()}
launches = max(n for n, _ in acc.values())
res = dict(out)
res["launches"] = launches
res["kernel"] = kname
res["per_cu_and_tick"] = {k: out[k] / n_cu / ticks_per_launch for k in out if k.startswith("SQ_INSTS")}
res["per_controller_and_tick"] = {k: out[k] / batch / ticks_per_launch for k in out if k.startswith("SQ_INSTS")}
cycles_per_tick = 4.0 * out["SQ_WAVE_CYCLES"] / out["SQ_WAVES"] / ticks_per_launch
res["cycles_per_tick"] = cycles_per_tick
waves_per_cu = out["SQ_WAVES"] / n_cu
res["waves_per_cu"] = waves_per_cu
res["valu_issue_utilisation_per_simd"] = out["SQ_INSTS_VALU"] / n_cu / 4 / ticks_per_launch * 4.0 / cycles_per_tick
try:
    sys.path.insert(0, root)
    from cgmres_cpp_amd import build as _b
    res["library_sha256_16"] = hashlib.sha256(open(os.environ.get("CGMRES_HIP_LIB") or _b.LIB_PATH, "rb").read()).hexdigest()[:16]
except OSError:
    res["library_sha256_16"] = None
res["note"] = (f"rocprofv3 --pmc (own pass, no trace domains) of `bench.py --steps 100 --warmup 20 --reps 1 --check-sample 0 "
               f"--no-cpu-baseline --no-ref-mode` at batch {batch}: {launches} launches of 10 ticks averaged; utilisation = VALU "
               f"instructions per SIMD x 4 cycles / cycles per tick (SQ_WAVE_CYCLES is counted in units of 4 cycles per wave)")
json.dump(res, open(os.path.join(root, "profiles", name + ".json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel", "per_controller_and_tick", "cycles_per_tick", "valu_issue_utilisation_per_simd")}, indent=1))
