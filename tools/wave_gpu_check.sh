#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/wave_gpu_check.sh <tag>
# Development round trip for the wave mapping: timing at small batches, the variant-4 parity tests, and one rocprofv3
# PMC pass (instruction counts of the tick kernel; --pmc alone, no trace domains).
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
timeout -k 10 200 python tools/wave_dev.py --variant 4 --batches 256,1024 --ticks 100 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py tests/test_gpu_closed_loop.py -m gpu -q -k "wave or 4" > gpurun_out/r04_wave_t_$TAG.log 2>&1
echo tests rc=$?
tail -5 gpurun_out/r04_wave_t_$TAG.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$R/gpurun_out/r04_wave_pmc_$TAG" -- python3 "$R/tools/wave_dev.py" --variant 4 --batches 1024 --ticks 100 --no-check > "$R/gpurun_out/r04_wave_pmc_$TAG.log" 2>&1
echo pmc rc=$?
cd "$R" && python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
for f in glob.glob(f"gpurun_out/r04_wave_pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "tick_wave" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        print(k, "launches", n, "per launch", v / n)
PY
