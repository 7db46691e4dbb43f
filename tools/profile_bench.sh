#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/profile_bench.sh <tag>
# Three rocprofv3 runs of the SAME bench command: kernel trace + stats, then one PMC pass per TCC counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).  Steps and warm-up are multiples of the 10 ticks a
# launch of the tick kernel advances, so every launch in the trace is the same amount of work.  Output: gpurun_out/prof_<tag>/ ; summarise with
# tools/summarise_profile.py and commit the summaries under profiles/.
set -eo pipefail
TAG=${1:-bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-ref-mode ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --steps 100 --warmup 20 $ARGS > "$O/bench_trace.json" 2> "$O/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 20 --warmup 10 $ARGS > "$O/bench_fetch.json" 2> "$O/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 20 --warmup 10 $ARGS > "$O/bench_write.json" 2> "$O/write.err"
echo done "$O"
