#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/profile_bench.sh <tag>
# The headline bench command under rocprofv3 (tools/profile_cmd.sh: kernel trace + stats, then one PMC pass per TCC
# counter).  Steps and warm-up are multiples of the 10 ticks a launch advances, one timed repetition and no oracle
# continuation (its 11 ticks would add a 1-tick launch), so every launch in the trace is the same amount of work.
exec bash "$(dirname "$0")/profile_cmd.sh" "${1:-bench}" bench.py --steps 100 --warmup 20 --reps 1 --check-sample 0 --no-cpu-baseline --no-ref-mode ${BENCH_ARGS:-}
