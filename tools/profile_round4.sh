#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/profile_round4.sh [traces|counters]   (default: both halves)
# rocprofv3 passes behind the committed profiles/r04_* summaries: the headline bench at the four per-GPU batch sizes of
# the strong-scaling run (4096 on the wg mapping; 2048 / 1024 / 512 on the wave mapping), the instruction counters of
# the wave kernel (its own --pmc pass: SQ_* counters, no trace domains), the in-kernel phase stamps of the wave kernel,
# and the other BASELINE configurations.  Summaries: tools/summarise_profile.py <tag> <name> afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
if [ "${1:-all}" != "counters" ]; then
  bash tools/profile_bench.sh r04_wg_bench || exit 1
  BENCH_ARGS="--flags 128" bash tools/profile_bench.sh r04_wg_serial_bench || exit 1   # (the wg kernel with the serial state sweep)
  for b in 2048 1024 512; do
    BENCH_ARGS="--batch $b" bash tools/profile_bench.sh r04_wave_bench_B$b || exit 1
  done
fi
[ "${1:-all}" = "traces" ] && exit 0
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d "$R/gpurun_out/prof_r04_wave_sq" -- python3 "$R/bench.py" --batch 1024 --steps 100 --warmup 20 --reps 1 \
    --check-sample 0 --no-cpu-baseline --no-ref-mode > "$R/gpurun_out/prof_r04_wave_sq.json" 2> "$R/gpurun_out/prof_r04_wave_sq.err" ) || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d "$R/gpurun_out/prof_r04_wg_sq" -- python3 "$R/bench.py" --steps 100 --warmup 20 --reps 1 \
    --check-sample 0 --no-cpu-baseline --no-ref-mode > "$R/gpurun_out/prof_r04_wg_sq.json" 2> "$R/gpurun_out/prof_r04_wg_sq.err" ) || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d "$R/gpurun_out/prof_r04_wg_serial_sq" -- python3 "$R/bench.py" --steps 100 --warmup 20 --reps 1 --flags 128 \
    --check-sample 0 --no-cpu-baseline --no-ref-mode > "$R/gpurun_out/prof_r04_wg_serial_sq.json" 2> "$R/gpurun_out/prof_r04_wg_serial_sq.err" ) || exit 1
python3 tools/phase_stamps.py --model=pendulum --variant=4 --batch=512 > gpurun_out/r04_wave_phase_stamps.txt 2>&1 || exit 1
python3 tools/phase_stamps.py --model=pendulum --variant=2 > gpurun_out/r04_wg_phase_stamps.txt 2>&1 || exit 1
python3 tools/phase_stamps.py --model=pendulum --variant=2 --flags=128 > gpurun_out/r04_wg_serial_phase_stamps.txt 2>&1 || exit 1
python3 tools/bench_configs.py --config all > gpurun_out/r04_other_configs.jsonl 2> gpurun_out/r04_other_configs.err || exit 1
python3 tools/bench_configs.py --config gmres >> gpurun_out/r04_other_configs.jsonl 2>> gpurun_out/r04_other_configs.err || exit 1
echo profiles done
