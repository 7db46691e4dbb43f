#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/profile_round.sh
# All rocprofv3 passes behind the committed profiles/r<NN>_* summaries of a round: the headline bench at the four
# per-GPU batch sizes of the strong-scaling run (4096 / 2048 / 1024 / 512), the other BASELINE configurations, and the
# in-kernel phase stamps of the headline.  Summaries: tools/summarise_profile.py <tag> <name> afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
bash tools/profile_bench.sh wg_bench || exit 1
for b in 2048 1024 512; do
  BENCH_ARGS="--batch $b" bash tools/profile_bench.sh wg_bench_B$b || exit 1
done
bash tools/profile_cmd.sh cfg5 tools/bench_configs.py --config 5 --tols 0 --steps 100 --warmup 30 --check-sample 0 || exit 1
bash tools/profile_cmd.sh cfg4 tools/bench_configs.py --config 4 --tols 0 --steps 100 --warmup 30 --check-sample 0 || exit 1
bash tools/profile_cmd.sh cfg4msd tools/bench_configs.py --config 4msd --tols 0 --steps 100 --warmup 30 --check-sample 0 || exit 1
bash tools/profile_cmd.sh cfg3 tools/bench_configs.py --config 3 --tols 0 --steps 100 --warmup 30 --check-sample 0 || exit 1
echo profiles done
