#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/profile_cmd.sh <tag> <script.py> [args...]
# Three rocprofv3 runs of the SAME python command: kernel trace + stats, then one PMC pass per TCC counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; --pmc is never combined with other trace domains).
# Output: gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write}/ + the command's stdout per pass; summarise with
# tools/summarise_profile.py <tag> <name> and commit the summaries under profiles/.
set -eo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$SCRIPT" "$@" > "$O/out_trace.json" 2> "$O/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$SCRIPT" "$@" > "$O/out_fetch.json" 2> "$O/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$SCRIPT" "$@" > "$O/out_write.json" 2> "$O/write.err"
echo "$SCRIPT $*" > "$O/command.txt"
echo done "$O"
