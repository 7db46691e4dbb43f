set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_icache
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d "$O/a" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-ref-mode > "$O/a.json" 2> "$O/a.err" || tail -5 "$O/a.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d "$O/b" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-ref-mode > "$O/b.json" 2> "$O/b.err" || tail -5 "$O/b.err"
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for f in glob.glob("$O/"+d+"/*/*counter_collection.csv"):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tick_wg" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d,k,sum(v)/len(v),len(v))
PY
