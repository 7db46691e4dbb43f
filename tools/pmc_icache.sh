#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/pmc_icache.sh [bench args]
# Instruction-cache and instruction-fetch counters of the headline bench command, one rocprofv3 --pmc pass per group
# (kernel trace only, as the pool requires).  Output: gpurun_out/icache/<group>/ + the available-counter list.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/icache; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
grep -i -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_CACHE\|WAIT_INST\)[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt
cat $O/names.txt
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o pmc --output-format csv -- python3 $R/bench.py --steps 100 --warmup 20 --reps 1 --check-sample 0 --no-cpu-baseline --no-ref-mode "$@" > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; echo "group $i failed"; }
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/icache'
for f in sorted(glob.glob(O+'/g*/**/*counter_collection.csv',recursive=True)):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if 'tick_wg' not in r['Kernel_Name']: continue
        a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for k,(n,v) in acc.items(): print(f.split('/')[-3] if 'g' in f else f, k, 'launches',n,'per launch',v/n)
PY
