#!/bin/bash
# PMC counters per kernel at a given batch size: scripts/pmc_k.sh <pairs-per-gpu> "CTR1 CTR2 ..." ["CTRA CTRB ..."]...   (one rocprofv3 pass per quoted group)
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pk_$i; echo "pass $i: $set"   # (a line per pass: a silent GPU command is taken to be hung after 7 minutes)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pk_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --no-events --min-seconds 0 --pairs-per-gpu $B > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
R="$R"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(R+'/gpurun_out/pk_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in ('k_nn_tiles','k_nn_iter<true>','k_icp_late<1>','k_accumulate_matches<1, true>'):
    if k in acc: print(k, {c: round(v/cnt[k][c]) for c,v in sorted(acc[k].items())})
PY
