#!/bin/bash
# per-kernel totals of config 5 (one scan-to-map registration per step) under rocprofv3: scripts/kstat5.sh <tag> [ENV=..]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/k5_$tag
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k5_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload config5 --steps 10 --warmup 2 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/k5_$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print(r['Name'].split('(')[0].replace('void ',''), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
