#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of serial steps of a given batch size: scripts/kstats.sh <tag> <pairs-per-gpu> [steps]
tag=$1; B=$2; K=${3:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -- python3 $R/bench.py --steps $K --warmup 2 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --min-seconds 0 --no-events --pairs-per-gpu $B > /dev/null 2>&1
python3 - <<PY
import csv,glob,os
f=max(glob.glob('$R/gpurun_out/ks_$tag/**/*kernel_stats.csv', recursive=True), key=os.path.getsize)
steps=$K+2
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void ','')
    t=float(r['TotalDurationNs'])/1e3/steps; tot+=t
    print(f"{n[:40]:40s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  per step {t:8.1f} us  per 8 pairs {t*8/$B:8.1f}")
print(f"sum of kernel time per step {tot:.0f} us = {tot*8/$B:.0f} us per 8 pairs")
PY
