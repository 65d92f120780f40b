#!/bin/bash
# k_nn_tiles / k_nn_iter / k_icp_late / k_accumulate average durations for a given lib and batch size: scripts/kt.sh <lib or -> <pairs>
R=${GRAFT_REPO_ROOT:-/root/repo}
lib=$1; B=$2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_tmp
if [ "$lib" != "-" ]; then export M3DREG_LIB=$R/$lib; fi; libname=${M3DREG_LIB:-in-tree}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_tmp -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --min-seconds 0 --no-events --pairs-per-gpu $B > /dev/null 2>&1
python3 - <<PY
import csv,glob,os
f=max(glob.glob('$R/gpurun_out/kt_tmp/**/*kernel_stats.csv', recursive=True), key=os.path.getsize)
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void ','')
    if any(k in n for k in ('k_nn_tiles','k_nn_iter','k_icp_late','k_accumulate','k_tile_build')): out.append(f"{n[:28]} {float(r['AverageNs'])/1e3*8/$B:.1f}")
print('$libname B=$B (us per launch per 8 pairs):', ' | '.join(out))
PY
