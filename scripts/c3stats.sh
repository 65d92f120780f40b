#!/bin/bash
# config 3 (ONE 100 k pair per step, serial): kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c3s
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3s -- python3 $R/bench.py --workload config3 --steps 10 --warmup 2 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --min-seconds 0 --no-events > /dev/null 2>&1
python3 - <<PY
import csv,glob,os
f=max(glob.glob('$R/gpurun_out/c3s/**/*kernel_stats.csv', recursive=True), key=os.path.getsize)
steps=12; tot=0
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void ','')
    t=float(r['TotalDurationNs'])/1e3/steps; tot+=t
    print(f"{n[:36]:36s} calls/step {int(r['Calls'])/steps:5.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  per step {t:7.1f}")
print('sum', round(tot))
PY
