#!/bin/bash
# kernel stats of the DEFAULT bench (four chains in flight): durations are under concurrency. scripts/kstats_bench.sh <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kb_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kb_$tag -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $R/gpurun_out/kb_$tag.json 2>/dev/null
python3 - <<PY
import csv,glob,os,json
f=max(glob.glob('$R/gpurun_out/kb_$tag/**/*kernel_stats.csv', recursive=True), key=os.path.getsize)
d=json.loads([l for l in open('$R/gpurun_out/kb_$tag.json') if l.startswith('{')][-1])
print('traced run: value', round(d['value']), 'ms/step', round(d['ms_per_step'],4))
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    n=r['Name'].split('(')[0].replace('void ','')
    print(f"{n[:36]:36s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  share {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
