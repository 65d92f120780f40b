#!/bin/bash
# round 4 work loop on the GPU box: gpu tests (optional), headline + serial bench, kernel stats of serial steps.  scripts/quick.sh <tag> [notest]
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$2" != "notest" ]; then
  python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log
fi
for mode in "" "--inflight 1 --queue-depth 1"; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra $mode 2>gpurun_out/${tag}_bench.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '$mode', round(d['value']), round(d['ms_per_step'],4), 'bucketing alone ms', round(d['ms_bucketing_batch_alone'],4), 'nn alone', round(d['roofline']['alone']['avg_launch_ms'],4), 'rot', d['max_rot_err_deg'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --min-seconds 0 --no-events > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=max(glob.glob('$R/gpurun_out/ks_$tag/**/*kernel_stats.csv', recursive=True), key=lambda x: __import__('os').path.getsize(x))
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    n=r['Name'].split('(')[0].replace('void ','')
    calls=int(r['Calls']); avg=float(r['AverageNs'])/1e3
    print(f"{n[:40]:40s} calls {calls:5d} avg {avg:8.1f} us  total/step {float(r['TotalDurationNs'])/1e3/7:8.1f}")
PY
