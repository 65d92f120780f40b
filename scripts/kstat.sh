#!/bin/bash
# per-kernel average durations of one serial bench run under rocprofv3: scripts/kstat.sh <tag> [ENV=..] ; prints selected kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ks_$tag
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --inflight 1 --no-events --min-seconds 0 $BENCH_ARGS > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/ks_$tag/*/*kernel_stats.csv')[0]
print('$tag', ' '.join(f"{r['Name'].split('(')[0].replace('void ','')}={float(r['AverageNs'])/1e3:.1f}" for r in csv.DictReader(open(f)) if r['Name'].startswith(('k_','void k_'))))
PY
