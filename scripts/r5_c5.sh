#!/bin/bash
# round 5: config 5, config 2 (the other pyramid) and config 3 (one pair per step) for two libraries on one box, interleaved: scripts/r5_c5.sh <other .so relative to the repo> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
other=$R/$1; N=${2:-3}
cd $R
for i in $(seq $N); do
  for v in "" "$other"; do
    for w in config5 config2 config3; do
      extra="--steps 10 --warmup 2"; [ $w = config2 ] && extra="--steps 60 --warmup 5 --inflight 1 --queue-depth 1"; [ $w = config3 ] && extra="--steps 200 --warmup 10 --inflight 1 --queue-depth 1"
      M3DREG_LIB=$v python bench.py --workload $w $extra --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('${v:-in-tree}'.split('/')[-1], '$w', 'ms', round(d.get('registration_ms', d['ms_per_step']),4), [round(l['ms_per_icp_iter'],4) for l in d.get('levels',[])], 'map', round(d.get('bucket_map_ms',0),3))"
    done
  done
done
