#!/bin/bash
# round 5: the bench itself as a soak — N runs of the headline over the eight LPT shards of config 4, every run's exit code and line checked, a failing run's stderr kept.  scripts/r5_soak_bench.sh [runs]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-24}
cd $R
python3 - <<PY > /tmp/shards.txt
import json,sys
sys.path.insert(0,'$R')
from mandala_mapping_amd import sharding
t=json.load(open('$R/mandala_mapping_amd/config4_costs.json'))
for s in sharding.lpt_assign(t['costs'][:64],8,capacity=8): print(','.join(map(str,s)))
PY
fail=0
for i in $(seq 1 $N); do
  s=$(sed -n "$(( (i % 8) + 1 ))p" /tmp/shards.txt)
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --pair-list $s --min-seconds 0.5 > gpurun_out/soakb.out 2> gpurun_out/soakb_$i.err; rc=$?
  if [ $rc -ne 0 ] || ! grep -q '"value"' gpurun_out/soakb.out; then echo "FAIL run $i rc=$rc shard $s"; grep -v "bench full result" gpurun_out/soakb_$i.err | tail -20; fail=$((fail+1)); else rm -f gpurun_out/soakb_$i.err; fi
done
echo "bench soak: $N runs, $fail failed"
