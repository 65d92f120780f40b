#!/bin/bash
# round 5: the eight LPT shards `bench.py --gpus 8` forms, each run ALONE on one GPU (headline schedule): what the 8-GPU line will be bounded by — 64 pairs / the slowest shard's step time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 - <<PY > /tmp/shards.txt
import json,sys
sys.path.insert(0,'$R')
from mandala_mapping_amd import sharding
t=json.load(open('$R/mandala_mapping_amd/config4_costs.json'))
for s in sharding.lpt_assign(t['costs'][:64],8,capacity=8): print(','.join(map(str,s)))
PY
r=0
while read -r s; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --pair-list $s --min-seconds 1.0 2>gpurun_out/r5_shard_$r.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('shard $r pairs $s:', round(d['value']), 'registrations/s', round(d['ms_per_step'],4), 'ms per step')"
  r=$((r+1))
done < /tmp/shards.txt
