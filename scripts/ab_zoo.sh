#!/bin/bash
# round 4: same-box interleaved A/B of every schedule the library used to pick by itself, before the ones worth < 2 % were deleted
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['value']), round(d['ms_per_step'],4))"; }
for i in 1 2 3; do
  for v in "A=1" "M3DREG_LATE_SMALL=0" "M3DREG_ACC_FILL=0"; do run "$v" ""; done
  for v in "A=1" "M3DREG_ACC_BPP=64"; do run "$v" "--inflight 1 --queue-depth 1"; done
  for v in "A=1" "M3DREG_TILE_CHUNK=512"; do run "$v" "--workload config3 --inflight 1 --queue-depth 1 --steps 60"; done
done
