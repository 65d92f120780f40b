#!/bin/bash
# the big-batch regime — B pairs per step in ONE launch chain, D chains in flight (B = 64: all of config 4 on one GPU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "8 4 2" "16 2 2" "16 3 2" "32 1 2" "32 2 2" "64 1 1" "64 1 2" "64 2 2"; do
  set -- $cfg
  python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --pairs-per-gpu $1 --inflight $2 --queue-depth $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs/step $1 inflight $2 queue $3:', round(d['value']), 'reg/s', round(d['ms_per_step'],3), 'ms/step', 'iter alone', round(d['roofline']['iteration']['alone']['frac'],4), 'rot', round(d['max_rot_err_deg'],4))"
done
