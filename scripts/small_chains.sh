#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "8 4 2" "4 8 2" "4 6 2" "4 4 2" "4 2 1" "4 1 1" "8 1 1" "8 2 1" "2 8 2"; do
  set -- $cfg
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --pairs-per-gpu $1 --inflight $2 --queue-depth $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs/step $1 inflight $2 queue $3:', round(d['value']), 'reg/s', round(d['ms_per_step'],3), 'ms/step')"
done
