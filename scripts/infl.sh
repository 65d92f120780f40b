#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for d in 3 4 5 6; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --inflight $d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflight $d', round(d['value']), round(d['ms_per_step'],4))"
done; done
