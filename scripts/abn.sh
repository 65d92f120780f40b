#!/bin/bash
# N-way A/B on ONE box, interleaved: scripts/abn.sh rounds "ENV_A=.." "ENV_B=.." ...
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_icp_iter_batch'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
