#!/bin/bash
# A/B/C... on ONE box, interleaved: scripts/abn.sh rounds "ENV_A=.." "ENV_B=.." ...   (each variant = one env assignment list, "X=1" for none)
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_icp_iter_batch'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
