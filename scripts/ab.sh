#!/bin/bash
# A/B on ONE box, interleaved: scripts/ab.sh "ENV_A=.." "ENV_B=.." [rounds]
A="$1"; B="$2"; N=${3:-3}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_icp_iter_batch'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
