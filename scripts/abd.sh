#!/bin/bash
# interleaved A/B with the DRIVER's command (20-step blocks repeated for 1.5 s, default event brackets): scripts/abd.sh rounds "ENV_A=.." "ENV_B=.." ...
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    env $v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],4), d['blocks']['n'])"
  done
done
