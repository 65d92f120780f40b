#!/bin/bash
# interleaved A/B of whole-bench throughput on ONE box, headline and serial: scripts/abx.sh rounds "ENV_A=.." "ENV_B=.." ...
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    for mode in "" "--inflight 1 --queue-depth 1 --steps 30"; do
      env $v python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-extra $mode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$mode'[:12], round(d['value']), round(d['ms_per_step'],3), 'nn alone ms', round(d['roofline']['avg_launch_ms'],4))"
    done
  done
done
