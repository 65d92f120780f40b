#!/bin/bash
# interleaved A/B, headline and serial, on one box: scripts/ab2.sh rounds "ENV_A=.." "ENV_B=.." ...
# an older build with another ABI version: add M3DREG_ALLOW_ABI_MISMATCH=1 to its ENV string (the binding refuses it otherwise)
N=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq $N); do
  for v in "$@"; do
    for mode in "" "--inflight 1 --queue-depth 1"; do
      env $v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra $mode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$mode'[:12], round(d['value']), round(d['ms_per_step'],4), 'buck', round(d['ms_bucketing_batch_alone'],3))"
    done
  done
done
