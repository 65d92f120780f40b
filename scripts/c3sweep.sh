#!/bin/bash
# config 3 (one registration at a time) for several numbers of reduction-pass workgroups per pair: scripts/c3sweep.sh
for r in 1 2; do for b in 0 98 128 196 256 391; do
M3DREG_ACC_BPP=$b python bench.py --workload config3 --steps 60 --warmup 5 --inflight 1 --queue-depth 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bpp $b', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
