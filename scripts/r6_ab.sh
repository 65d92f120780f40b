#!/bin/bash
# round 6: A/B of library builds on one box, interleaved.  scripts/r6_ab.sh name=path.so [name=path.so ...]   (paths relative to the repo root)
# per build and round: the headline (8 handles / 4 streams), serial synchronous steps, one 64-pair call, and the rotate-pairs leg (per-shard steps -> ceiling of the 8-GPU line)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ROUNDS=${ROUNDS:-2}
for r in $(seq 1 $ROUNDS); do
  for kv in "$@"; do
    name=${kv%%=*}; lib=$R/${kv#*=}
    export M3DREG_LIB=$lib M3D_BENCH_FULL_LINE=1
    h=$(python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --min-seconds 1.0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
    s=$(python bench.py --steps 30 --warmup 3 --inflight 1 --queue-depth 1 --no-extra --no-cpu-baseline --min-seconds 0.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
    b=$(python bench.py --pairs-per-gpu 64 --inflight 1 --queue-depth 1 --steps 6 --warmup 2 --no-extra --no-cpu-baseline --min-seconds 0.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
    c=$(python bench.py --rotate-pairs --steps 24 --warmup 8 --no-extra --no-cpu-baseline --min-seconds 0.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=d['scale_ceiling']; print(round(d['value']), 'ceiling', round(sc['n8_over_n1_measured_shards'],3), 'balance', round(sc['balance_mean_over_max'],4), 'shards', [round(x,3) for x in sc['per_shard_ms_per_step']])")
    echo "round $r $name: headline $h serial $s batch64 $b rotate $c"
  done
done
