#!/bin/bash
# round 6: the tile / fused-late boundary again on the final library (round 4 measured 7 / 9 / 10 around the 8 that ships). Headline + serial per setting, interleaved, two rounds.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for r in 1 2; do
  for b in 8 6 7 5 9 12; do
    export M3DREG_TILE_ITERS=$b M3DREG_FUSE_FROM=$b
    h=$(python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --min-seconds 1.0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
    s=$(python bench.py --steps 30 --warmup 3 --inflight 1 --queue-depth 1 --no-extra --no-cpu-baseline --min-seconds 0.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
    echo "round $r boundary $b: headline $h serial $s"
  done
done
