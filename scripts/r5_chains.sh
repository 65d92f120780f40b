#!/bin/bash
# round 5: the synchronous call's internal chains — GPU tests, then serial steps / one 64-pair call per chain count, interleaved.  scripts/r5_chains.sh <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/${tag}_tests.log; tail -5 gpurun_out/${tag}_tests.log
for rep in 1 2; do
for ch in 1 2 3 4; do
  python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --batch-chains $ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial8 chains $ch', round(d['value']), round(d['ms_per_step'],4))"
done
done
for ch in 1 2 4; do
  python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --pairs-per-gpu 64 --batch-chains $ch --min-seconds 0.5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial64 chains $ch', round(d['value']), round(d['ms_per_step'],4))"
done
python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --async-calls 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial8 async one chain', round(d['value']), round(d['ms_per_step'],4))"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['value']), round(d['ms_per_step'],4))"
