#!/bin/bash
# round 5: the bucketing pipeline's side stream (M3DREG_SIDE_STREAM=0/1): GPU tests with it on, then headline / serial / config 3 / config 5 interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r5_side_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_side_tests.log; tail -3 gpurun_out/r5_side_tests.log
bash scripts/ab2.sh 3 M3DREG_SIDE_STREAM=0 M3DREG_SIDE_STREAM=1
for rep in 1 2; do for v in 0 1; do
  for w in config3 config2; do
    M3DREG_SIDE_STREAM=$v python bench.py --workload $w --steps 60 --warmup 5 --inflight 1 --queue-depth 1 --no-cpu-baseline --no-extra --min-seconds 0.5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SIDE=$v $w', round(d['ms_per_step'],4), 'buck', round(d['ms_bucketing_batch_alone'],4))"
  done
  M3DREG_SIDE_STREAM=$v python bench.py --workload config5 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SIDE=$v config5', round(d['registration_ms'],4), 'map bucketing', round(d['bucket_map_ms'],4))"
done; done
