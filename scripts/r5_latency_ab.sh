#!/bin/bash
# round 5: same-box A/B of two libraries on the LATENCY legs (config 3 / 2 / 5, serial steps) and the headline: scripts/r5_latency_ab.sh <other .so relative to the repo> [tests]
R=${GRAFT_REPO_ROOT:-/root/repo}
other=$R/$1
cd $R
if [ "$2" == "tests" ]; then python -m pytest tests -m gpu -x -q > gpurun_out/r5_lat_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_lat_tests.log; tail -3 gpurun_out/r5_lat_tests.log; fi
for rep in 1 2 3; do for lib in "" "$other"; do
  for w in config3 config2; do
    M3DREG_LIB=$lib python bench.py --workload $w --steps 60 --warmup 5 --inflight 1 --queue-depth 1 --no-cpu-baseline --no-extra --min-seconds 0.5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-in-tree} $w', round(d['ms_per_step'],4))"
  done
  M3DREG_LIB=$lib python bench.py --workload config5 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-in-tree} config5', round(d['registration_ms'],4))"
done; done
bash scripts/ab2.sh 2 "M3DREG_LIB=" "M3DREG_LIB=$other"
