#!/bin/bash
# config 2 as the bench leg runs it (one registration at a time), per build: scripts/c2ab.sh build/a.so build/b.so ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  M3DREG_LIB=$lib python bench.py --workload config2 --steps 60 --warmup 5 --inflight 1 --queue-depth 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'config2 ms per registration', round(d['ms_per_step'],3), 'iterations', d.get('iterations_executed_pair0'), 'rot', d.get('max_rot_err_deg'), 'trans', d.get('max_trans_err_m'))"
done
