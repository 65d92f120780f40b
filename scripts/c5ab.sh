#!/bin/bash
# config 5 and config 2 legs for a list of builds: scripts/c5ab.sh build/a.so build/b.so ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  for w in config5 config2; do
    M3DREG_LIB=$lib python bench.py --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', round(d['ms_per_step'],3), [round(l['ms_per_icp_iter'],4) for l in d.get('levels',[])])"
  done
done
