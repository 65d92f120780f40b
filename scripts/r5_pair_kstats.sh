#!/bin/bash
# round 5: kernel stats of ONE pair of config 4 registered alone (serial steps): scripts/r5_pair_kstats.sh <pair> [<pair> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for k in "$@"; do
  rm -rf $R/gpurun_out/ks_pair$k
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_pair$k -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --min-seconds 0 --no-events --pairs-per-gpu 1 --pair-list $k > /dev/null 2>&1
  python3 - <<PY
import csv,glob,os
f=max(glob.glob('$R/gpurun_out/ks_pair$k/**/*kernel_stats.csv', recursive=True), key=os.path.getsize)
rows=[(r['Name'].split('(')[0].replace('void ',''), int(r['Calls']), float(r['TotalDurationNs'])/1e3/7) for r in csv.DictReader(open(f))]
print('pair $k:', ' | '.join(f"{n[:22]} {t:.0f}" for n,c,t in rows if t > 8), '| sum %.0f us per registration' % sum(t for _,_,t in rows))
PY
done
