#!/bin/bash
# round 5: XCD-sliced rows (M3DREG_SLICED bit 0 = k_finalize_level, 1 = k_rs_scatter, 2 = k_tile_build): kernel averages of serial steps per setting, then headline A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 7; do
  export M3DREG_SLICED=$v
  rm -rf $R/gpurun_out/ks_sl$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_sl$v -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --inflight 1 --queue-depth 1 --min-seconds 0 --no-events > /dev/null 2>&1
  python3 - <<PY
import csv,glob,os
f=max(glob.glob('$R/gpurun_out/ks_sl$v/**/*kernel_stats.csv', recursive=True), key=os.path.getsize)
d={r['Name'].split('(')[0].replace('void ',''): float(r['AverageNs'])/1e3 for r in csv.DictReader(open(f))}
print('SLICED=$v', ' '.join(f"{k}={d.get(k,0):.1f}" for k in ('k_finalize_level','k_rs_scatter','k_tile_build','k_bucket_counts','k_nrm_moments','k_chunk_boxes')))
PY
done
unset M3DREG_SLICED
cd $R
bash scripts/ab2.sh 2 M3DREG_SLICED=0 M3DREG_SLICED=1 M3DREG_SLICED=7
