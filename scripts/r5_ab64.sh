#!/bin/bash
# round 5: same-box A/B of libraries (and M3DREG_SLICED settings): kernel sums at B = 64 (one chain) and the headline. scripts/r5_ab64.sh "ENV_A" "ENV_B" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for v in "$@"; do
  ( export $v; bash scripts/kstats.sh ab64 64 3 2>&1 | grep -E "sum of|finalize|rs_scatter|tile|nrm_solve|bucket_counts|chunk_boxes|nrm_moments" | awk -v t="$v" '{print t, $0}' | cut -c1-150 )
done
done
bash scripts/ab2.sh 3 "$@"
