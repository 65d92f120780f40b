#!/bin/bash
# round 5: kernel sums of two libraries at B = 64 and B = 8, twice, on one box (no tests, no headline): scripts/r5_kab.sh <other .so relative to the repo> [filter regex]
R=${GRAFT_REPO_ROOT:-/root/repo}
other=$R/$1; F=${2:-.}
cd $R
for rep in 1 2; do for B in 64 8; do
  echo "== in-tree B=$B"; bash scripts/kstats.sh abA $B 3 | grep -E "$F|sum of"
  echo "== $1 B=$B"; M3DREG_LIB=$other bash scripts/kstats.sh abB $B 3 | grep -E "$F|sum of"
done; done
