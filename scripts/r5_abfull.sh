#!/bin/bash
# round 5: same-box A/B of two LIBRARIES over every kernel: GPU tests of the in-tree one, whole-step kernel sums at B = 64 and B = 8 for both, headline / serial interleaved.
# scripts/r5_abfull.sh <other .so, path relative to the repo> [rounds of ab2]
R=${GRAFT_REPO_ROOT:-/root/repo}
other=$R/$1; N=${2:-3}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r5_abfull_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_abfull_tests.log; tail -3 gpurun_out/r5_abfull_tests.log
for B in 64 8; do
  echo "== in-tree B=$B"; bash scripts/kstats.sh abA $B 3 | grep -v "calls     [0-9] avg      [0-9]\.[0-9] us  per step      0\."
  echo "== $1 B=$B"; M3DREG_LIB=$other bash scripts/kstats.sh abB $B 3 | grep -v "calls     [0-9] avg      [0-9]\.[0-9] us  per step      0\."
done
bash scripts/ab2.sh $N "M3DREG_LIB=" "M3DREG_LIB=$other"
