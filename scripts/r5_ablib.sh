#!/bin/bash
# round 5: same-box A/B of two LIBRARIES: GPU tests of the in-tree one, iteration kernels alone (8 pairs) and at B = 64, then headline / serial interleaved.  scripts/r5_ablib.sh <other .so, path relative to the repo>
R=${GRAFT_REPO_ROOT:-/root/repo}
other=$R/$1
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r5_ablib_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_ablib_tests.log; tail -3 gpurun_out/r5_ablib_tests.log
for rep in 1 2; do
  bash scripts/kt.sh - 8; M3DREG_LIB=$other bash scripts/kt.sh - 8
  bash scripts/kt.sh - 64; M3DREG_LIB=$other bash scripts/kt.sh - 64
done 2>&1 | sed "s|^- B|in-tree B|"
bash scripts/ab2.sh 3 "M3DREG_LIB=" "M3DREG_LIB=$other"
