#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the default bench, then one PMC pass per counter
# group (never combined with the trace domains gpurun refuses). Results land in gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/stats $R/gpurun_out/stats_serial $R/gpurun_out/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $R/gpurun_out/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_serial -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --inflight 1 --queue-depth 1 --min-seconds 0 > $R/gpurun_out/stats_serial.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  n=$(echo $set | cut -d" " -f1)
  # (M3DREG_FUSE_FROM=0: the two- / three-launch chain in every iteration — the counters are per kernel of the correspondence step, which the fused
  #  late launches do not have)
  M3DREG_FUSE_FROM=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --queue-depth 1 --no-events --min-seconds 0 > /dev/null 2>&1
done
ls $R/gpurun_out
