#!/bin/bash
# Everything a profiles/<tag>_* set is made of, in ONE gpurun call:  gpurun --timeout 1200 -- 'bash scripts/profile_all.sh r03_v4'
# then (in the build container)  scripts/save_profiles.sh r03_v4  copies the summaries into profiles/.
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
t=${tag//_/}
bash scripts/profile_gpu.sh > gpurun_out/profile_gpu.log 2>&1 && echo "profile_gpu done"
python3 scripts/pmc_summary.py gpurun_out > gpurun_out/pmc_summary.log 2>&1 && cp profiles/pmc_summary.json gpurun_out/pmc_summary.json && echo "pmc_summary done"
bash scripts/kiter.sh $t > gpurun_out/kiter_$t.txt 2>&1
bash scripts/kiter.sh ${t}c M3DREG_FUSE_FROM=0 >> gpurun_out/kiter_$t.txt 2>&1 && echo "kiter done"
bash scripts/step_traffic.sh > gpurun_out/step_traffic_$t.txt 2>&1 && echo "step_traffic done"
bash scripts/kstat5.sh $t > gpurun_out/config5_kernel_stats_$t.txt 2>&1 && echo "kstat5 done"
python3 scripts/map_bench.py > gpurun_out/f4_map_bench_$t.json 2>gpurun_out/map_bench.err && echo "map_bench done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_line.json 2>gpurun_out/bench_line.err && echo "bench done"
# the traces themselves stay on the box: gpurun merges at most 64 MiB back
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out -name "*agent_info.csv" -delete
du -sh gpurun_out | tail -1
tail -1 gpurun_out/bench_line.json | cut -c1-400
