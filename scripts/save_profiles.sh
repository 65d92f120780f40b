#!/bin/bash
# Copies what scripts/profile_gpu.sh, scripts/pmc_summary.py, scripts/kiter.sh and a default `python bench.py` run left in gpurun_out/
# into profiles/ under one tag:  scripts/save_profiles.sh r02_v1
set -eu
tag=$1
cd "$(dirname "$0")/.."
# the newest run's file (gpurun merges every call's files into gpurun_out/: older runs stay there), the largest of that run (a bench run has child processes)
big() { python3 -c "import glob,os,sys; f=glob.glob('gpurun_out/'+sys.argv[1]+'/*/*kernel_stats.csv'); t=max(os.path.getmtime(x) for x in f); print(max((x for x in f if os.path.getmtime(x) > t - 120), key=os.path.getsize))" $1; }
cp "$(big stats)" profiles/${tag}_bench_kernel_stats.csv
cp "$(big stats_serial)" profiles/${tag}_serial_kernel_stats.csv
cp gpurun_out/pmc_summary.json profiles/pmc_summary.json
cp gpurun_out/pmc_summary.json profiles/${tag}_pmc_summary.json
[ -f gpurun_out/kiter_${tag//_/}.txt ] && grep -v amdgpu.ids gpurun_out/kiter_${tag//_/}.txt > profiles/${tag}_nn_stage_per_iteration.txt
[ -f gpurun_out/bench_line.json ] && tail -1 gpurun_out/bench_line.json > profiles/${tag}_bench.json
# the full result of THAT run: its stderr carries it (gpurun_out/bench_result_full.json is rewritten by every later bench run)
[ -f gpurun_out/bench_line.err ] && grep "^\[bench full result\] " gpurun_out/bench_line.err | tail -1 | sed 's/^\[bench full result\] //' > profiles/${tag}_bench_full.json
ls -la profiles | grep ${tag}
