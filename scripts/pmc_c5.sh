#!/bin/bash
# PMC counters of the config-5 kernels: scripts/pmc_c5.sh "CTR1 CTR2 ..." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/p5_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/p5_$i -- python3 $R/bench.py --workload config5 --steps 2 --warmup 1 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
R="$R"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(R+'/gpurun_out/p5_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in ('k_nn_coop','k_nn_iter<false>','k_nn_tiles'):
    if k in acc: print(k, cnt[k], {c: round(v/cnt[k][c]) for c,v in sorted(acc[k].items())})
PY
