#!/bin/bash
# Derived rocprofv3 metrics of k_nn_iter per launch (serial steps): scripts/pmc_derived.sh <tag> "<metrics...>"
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pd_$tag
rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $R/gpurun_out/pd_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --queue-depth 1 --no-events > $R/gpurun_out/pd_$tag.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob('$R/gpurun_out/pd_$tag/*/*counter_collection.csv')
if not fs: print(open('$R/gpurun_out/pd_$tag.log').read()[-1500:]); raise SystemExit
per=collections.defaultdict(dict)
for r in csv.DictReader(open(fs[0])):
    if 'k_nn_iter' not in r['Kernel_Name']: continue
    per[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(per)[-20:]
print('$tag'.ljust(28),' '.join(f'it{k:<8d}' for k in (0,1,2,4,6,9,12,19)))
for c in per[ids[0]]:
    v=[per[i].get(c,0.0) for i in ids]
    print(c.ljust(28),' '.join(f'{v[k]:<10.4g}' for k in (0,1,2,4,6,9,12,19)))
PY
