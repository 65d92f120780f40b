#!/bin/bash
# PMC counters of ONE kernel per launch (serial steps, no event records), one rocprofv3 --pmc pass per counter group:
#   scripts/pmck.sh <kernel name substring> "<group 1 counters>" ["<group 2 counters>" ...]
# prints the values of the last step's 20 launches at selected Gauss-Newton iterations.
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pk_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pk_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --queue-depth 1 --no-events $BENCH_ARGS > $R/gpurun_out/pk_$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
R="$R"; K="$K"
res=collections.OrderedDict()
for d in sorted(glob.glob(R+'/gpurun_out/pk_*')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        per=collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if K not in r['Kernel_Name']: continue
            per[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
        ids=sorted(per)[-20:]
        if not ids: continue
        for c in per[ids[0]]:
            res[c]=[per[i].get(c,0.0) for i in ids]
sel=(0,1,2,4,6,9,12,19)
print((K+' counter').ljust(34),' '.join(f'it{k:<8d}' for k in sel))
for c,v in res.items():
    print(c.ljust(34),' '.join(f'{v[k]:<10.4g}' for k in sel if k < len(v)))
PY
