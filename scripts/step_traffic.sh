#!/bin/bash
# HBM bytes per bench step, all kernels (two rocprofv3 --pmc passes, serial steps): scripts/step_traffic.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/st_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/st_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 --queue-depth 1 --no-events --min-seconds 0 $BENCH_ARGS > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
R="$R"
tot=collections.defaultdict(lambda:[0.0,0.0,0])
for i,c in enumerate(("FETCH_SIZE","WRITE_SIZE")):
    for f in glob.glob(R+'/gpurun_out/st_'+c+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','').split('<')[0]
            tot[k][i]+=float(r['Counter_Value'])*1024.0*(2.0 if i==0 else 1.0)
            if i==0: tot[k][2]+=1
steps=5+1   # 4 timed + 1 warm-up + the untimed 'alone' step
allb=0
for k,(rd,wr,n) in sorted(tot.items(), key=lambda x:-(x[1][0]+x[1][1])):
    print(f"{k[:32]:32s} launches {n:5d}  read {rd/steps/1e6:8.1f} MB/step  write {wr/steps/1e6:8.1f} MB/step")
    allb+=rd+wr
print("total %.0f MB per step (FETCH_SIZE x 2 + WRITE_SIZE)" % (allb/steps/1e6))
PY
