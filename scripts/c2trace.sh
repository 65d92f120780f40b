#!/bin/bash
# config 2: kernel durations per launch, in order, for ONE registration (the last of the run)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c2t
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c2t -- python3 $R/bench.py --workload config2 --steps 6 --warmup 2 --inflight 1 --queue-depth 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/c2t/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
names=[(r['Kernel_Name'].split('(')[0].replace('void ','')[:22], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
# find the registrations of the main loop: sequences starting with k_patch_jobs; take the 3rd
idx=[i for i,(n,_) in enumerate(names) if n.startswith('k_patch_jobs')]
i0=idx[-3]; i1=idx[-2]   # (a registration of the steady state: the handle knows its dense levels)
seq=names[i0:i1]
out=[]; tot={}
for n,d in seq:
    tot[n]=tot.get(n,0)+d
print(' '.join(f"{n.replace('k_','')[:10]}:{d:.0f}" for n,d in seq))
print({k: round(v) for k,v in tot.items()}, 'sum', round(sum(tot.values())))
PY
