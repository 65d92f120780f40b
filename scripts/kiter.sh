#!/bin/bash
# k_nn_iter / k_nn_tiles / k_accumulate_matches duration per Gauss-Newton iteration index (serial steps, rocprofv3 kernel trace):
#   scripts/kiter.sh <tag> [ENV=..]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ki_$tag
env "$@" rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ki_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --inflight 1 --queue-depth 1 --no-events > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/ki_$tag/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
d=lambda r:(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
nn=[d(r) for r in rows if 'k_nn_iter' in r['Kernel_Name']][-100:]
nt=[d(r) for r in rows if 'k_nn_tiles' in r['Kernel_Name']][-100:]
gw=[d(r) for r in rows if 'k_nn_gwalk' in r['Kernel_Name']][-100:]
ac=[d(r) for r in rows if 'k_accumulate_matches' in r['Kernel_Name']][-100:]
per=[sum(nn[i::20])/len(nn[i::20]) for i in range(20)]
print('$tag nn_iter/iter:', ' '.join(f'{x:.0f}' for x in per), '| sum %.0f' % sum(per))
if nt:
    pert=[sum(nt[i::20])/len(nt[i::20]) for i in range(20)]
    print('$tag nn_tiles/iter:', ' '.join(f'{x:.0f}' for x in pert), '| sum %.0f' % sum(pert))
    perg=[sum(gw[i::20])/len(gw[i::20]) for i in range(20)] if gw else [0]*20
    print('$tag nn_gwalk/iter:', ' '.join(f'{x:.0f}' for x in perg), '| sum %.0f' % sum(perg))
    print('$tag nn stage/iter:', ' '.join(f'{x+y+z:.0f}' for x, y, z in zip(per, pert, perg)), '| sum %.0f' % (sum(per) + sum(pert) + sum(perg)))
print('$tag acc sum %.0f' % (sum(ac)/5))
PY
