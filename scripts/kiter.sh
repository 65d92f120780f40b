#!/bin/bash
# Kernel durations per Gauss-Newton iteration index (serial steps, no event brackets, rocprofv3 kernel trace):
#   scripts/kiter.sh <tag> [ENV=..]     (BENCH_ARGS="--pair-list ..." selects the batch; M3DREG_FUSE_FROM=0 shows the two-launch chain everywhere)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ki_$tag
env "$@" rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ki_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --inflight 1 --queue-depth 1 --no-events --min-seconds 0 $BENCH_ARGS > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/ki_$tag/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
d=lambda r:(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
its=[]   # one entry per Gauss-Newton iteration: [k_nn_iter, k_nn_tiles, k_accumulate_matches, start, end, k_icp_late]
for r in rows:
    n=r['Kernel_Name']
    if 'k_nn_iter' in n: its.append([d(r),0.0,0.0,int(r['Start_Timestamp']),int(r['End_Timestamp']),0.0])
    elif 'k_icp_late' in n: its.append([0.0,0.0,0.0,int(r['Start_Timestamp']),int(r['Start_Timestamp']),d(r)])   # search + reduction in one launch: no separate stage
    elif its and 'k_nn_tiles' in n: its[-1][1]+=d(r); its[-1][4]=int(r['End_Timestamp'])
    elif its and 'k_accumulate_matches' in n: its[-1][2]+=d(r)
its=its[-100:]
K=20
col=lambda j:[sum(x[j] for x in its[i::K])/len(its[i::K]) for i in range(K)]
nn,nt,ac,fl=col(0),col(1),col(2),col(5)
span=[sum((x[4]-x[3])/1e3 for x in its[i::K])/len(its[i::K]) for i in range(K)]
print('$tag nn_iter/iter :', ' '.join(f'{x:.0f}' for x in nn), '| sum %.0f' % sum(nn))
print('$tag nn_tiles/iter:', ' '.join(f'{x:.0f}' for x in nt), '| sum %.0f' % sum(nt))
print('$tag nn stage/iter:', ' '.join(f'{x:.0f}' for x in span), '| sum %.0f avg %.1f (first launch start to last launch end, gaps included; 0 = the iteration ran as k_icp_late)' % (sum(span), sum(span)/K))
print('$tag accumulate/iter:', ' '.join(f'{x:.0f}' for x in ac), '| sum %.0f' % sum(ac))
print('$tag icp_late/iter:', ' '.join(f'{x:.0f}' for x in fl), '| sum %.0f (search + reduction + solve in one launch)' % sum(fl))
print('$tag whole iteration chain, 20 iterations: %.0f us of kernels' % (sum(nn)+sum(nt)+sum(ac)+sum(fl)))
PY
