#!/bin/bash
# Diagnosis counters of k_nn_iter per launch (serial steps, no event records): one rocprofv3 --pmc pass per group.
# (SQ groups only: a pass with TA_* / TCP_* latency counters did not finish within 7 minutes on this pool and was killed.)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_CYCLES SQ_LEVEL_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/diag_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/diag_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --no-events > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
R="$R"
res=collections.OrderedDict()
for d in sorted(glob.glob(R+'/gpurun_out/diag_*')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        per=collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if 'k_nn_iter' not in r['Kernel_Name']: continue
            per[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
        ids=sorted(per)[-20:]   # the last step's 20 iterations
        for c in per[ids[0]]:
            res[c]=[per[i].get(c,0.0) for i in ids]
print('counter'.ljust(30),' '.join(f'it{k:<8d}' for k in (0,1,2,4,6,9,12,19)))
for c,v in res.items():
    print(c.ljust(30),' '.join(f'{v[k]:<10.3g}' for k in (0,1,2,4,6,9,12,19)))
PY
