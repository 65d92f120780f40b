#!/usr/bin/env python3
"""Round 6: hunting a rare nondeterminism of the pipelined schedule (tests/test_gpu_pipelined.py caught 1 differing step in 240 once).
python scripts/r6_hunt.py [steps=60000] [shard=r5_shard6|r5_shard0|pairs0] [every=7] [inflight=4] [queue=2]
Every step's poses + statistics are compared with the single-handle result; every mismatch is printed in detail."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mandala_mapping_amd import abi, binding, synth
from mandala_mapping_amd.pointcloud2 import encode_xyz
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
which = sys.argv[2] if len(sys.argv) > 2 else "r5_shard6"
every = int(sys.argv[3]) if len(sys.argv) > 3 else 7
D = int(sys.argv[4]) if len(sys.argv) > 4 else 4
Q = int(sys.argv[5]) if len(sys.argv) > 5 else 2
shard = {"r5_shard6": [8, 10, 15, 22, 29, 41, 51, 52], "r5_shard0": [2, 19, 20, 24, 31, 35, 50, 59], "pairs0": list(range(8))}[which]
dev = torch.device("cuda", 0)
data = [synth.config4_pair(k) for k in shard]
pay = []
for src, tgt, _ in data:
    ms, mt = encode_xyz(src), encode_xyz(tgt)
    pay.append((torch.frombuffer(bytearray(ms.data), dtype=torch.uint8).to(dev), ms.n, torch.frombuffer(bytearray(mt.data), dtype=torch.uint8).to(dev), mt.n))
torch.cuda.synchronize()
p = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
items = []
for ds, ns, dt, nt in pay:
    items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
R = binding.Registrar(p)
cl = R.clouds_from_device(items, source_only=[True, False] * 8)
Tref, stref = R.align_batch([(cl[2 * j], cl[2 * j + 1], None) for j in range(8)])
for c in cl:
    c.free()
CYCLE = int(os.environ.get("HUNT_CYCLE", "0"))   # > 0: fresh handles every CYCLE steps (the early phase of a handle's life — pool growth, workspace allocation — over and over)
streams = [torch.cuda.Stream(device=dev) for _ in range(D)]
regs = []


def fresh_handles():
    global regs
    for r in regs:
        r.close()
    regs = [binding.Registrar(p, device=0, stream=C.c_void_p(streams[j % D].cuda_stream)) for j in range(D * Q)]
    for r in regs:
        r.profile_enable(every > 0, every=max(1, every))


fresh_handles()
if CYCLE:
    total, steps = steps, CYCLE

def enq(i):
    r = regs[i % len(regs)]
    c = r.clouds_from_device(items, wait=False, source_only=[True, False] * 8)
    r.align_batch_async(r._pairs([(c[2 * j], c[2 * j + 1], None) for j in range(8)]), 8)
    return c

t0 = time.time()
bad = 0
cyc = 0
while True:
  pending, nxt = [], 0
  if True:
    while nxt < min(len(regs), steps):
        pending.append((nxt, enq(nxt))); nxt += 1
    for i in range(steps):
        idx, c = pending.pop(0)
        T, st = regs[idx % len(regs)].batch_wait(8)
        if not np.array_equal(T, Tref) or any((a.status, a.iterations, a.n_corr, a.rms) != (b.status, b.iterations, b.n_corr, b.rms) for a, b in zip(st, stref)):
            bad += 1
            for j in range(8):
                if not np.array_equal(T[j], Tref[j]) or st[j].n_corr != stref[j].n_corr:
                    print(f"MISMATCH step {idx} handle {idx % len(regs)} pair {j} (config-4 pair {shard[j]}): max |dT| {np.abs(T[j] - Tref[j]).max():.3e}  stats {st[j].as_dict()}  ref {stref[j].as_dict()}", flush=True)
        for x in c:
            x.free()
        if nxt < steps:
            pending.append((nxt, enq(nxt))); nxt += 1
        if (i + 1) % 20000 == 0:
            print(f"{i + 1} steps, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
    torch.cuda.synchronize()
    cyc += 1
    if not CYCLE or cyc * CYCLE >= total:
        break
    fresh_handles()
    if cyc % 50 == 0:
        print(f"cycle {cyc}: {cyc * CYCLE} steps, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"done: {steps if not CYCLE else cyc * CYCLE} steps, {bad} mismatching steps ({which}, brackets every {every}, {D} x {Q})")
