#!/usr/bin/env python3
"""Round 6 experiment: does the ORDER of the pairs inside a batch matter (their workgroups are dispatched pair after pair)? The eight LPT shards of config 4 under the
headline schedule, each with its pairs ordered heaviest-first, heaviest-last and as tabulated (ascending pair id).  python scripts/r6_pair_order.py"""
import ctypes as C, json, multiprocessing as mp, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mandala_mapping_amd import synth
with mp.get_context("fork").Pool(16) as pool:
    data = pool.starmap(synth.config4_pair, [(k, 3125) for k in range(64)])
import torch
from mandala_mapping_amd import abi, binding, sharding
from mandala_mapping_amd.pointcloud2 import encode_xyz
dev = torch.device("cuda", 0)
pay = []
for src, tgt, _ in data:
    ms, mt = encode_xyz(src), encode_xyz(tgt)
    pay.append((torch.frombuffer(bytearray(ms.data), dtype=torch.uint8).to(dev), ms.n, torch.frombuffer(bytearray(mt.data), dtype=torch.uint8).to(dev), mt.n))
torch.cuda.synchronize()
tab = json.load(open(os.path.join(ROOT, "mandala_mapping_amd", "config4_costs.json")))
w = sharding.table_costs(tab, 64)[0]
shards = sharding.lpt_assign(w, 8, capacity=8)
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
regs = [binding.Registrar(params, device=0, stream=C.c_void_p(streams[j % 4].cuda_stream)) for j in range(8)]
for r in regs:
    r.profile_enable(True, every=7)

def run(ids, steps):
    items = []
    for k in ids:
        ds, ns, dt, nt = pay[k]; items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
    B = len(ids)
    def enq(i):
        r = regs[i % 8]
        cl = r.clouds_from_device(items, wait=False, source_only=[True, False] * B)
        r.align_batch_async(r._pairs([(cl[2 * j], cl[2 * j + 1], None) for j in range(B)]), B)
        return cl
    pending, nxt = [], 0
    while nxt < min(8, steps):
        pending.append((nxt, enq(nxt))); nxt += 1
    for i in range(steps):
        idx, cl = pending.pop(0)
        regs[idx % 8].batch_wait(B)
        for c in cl: c.free()
        if nxt < steps:
            pending.append((nxt, enq(nxt))); nxt += 1
    torch.cuda.synchronize()

def ms(ids, steps=24, reps=5):
    run(ids, 8); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(ids, steps); ts.append((time.perf_counter() - t0) / steps)
    return 1e3 * sorted(ts)[len(ts) // 2]

import gc; gc.disable()
for si, sh in enumerate(shards):
    first = sorted(sh, key=lambda k: -w[k]); last = first[::-1]
    a, b, c = ms(sorted(sh)), ms(first), ms(last)
    print(f"shard {si} (heaviest pair cost {max(w[k] for k in sh):.3f}): as tabulated {a:.4f} ms, heaviest first {b:.4f}, heaviest last {c:.4f}", flush=True)
