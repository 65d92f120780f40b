#!/usr/bin/env python3
"""Time of single registrations of chosen bench pairs (20 fixed iterations): python scripts/pair_time.py 0 17 29 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
reg = binding.Registrar(params, device=0)
for k in [int(a) for a in sys.argv[1:]] or [0]:
    src, tgt, Tgt = synth.config4_pair(k, 3125)
    v = np.floor((tgt - tgt.min(0)) / np.float32(0.1)).astype(np.int64)
    _, cnt = np.unique(v[:, 0] + 4096 * (v[:, 1] + 4096 * v[:, 2]), return_counts=True)
    t0 = time.perf_counter(); cs, ct = reg.clouds([src, tgt]); reg.align(cs, ct); 
    tb = []
    for _ in range(3):
        t0 = time.perf_counter(); cs, ct = reg.clouds([src, tgt]); t1 = time.perf_counter(); T, st = reg.align(cs, ct); tb.append((t1 - t0, time.perf_counter() - t1))
    b, a = min(x[0] for x in tb), min(x[1] for x in tb)
    rot, tra = synth.pose_error(T, Tgt)
    print(f"pair {k}: max {cnt.max()} pts/voxel, {int(cnt[cnt > 32].sum())} pts in voxels > 32; bucketing {1e3 * b:.2f} ms, 20 iterations {1e3 * a:.2f} ms; error {rot:.3f} deg {1e3 * tra:.1f} mm")
