#!/usr/bin/env python3
"""A sweep taken close to a wall (tens to hundreds of points per 10 cm voxel): time of one 100k-point registration, 20 fixed
iterations, and the voxel population it meets. usage: [M3DREG_LIB=...] python scripts/crowded_bench.py [distance to the wall, m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth

dist = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
order = sys.argv[2] if len(sys.argv) > 2 else "input"   # input | shuffle | fine (input order = 2.5 cm Morton cells: what a sub-voxel sort key would give)
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
reg = binding.Registrar(params, device=0)
src, tgt, Tgt = synth.hdl32_pair(3125, 500, 501, dx=0.2, dy=0.05, dyaw_deg=2.0, base=(20.0 - dist, 3.0, 10.0))
def reorder(c, how):
    if how == "shuffle":
        return c[np.random.default_rng(1).permutation(len(c))]
    if how == "fine":
        q = np.floor((c - c.min(0)) / np.float32(0.025)).astype(np.int64)
        code = np.zeros(len(c), np.int64)
        for b in range(12):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + a)
        return c[np.argsort(code, kind="stable")]
    return c
src, tgt = reorder(src, order), reorder(tgt, order)
v = np.floor((tgt - tgt.min(0)) / np.float32(0.1)).astype(np.int64)
_, cnt = np.unique(v[:, 0] + 4096 * (v[:, 1] + 4096 * v[:, 2]), return_counts=True)
cs, ct = reg.clouds([src, tgt])
reg.align(cs, ct)
t = []
for _ in range(5):
    t0 = time.perf_counter(); T, st = reg.align(cs, ct); t.append(time.perf_counter() - t0)
rot, tra = synth.pose_error(T, Tgt)
print(f"[{order}] wall at {dist} m: {len(tgt)} pts, max {cnt.max()} pts/voxel, {int((cnt > 32).sum())} voxels > 32 pts holding {int(cnt[cnt > 32].sum())} pts; "
      f"registration {1e3 * min(t):.2f} ms (20 iterations), error {rot:.3f} deg {1e3 * tra:.1f} mm, status {st.status}")
