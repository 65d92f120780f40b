#!/usr/bin/env python3
"""Phase times of k_tile_build (diagnosis build: hipcc ... -DM3D_TB_STAMPS -c bucket.hip, linked into build/libm3dreg_tb.so):
  M3DREG_LIB=build/libm3dreg_tb.so python scripts/tb_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
reg = binding.Registrar(params, device=0)
L = binding.lib()
PAIRS = [int(x) for x in os.environ.get('M3D_PAIRS', '').split(',') if x] or list(range(8))
clouds = []
for i in PAIRS:
    src, tgt, _ = synth.config4_pair(i, 3125)
    clouds += [src, tgt]
buf = (C.c_ulonglong * (4096 * 10))(); n = C.c_uint(0)
reg.clouds(clouds); reg.synchronize()
L.m3d_debug_read_tb(buf, C.byref(n))
cs = reg.clouds(clouds); reg.synchronize()
L.m3d_debug_read_tb(buf, C.byref(n))
k = min(n.value, 4096)
a = np.array(buf[:], dtype=np.uint64).reshape(4096, 10)[:k].astype(np.int64)
t = a[:, :9] / 100.0
base = t[:, 0].min()
names = ["init+heads", "cand set", "probe", "entries", "cut", "dir+src", "copy", "slots", ]
print(f"{k} workgroups; kernel span {t[:, 8].max() - base:.1f} us; start p50 {np.median(t[:, 0]) - base:.1f} max {t[:, 0].max() - base:.1f}; duration mean {np.mean(t[:, 8] - t[:, 0]):.1f} p50 {np.median(t[:, 8] - t[:, 0]):.1f} max {np.max(t[:, 8] - t[:, 0]):.1f}")
for i, nm in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"  {nm:11s} mean {d.mean():6.2f} p50 {np.median(d):6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f} us")
nimg = a[:, 9] >> 32; ne = a[:, 9] & 0xFFFFFFFF
print("  images per tile:", np.bincount(nimg)[:8], " staged buckets p50", int(np.median(ne)), "max", int(ne.max()))
