#!/usr/bin/env python3
"""Phase times of k_icp_late's workgroups in the LAST launch of a registration batch (diagnosis build:
hipcc ... -DM3D_LATE_STAMPS -c icp.hip, linked into build/libm3dreg_ls.so):
  M3DREG_LIB=build/libm3dreg_ls.so python scripts/late_stamps.py [iterations]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
params = abi.Params.make(leaf=0.1, iterations=iters, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
reg = binding.Registrar(params, device=0)
L = binding.lib()
clouds, so = [], []
PAIRS = [int(x) for x in os.environ.get("M3D_PAIRS", "0,1,2,3,4,5,6,7").split(",")]
for i in PAIRS:
    src, tgt, _ = synth.config4_pair(i, 3125)
    clouds += [src, tgt]; so += [True, False]
for rep in range(2):
    cs = reg.clouds(clouds, source_only=so)
    reg.align_batch([(cs[2 * i], cs[2 * i + 1], None) for i in range(len(PAIRS))])
buf = (C.c_ulonglong * (4096 * 8))()
L.m3d_debug_read_late(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(4096, 8).astype(np.int64)
a = a[a[:, 0] > 0]
t = a[:, :6] / 100.0
base = t[:, 0].min()
print(f"iteration {iters - 1}: {len(a)} workgroups; kernel span {t[:, 5].max() - base:.1f} us; start p50 {np.median(t[:, 0]) - base:.1f} max {t[:, 0].max() - base:.1f}; "
      f"duration mean {np.mean(t[:, 5] - t[:, 0]):.1f} p50 {np.median(t[:, 5] - t[:, 0]):.1f} max {np.max(t[:, 5] - t[:, 0]):.1f}")
for i, nm in enumerate(["pose+setup", "stream", "walk", "reduce", "tail"]):
    d = t[:, i + 1] - t[:, i]
    print(f"  {nm:11s} mean {d.mean():6.2f} p50 {np.median(d):6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f} us")
nw = a[:, 6]
print("  worklist entries per workgroup: mean %.1f p50 %d p90 %d max %d; end of stream p50 %.1f max %.1f, end of walk p50 %.1f max %.1f, end of reduce max %.1f (us after the first start)" % (
    nw.mean(), np.median(nw), np.percentile(nw, 90), nw.max(), np.median(t[:, 2]) - base, t[:, 2].max() - base, np.median(t[:, 3]) - base, t[:, 3].max() - base, t[:, 4].max() - base))

if hasattr(L, "m3d_debug_read_tail"):   # pair 0's last workgroup, phase by phase
    tb = (C.c_ulonglong * 8)()
    L.m3d_debug_read_tail(tb)
    tt = np.array(tb[:6], dtype=np.int64) / 100.0
    print("  tail of pair 0's last workgroup: stores drained + barrier %.2f, tickets %.2f, partial loads + LDS sums %.2f, solve %.2f, progress report %.2f us" % tuple(np.diff(tt)))
