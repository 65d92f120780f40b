#!/usr/bin/env python3
"""What the search does, per Gauss-Newton iteration (instrumented build, see M3D_STATS in csrc/icp.hip):
  hipcc ... -DM3D_STATS -c icp.hip ; link to build/libm3dreg_stats.so ; M3DREG_LIB=build/libm3dreg_stats.so python scripts/walk_stats.py
One 8-pair batch of the bench workload, 20 fixed iterations; counters are summed over the 8 pairs."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth

B = 8
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
reg = binding.Registrar(params, device=0)
L = binding.lib()
buf = (C.c_ulonglong * (64 * 24))()
pairs = []
PAIRS = [int(x) for x in os.environ.get('M3D_PAIRS', '').split(',') if x] or list(range(8))
B = len(PAIRS)
for i in PAIRS:
    src, tgt, _ = synth.config4_pair(i, 3125)
    cs, ct = reg.clouds([src, tgt])
    pairs.append((cs, ct, None))
L.m3d_debug_read_stats(buf, 1)
reg.align_batch(pairs)
L.m3d_debug_read_stats(buf, 1)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 24).astype(np.float64)
names = ["queries", "certified", "none-cached", "seeded", "full", "blk lane", "blk coop", "probes", "found", "rows", "rows>0", "pruned>0", "batches", "cands", "wave trips", "coop passes"]
print(" it " + " ".join(f"{n:>11s}" for n in names))
for it in range(20):
    print(f"{it:3d} " + " ".join(f"{a[it, k]:11.0f}" for k in range(16)))
s = a[:20].sum(0)
srch = s[3] + s[4]
print(f"gather-loop trips per WAVE-slot vs per lane: {s[14]:.0f} wave trips x 64 lanes = {64*s[14]:.0f} lane slots for {s[12]:.0f} lane batches ({100*s[12]/max(1,64*s[14]):.1f} % of the slots useful)")
w = a[:20]
print("wave-level executions per iteration (it: bucket-loop trips, found-bucket bodies, row-loop trips, valid-row bodies, rows reaching the scan, gather trips):")
for it in (0, 1, 3, 5, 8, 12):
    print(f"  it{it:2d}: {w[it,16]:9.0f} {w[it,17]:9.0f} {w[it,21]:9.0f} {w[it,18]:9.0f} {w[it,19]:9.0f} {w[it,14]:9.0f}")
print(f"per searched query: probes {s[7]/srch:.2f} found {s[8]/srch:.2f} rows {s[9]/srch:.2f} nonempty {s[10]/srch:.2f} pruned-nonempty {s[11]/srch:.2f} batches {s[12]/srch:.2f} candidates {s[13]/srch:.2f}")
