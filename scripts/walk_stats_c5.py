#!/usr/bin/env python3
"""scripts/walk_stats.py for config 5 (100 k live scan vs the 2 M-point map, 0.4 / 0.2 / 0.1 m, 10 iterations each): what the search does per
Gauss-Newton iteration. Instrumented build only: M3DREG_LIB=build/libm3dreg_stats.so python scripts/walk_stats_c5.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth

leaves, dmaxs = (0.4, 0.2, 0.1), (1.0, 0.6, 0.5)
p = abi.Params.make(leaf=leaves, iterations=(10, 10, 10), max_corr_dist=dmaxs, metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
live, mp, Tgt, T0 = synth.config5()
R = binding.Registrar(p)
L = binding.lib()
buf = (C.c_ulonglong * (64 * 24))()
tgt = R.cloud(mp)
src = R.clouds([live], source_only=[True])[0]
R.align(src, tgt, T0)
L.m3d_debug_read_stats(buf, 1)
R.align(src, tgt, T0)
L.m3d_debug_read_stats(buf, 1)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 24).astype(np.float64)
names = ["queries", "certified", "none-cached", "seeded", "full", "blk lane", "blk coop", "probes", "found", "rows", "rows>0", "pruned>0", "batches", "cands", "wave trips", "coop passes"]
print(" it " + " ".join(f"{n:>11s}" for n in names))
for it in range(32):
    print(f"{it:3d} " + " ".join(f"{a[it, k]:11.0f}" for k in range(16)))
for it in range(32):
    srch = a[it, 3] + a[it, 4]
    if srch > 0:
        print(f"it{it:2d} per searched query: probes {a[it,7]/srch:.2f} found {a[it,8]/srch:.2f} rows {a[it,9]/srch:.2f} nonempty {a[it,10]/srch:.2f} pruned-nonempty {a[it,11]/srch:.2f} "
              f"gather batches {a[it,12]/srch:.2f} candidates {a[it,13]/srch:.2f}; lane slots useful {100*a[it,12]/max(1,64*a[it,14]):.1f} %")
