#!/usr/bin/env python3
"""scripts/walk_stats.py for config 2 (70 k pair, point-to-point, 0.8 / 0.4 / 0.2 m from identity): what the search does per Gauss-Newton iteration.
Instrumented build only: M3DREG_LIB=build/libm3dreg_stats.so python scripts/walk_stats_c2.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth

p = abi.Params.make(leaf=(0.8, 0.4, 0.2), iterations=(30, 30, 150), max_corr_dist=(2.0, 0.6, 0.2), metric=abi.POINT_TO_POINT, eps_rot=1e-5, eps_trans=1e-5)
src, tgt, Tg = synth.config2()
R = binding.Registrar(p)
L = binding.lib()
buf = (C.c_ulonglong * (64 * 24))()
cs, ct = R.clouds([src, tgt], source_only=[True, False])
R.align(cs, ct)
L.m3d_debug_read_stats(buf, 1)
R.align(cs, ct)
L.m3d_debug_read_stats(buf, 1)
a = np.array(buf[:], dtype=np.uint64).reshape(64, 24).astype(np.float64)
names = ["queries", "certified", "none-cached", "seeded", "full", "blk lane", "blk coop", "probes", "found", "rows", "rows>0", "pruned>0", "batches", "cands", "wave trips", "coop passes"]
print(" it " + " ".join(f"{n:>11s}" for n in names))
for it in range(64):
    if a[it].sum() > 0:
        print(f"{it:3d} " + " ".join(f"{a[it, k]:11.0f}" for k in range(16)))
