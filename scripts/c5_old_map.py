#!/usr/bin/env python3
"""config 5 on the map of rounds 1-3 (20 sweeps de-duplicated at 2 cm: 1.51 M points), for like-for-like comparisons with their numbers"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth
from mandala_mapping_amd.pointcloud2 import encode_xyz
p = abi.Params.make(leaf=(0.4, 0.2, 0.1), iterations=(10, 10, 10), max_corr_dist=(1.0, 0.6, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
live, mp, Tgt, T0 = synth.config5(n_scans=20, dedup=0.02)
R = binding.Registrar(p)
tgt = R.cloud(mp)
msg = encode_xyz(live)
ts = []
for i in range(14):
    R.synchronize(); t0 = time.perf_counter()
    s = R.clouds([msg], source_only=[True])[0]
    T, st = R.align(s, tgt, T0)
    ts.append(1e3 * (time.perf_counter() - t0)); s.free()
print(len(mp), "points; registration ms (median of 10):", sorted(ts[4:])[5], synth.pose_error(T, Tgt))
