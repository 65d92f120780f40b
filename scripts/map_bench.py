#!/usr/bin/env python3
"""SURVEY.md §8 row f4 measurement: build the config-5 style map on the device (20 scans of 100k points along a
trajectory, dedup 0.02 m), then time insert and in-place bucketing. Prints one JSON line."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth

p = abi.Params.make(leaf=(0.4, 0.2, 0.1), iterations=(10, 10, 10), max_corr_dist=(1.0, 0.6, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
R = binding.Registrar(p)
m = binding.Map(R, dedup_leaf=0.02, capacity=4 << 20)
scans = [(synth.hdl32_scan(synth.sensor_pose(0.5 * k, 0.05 * k, 1.5 * k), 3125, 200 + k), synth.sensor_pose(0.5 * k, 0.05 * k, 1.5 * k)) for k in range(20)]
clouds = [R.cloud(x) for x, _ in scans]
m.insert(clouds[0], scans[0][1]); m.clear()                      # warm-up
t_ins = []
for c, (_, T) in zip(clouds, scans):
    R.synchronize(); t0 = time.perf_counter()
    added = m.insert(c, T)
    t_ins.append((time.perf_counter() - t0, added))
n = len(m)
R.synchronize(); t0 = time.perf_counter()
tgt = m.as_cloud()
R.synchronize(); t_bucket = time.perf_counter() - t0
live = R.cloud(synth.hdl32_scan(synth.sensor_pose(4.2, 0.4, 12.0), 3125, 999))
T0 = synth.perturb(synth.sensor_pose(4.2, 0.4, 12.0), np.random.default_rng(0), 0.5, 0.05)
R.synchronize(); t0 = time.perf_counter()
T, st = R.align(live, tgt, T0)
t_align = time.perf_counter() - t0
rot, tra = synth.pose_error(T, synth.sensor_pose(4.2, 0.4, 12.0))
print(json.dumps({"map_points": n, "scans": len(scans), "insert_ms_first": 1e3 * t_ins[0][0], "insert_ms_last": 1e3 * t_ins[-1][0],
                  "added_first": t_ins[0][1], "added_last": t_ins[-1][1], "bucket_map_in_place_ms": 1e3 * t_bucket,
                  "scan_to_map_align_ms": 1e3 * t_align, "align_status": st.as_dict()["status"], "rot_err_deg": rot, "trans_err_m": tra}))
