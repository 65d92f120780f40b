#!/usr/bin/env python3
"""SURVEY.md §8 row f2 measurement: evaluations of the calibration cost function per second — m3dcal_evaluate on the
GPU (many candidates per launch) next to the CPU oracle (one core) on the same synthetic sweep. Prints one JSON line."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import binding, synth
from oracle import orc

n_seg, n_rays, K = 720, 720, 32
segs = synth.calibration_sweep(n_seg=n_seg, n_rays=n_rays, seed=9)
npts = sum(len(x) for x, _ in segs)
R = binding.Registrar()
cal = binding.Calibrator(R, 1)
co = orc.Calibration(1)
orc.build()
for xyz, T in segs:
    cal.add_segment(xyz, T)
    co.add_segment(xyz, T)
rng = np.random.default_rng(0)
params = np.concatenate([np.zeros((K, 1)), rng.uniform(-0.05, 0.05, (K, 5))], axis=1).astype(np.float32)
cal.evaluate(params)                                   # warm-up (allocations, upload)
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    got = cal.evaluate(params)
gpu_s = (time.perf_counter() - t0) / (reps * K)
t0 = time.perf_counter()
got1 = [cal.evaluate(params[i:i + 1])[0] for i in range(8)]
gpu1_s = (time.perf_counter() - t0) / 8
t0 = time.perf_counter()
ref = [co.test_data(params[i])[0] for i in range(4)]
cpu_s = (time.perf_counter() - t0) / 4
assert list(got[:4]) == ref and got1 == list(got[:8])
# algorithmic bytes per evaluation: raw point (16 B) + its 12-float transform is cached; table slots touched: keys 8 + cnt 4 + sums 24 per occupied voxel twice
print(json.dumps({"workload": f"calibration sweep {n_seg} segments x {n_rays} rays = {npts} points, {K} candidates per launch",
                  "gpu_evaluations_per_s_batched": 1.0 / gpu_s, "gpu_ms_per_evaluation_batched": 1e3 * gpu_s,
                  "gpu_ms_per_evaluation_single": 1e3 * gpu1_s, "cpu_oracle_ms_per_evaluation": 1e3 * cpu_s, "cpu_cores": 1,
                  "speedup_batched": cpu_s / gpu_s, "counts_identical": True}))
