#!/usr/bin/env python3
"""Randomised parity soak: many random registrations (cloud sizes, leaf, metric, levels, batch composition, non-finite points,
initial offsets) — every per-iteration pose of the HIP path must equal the CPU oracle's bit for bit.
usage: python scripts/stress_parity.py [n_cases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth
from oracle import orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
orc.build()
t0 = time.time()
regs = 0
for case in range(n_cases):
    metric = int(rng.integers(0, 2))
    two = bool(rng.integers(0, 2))
    leaf = float(rng.choice([0.1, 0.15, 0.2, 0.3]))
    p = abi.Params.make(leaf=(2 * leaf, leaf) if two else leaf, iterations=(5, 8) if two else int(rng.integers(6, 14)),
                        max_corr_dist=(4 * leaf, 2.5 * leaf) if two else float(rng.choice([2.5, 4.0])) * leaf, metric=metric,
                        normal_leaf=max(0.3, 2 * leaf), eps_rot=float(rng.choice([0.0, 1e-6])), eps_trans=float(rng.choice([0.0, 1e-6])))
    # the schedule switches that are left, pinned at random per case: every combination must give the same bits
    sched = {"M3DREG_LEAN": str(rng.choice(["0", "1"])), "M3DREG_TILES": str(rng.choice(["1", "1", "0"])), "M3DREG_FUSE_FROM": str(rng.choice(["8", "2", "5"]))}
    sched["M3DREG_TILE_ITERS"] = sched["M3DREG_FUSE_FROM"]
    os.environ.update(sched)
    R = binding.Registrar(p)
    lat = bool(rng.integers(0, 2))
    R.set_latency_mode(lat)   # (ABI 7: the serial caller's statement changes launch grids, never a bit)
    k_pairs = int(rng.integers(1, 10))
    pairs, refs = [], []
    for k in range(k_pairs):
        n_az = int(rng.integers(100, 700)) if rng.integers(0, 4) else int(rng.integers(1500, 3200))   # a quarter at full sweep density (crowded voxels near walls)
        src, tgt, Tgt = synth.hdl32_pair(n_az, int(rng.integers(1, 10**6)), int(rng.integers(1, 10**6)), dx=float(rng.uniform(-0.4, 0.4)),
                                         dy=float(rng.uniform(-0.3, 0.3)), dyaw_deg=float(rng.uniform(-4, 4)),
                                         base=(float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3)), float(rng.uniform(-180, 180))))
        if rng.integers(0, 5) == 0:   # a fifth: dense planes (hundreds of points per coarse voxel: the crowded-level kernels, k_nn_coop on a pyramid's coarse level)
            size = float(rng.choice([2.0, 3.0, 5.0]))
            tgt = synth.planes_cloud(int(rng.integers(8000, 50000)), int(rng.integers(1, 10**6)), sigma=0.01, size=size)
            Tgt = synth.make_T(synth.rot_z(np.radians(float(rng.uniform(-2, 2)))) @ synth.rot_x(np.radians(float(rng.uniform(-1, 1)))), rng.uniform(-0.1, 0.1, 3))
            src = synth.apply_T(synth.inv_T(Tgt), synth.planes_cloud(int(rng.integers(3000, 20000)), int(rng.integers(1, 10**6)), sigma=0.01, size=size).astype(np.float64)).astype(np.float32)
        if rng.integers(0, 3) == 0:
            src = src.copy(); src[:: int(rng.integers(17, 90))] = np.nan
        if rng.integers(0, 4) == 0:
            tgt = tgt[: max(50, len(tgt) // int(rng.integers(2, 9)))]
        T0 = synth.perturb(Tgt, rng, 1.0, 0.1) if rng.integers(0, 2) else np.eye(4)
        cs, ct = R.clouds([src, tgt], wait=bool(rng.integers(0, 2)), source_only=[bool(rng.integers(0, 2)), False])   # half of the clouds through the enqueue-only bucketing, half of the sources source_only
        pairs.append((cs, ct, T0))
        refs.append(orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt), T0, trace_cap=64))
    Tb, stb = R.align_batch(pairs)
    for k in range(k_pairs):
        assert np.array_equal(Tb[k], refs[k][0]), (case, k, "batch pose")
        assert (stb[k].status, stb[k].iterations, stb[k].n_corr, stb[k].rms) == (refs[k][1].status, refs[k][1].iterations, refs[k][1].n_corr, refs[k][1].rms), (case, k)
        T1, _ = R.align(*pairs[k])
        assert np.array_equal(R.trace(), refs[k][2]), (case, k, "trace")
        regs += 1
    print(f"case {case}: {k_pairs} pairs ok (metric {metric}, leaf {leaf}, levels {2 if two else 1}; " + " ".join(f"{k[7:].lower()}={v}" for k, v in sched.items()) + ")", flush=True)
print(f"{regs} registrations bit-identical to the oracle in {time.time() - t0:.0f} s")
