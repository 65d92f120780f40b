#!/usr/bin/env python3
"""Randomised bucketing soak (round 5: the rebuilt pipeline — normals from a hash table, k_post_finalize, k_tiles_normals): random clouds (size, extent, crowded
patches, sparse far field, non-finite points, duplicates), random leaf / normal_leaf / levels / metric, batches of 1 ... 5 clouds with source-only members —
grid geometry, keys, permutation, sorted points and normals of every level must equal the CPU oracle's bit for bit.
usage: python scripts/stress_bucketing.py [n_cases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth
from oracle import orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
orc.build()


def random_cloud():
    n = int(rng.choice([7, 300, 5000, 40000, 120000, 300000], p=[0.05, 0.1, 0.25, 0.35, 0.2, 0.05]))
    ext = float(rng.choice([2.0, 12.0, 40.0]))
    parts = [synth.planes_cloud(max(4, n // 2), int(rng.integers(1, 1 << 30)), sigma=float(rng.choice([0.0, 0.01, 0.05])), size=ext)]
    parts.append(rng.uniform(-ext, ext, (max(1, n // 4), 3)).astype(np.float32))
    c = rng.uniform(-1, 1, 3).astype(np.float32)
    parts.append((rng.normal(0, float(rng.choice([0.005, 0.05])), (max(1, n // 4), 3)) + c).astype(np.float32))          # a crowded patch
    a = np.concatenate(parts).astype(np.float32)
    if rng.random() < 0.3:
        a = np.concatenate([a, a[: len(a) // 10]])       # exact duplicates
    a = a[rng.permutation(len(a))]
    if rng.random() < 0.5:
        a[rng.integers(0, len(a), max(1, len(a) // 200))] = np.nan
    return a


t0, clouds_checked = time.time(), 0
for case in range(n_cases):
    metric = int(rng.integers(0, 2))
    levels = int(rng.choice([1, 1, 2, 3]))
    leaf0 = float(rng.choice([0.1, 0.2, 0.35]))
    leaf = tuple(leaf0 * 2 ** (levels - 1 - l) for l in range(levels)) if levels > 1 else leaf0
    nl = float(rng.choice([0.05, 0.2, 0.4, 0.9, 1.7]))
    p = abi.Params.make(leaf=leaf, iterations=(1,) * levels if levels > 1 else 1, max_corr_dist=(0.5,) * levels if levels > 1 else 0.5, metric=metric, normal_leaf=nl)
    R = binding.Registrar(p)
    k = int(rng.integers(1, 6))
    arrs = [random_cloud() for _ in range(k)]
    so = [bool(rng.random() < 0.3) for _ in range(k)]
    try:
        cl = R.clouds(arrs, source_only=so)
    except abi.M3dregError as e:      # (a grid that needs more than 31 key bits: the whole batch is refused — the oracle must refuse the same cloud)
        bad = 0
        for a in arrs:
            try:
                orc.Cloud(p, a)
            except Exception:
                bad += 1
        assert bad > 0, f"case {case}: the library refused a batch the oracle accepts: {e}"
        print(f"case {case}: batch refused ({e.code}), as by the oracle")
        continue
    for a, c, s in zip(arrs, cl, so):
        oc = orc.Cloud(p, a, source_only=s)
        for l in range(levels):
            if s and l < levels - 1:
                continue      # (a source-only cloud's coarser levels are not built)
            g, go = c.grid_info(l), oc.grid_info(l)
            assert bytes(g) == bytes(go), (case, l, g.as_dict(), go.as_dict())
            e, eo = c.export(l), oc.export(l)
            nv = g.n_valid
            for key in ("keys", "sorted_keys", "perm"):
                assert np.array_equal(e[key], eo[key]), (case, l, key)
            assert np.array_equal(e["sorted_xyz"][:nv].view(np.uint32), eo["sorted_xyz"][:nv].view(np.uint32)), (case, l)
            if g.has_normals:
                assert np.array_equal(e["normals"][:nv].view(np.uint32), eo["normals"][:nv].view(np.uint32)), (case, l, "normals")
        clouds_checked += 1
    print(f"case {case}: {k} clouds ok (metric {metric}, leaf {leaf}, normal_leaf {nl}, sizes {[len(a) for a in arrs]}, source_only {so})", flush=True)
print(f"{clouds_checked} clouds bucketed bit-identically to the oracle in {time.time() - t0:.0f} s")
