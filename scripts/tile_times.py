#!/usr/bin/env python3
"""When and where every workgroup of k_nn_tiles ran in one Gauss-Newton iteration (instrumented build, -DM3D_BLOCKTIME):
  M3DREG_LIB=build/libm3dreg_bt.so python scripts/tile_times.py [iteration ...]
One 8-pair batch of the bench workload; per iteration: kernel span, duration of tile / global-walk workgroups, staging share."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth

iters = [int(a) for a in sys.argv[1:]] or [0]
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
reg = binding.Registrar(params, device=0)
L = binding.lib()
buf = (C.c_ulonglong * (16384 * 8))()
nbuf = (C.c_ulonglong * (8192 * 8))()
PAIRS = [int(x) for x in os.environ.get('M3D_PAIRS', '').split(',') if x] or list(range(8))
pairs = []
for i in PAIRS:
    src, tgt, _ = synth.config4_pair(i, 3125)
    cs, ct = reg.clouds([src, tgt])
    pairs.append((cs, ct, None))
reg.align_batch(pairs)
for it in iters:
    L.m3d_debug_read_blocks(nbuf, it)
    L.m3d_debug_read_tile_blocks(buf)
    reg.align_batch(pairs)
    L.m3d_debug_read_tile_blocks(buf)
    a = np.array(buf[:], dtype=np.uint64).reshape(16384, 8)
    a = a[a[:, 1] > 0]
    if not len(a):
        print(f"--- iteration {it}: no workgroup of k_nn_tiles had work")
        continue
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0
    d = e - s
    kind, nrec, npts, pair = a[:, 3].astype(int), a[:, 4].astype(int), a[:, 5].astype(int), a[:, 7].astype(int)
    stg = a[:, 6].astype(np.int64) / 100.0   # time between 'image staged' and 'every wave done searching it', summed over the passes
    xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
    print(f"--- iteration {it}: {len(a)} workgroups with work, span {e.max():.1f} us; sum of durations / span = {d.sum() / e.max():.0f} in flight on average")
    for k, name in ((0, 'tile'), (1, 'global walk')):
        m = kind == k
        if not m.any():
            continue
        print(f"    {name:11s}: {m.sum():5d} workgroups, {nrec[m].sum():7d} records; duration mean {d[m].mean():.1f} p50 {np.median(d[m]):.1f} p90 {np.percentile(d[m], 90):.1f} max {d[m].max():.1f} us; start p50 {np.median(s[m]):.1f} max {s[m].max():.1f}; last end {e[m].max():.1f}"
              + (f"; search (slowest wave, summed over passes) mean {stg[m].mean():.1f} max {stg[m].max():.1f} us, staged points mean {npts[m].mean():.0f} max {npts[m].max()}, records mean {nrec[m].mean():.0f} max {nrec[m].max()}" if k == 0 else ""))
    o = np.argsort(-e)[:6]
    print("    last to end: " + ", ".join(f"{'T' if kind[j] == 0 else 'G'} pair {pair[j]} xcc {xcc[j]} {d[j]:.0f}us (start {s[j]:.0f}) rec {nrec[j]} pts {npts[j]} search {stg[j]:.0f}us" for j in o))
    for p in sorted(set(pair)):
        m = pair == p
        print(f"    pair {p}: {m.sum():4d} wgs, records tile {nrec[m & (kind == 0)].sum():6d} gw {nrec[m & (kind == 1)].sum():6d}, xcc {sorted(set(xcc[m]))}, first start {s[m].min():6.1f} last end {e[m].max():6.1f}, busy sum {d[m].sum():7.0f} us")
