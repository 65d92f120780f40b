#!/usr/bin/env python3
"""When and where every block of k_nn_iter ran in one Gauss-Newton iteration (instrumented build, -DM3D_BLOCKTIME):
  M3DREG_LIB=build/libm3dreg_bt.so python scripts/block_times.py [iteration ...]
One 8-pair batch of the bench workload; per iteration: kernel span, block duration distribution, per-XCD occupancy over time."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth

iters = [int(a) for a in sys.argv[1:]] or [0]
B = 8
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
reg = binding.Registrar(params, device=0)
L = binding.lib()
buf = (C.c_ulonglong * (8192 * 8))()
pairs = []
PAIRS = [int(x) for x in os.environ.get('M3D_PAIRS', '').split(',') if x] or list(range(8))
B = len(PAIRS)
for i in PAIRS:
    src, tgt, _ = synth.config4_pair(i, 3125)
    cs, ct = reg.clouds([src, tgt])
    pairs.append((cs, ct, None))
reg.align_batch(pairs)   # warm-up (pools, first-touch)
for it in iters:
    L.m3d_debug_read_blocks(buf, it)
    reg.align_batch(pairs)
    L.m3d_debug_read_blocks(buf, it)
    a = np.array(buf[:], dtype=np.uint64).reshape(8192, 8)
    a = a[a[:, 1] > 0]
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0           # microseconds (100 MHz clock)
    d = e - s
    xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
    hw = a[:, 2].astype(np.int64) & 0xFFFFFFFF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
    ns = a[:, 3].astype(np.int64)
    print(f"--- iteration {it}: {len(a)} blocks, span {e.max():.1f} us; block duration mean {d.mean():.1f} p50 {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} p99 {np.percentile(d, 99):.1f} max {d.max():.1f} us; "
          f"sum of durations / span = {d.sum() / e.max():.0f} blocks in flight on average")
    print("    start time of blocks: p50 %.1f p90 %.1f max %.1f us" % (np.median(s), np.percentile(s, 90), s.max()))
    order = np.argsort(-d)[:8]
    tr, mxl, ch, pr = (a[:, k].astype(np.int64) for k in (4, 5, 6, 7))
    print("    slowest blocks: " + ", ".join(f"#{o} {d[o]:.0f}us start {s[o]:.0f} n={ns[o]} trips sum {tr[o]} max/lane {mxl[o]} chunks {ch[o]} probes {pr[o]}" for o in order))
    typ = np.argsort(d)[len(d) // 2 - 3: len(d) // 2 + 3]
    print("    median blocks:  " + ", ".join(f"#{o} {d[o]:.0f}us n={ns[o]} trips sum {tr[o]} max/lane {mxl[o]} chunks {ch[o]} probes {pr[o]}" for o in typ))
    print("    correlation of block duration with: sum of trips %.2f, max trips of a lane %.2f, probes %.2f" % (np.corrcoef(d, tr)[0, 1], np.corrcoef(d, mxl)[0, 1], np.corrcoef(d, pr)[0, 1]))
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"    xcc {x}: {m.sum():4d} blocks, {len(np.unique(cu[m])):3d} CUs, first start {s[m].min():6.1f} last end {e[m].max():6.1f}, mean dur {d[m].mean():5.1f}, in flight avg {d[m].sum() / max(e[m].max() - s[m].min(), 1e-9):5.1f}")
    edges = np.linspace(0, e.max(), 11)
    infl = [int(((s < edges[k + 1]) & (e > edges[k])).sum()) for k in range(10)]
    print("    blocks alive per tenth of the span: " + " ".join(str(v) for v in infl))
