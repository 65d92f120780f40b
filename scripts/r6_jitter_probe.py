#!/usr/bin/env python3
"""Round 6: does the jitter build (csrc `make jitter`: single waves pause behind every barrier) expose a missing barrier at once?
One handle, synchronous calls, nothing else on the GPU: the targets of config 4's crowded pairs are bucketed `reps` times and every tile image is checked against
the cloud's own sorted points (mandala_mapping_amd/diag.py); the checked build's report is printed. With the tile builder of rounds 5-6 (s_over re-used)
nearly every crowded tile is damaged in every build; with the fixed one none is.   usage: M3DREG_LIB=.../libm3dreg_jitter.so python scripts/r6_jitter_probe.py [reps=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mandala_mapping_amd import abi, binding, diag, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
R = binding.Registrar(p)
try:
    R.checks(reset=True); checked = True
except abi.M3dregError:
    checked = False
tiles = bad_clouds = clouds = 0
msgs = []
for k in (31, 2, 7, 17):
    src, tgt, _ = synth.config4_pair(k, 1600)
    for r in range(reps):
        cs, ct = R.clouds([src, tgt], source_only=[True, False])
        pr = diag.tile_image_problems(ct, max_report=1000)
        clouds += 1; bad_clouds += bool(pr); tiles += len(pr)
        if pr and len(msgs) < 4:
            msgs.append(f"pair {k} build {r}: {pr[0]}")
        cs.free(); ct.free()
print(f"library {os.environ.get('M3DREG_LIB', 'libm3dreg.so')}: {clouds} target clouds bucketed, {bad_clouds} with damaged tile images, {tiles} problems in all")
for m in msgs:
    print("  ", m)
if checked:
    c = R.checks()
    print("   index checks {offences, site, index, bound}: iteration kernels", list(c["icp"]), "bucketing", list(c["bucket"]))
raise SystemExit(1 if bad_clouds else 0)
