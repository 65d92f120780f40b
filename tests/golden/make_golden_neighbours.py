#!/usr/bin/env python3
"""Generates tests/golden/golden_neighbours_v1.json: frozen vectors for SURVEY.md §8 rows f2 (calibration cost),
f3 (PointCloud2 layouts) and f4 (persistent map).

The reference holds no vectors for these either (no tests, no fixtures in the tree: SURVEY.md §4), so they are produced
by THIS repo's CPU oracles on seeded synthetic inputs and frozen: they pin the oracles against regressions (CPU suite)
and the HIP paths against the frozen oracles (GPU suite). Inputs are regenerated from seeds by mandala_mapping_amd.synth.

Run from the repo root:  python tests/golden/make_golden_neighbours.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import fnv64  # noqa: E402
from mandala_mapping_amd import synth  # noqa: E402
from mandala_mapping_amd import pointcloud2 as pc2  # noqa: E402

CAL_PARAMS = [(0, 0, 0, 0, 0, 0), (0.0, 0.03, -0.02, 0.03, 0.0, 0.02), (0.0, -0.01, 0.02, 0.01, -0.02, 0.0), (0.01, 0.0, 0.0, 0.0, 0.0, 0.05)]


def cal_segments():
    return synth.calibration_sweep(n_seg=240, n_rays=300, seed=21)


def map_scans():
    out = []
    for k in range(3):
        pose = synth.sensor_pose(0.5 * k, -0.1 * k, 3.0 * k)
        out.append((synth.hdl32_scan(pose, 300, 70 + k), pose))
    return out


def layout_messages():
    F = pc2.PointField
    xyz = synth.planes_cloud(2000, 31)
    return {
        "f64_unaligned_bigendian": pc2.encode_general(xyz, [F("x", 3, pc2.FLOAT64), F("y", 11, pc2.FLOAT64), F("z", 19, pc2.FLOAT64)], 29, big_endian=True),
        "organised_padded": pc2.encode_general(xyz, [F("z", 0), F("intensity", 4), F("y", 8), F("x", 12)], 20, width=40, height=50, row_pad=12),
    }


def main():
    from oracle import orc
    orc.build()
    out = {"version": 1, "generator": "tests/golden/make_golden_neighbours.py (CPU oracles; the reference has no vectors)"}
    c = orc.Calibration(1)
    segs = cal_segments()
    for xyz, T in segs:
        c.add_segment(xyz, T)
    out["calibration"] = {"points": int(sum(len(x) for x, _ in segs)), "input_fnv64": fnv64(np.concatenate([x for x, _ in segs])),
                          "costs": [int(c.test_data(p)[0]) for p in CAL_PARAMS], "voxels": [[int(v) for v in c.test_data(p)[1][2:]] for p in CAL_PARAMS]}
    m = orc.Map(0.05, 200000)
    added = [m.insert(x, T) for x, T in map_scans()]
    out["map"] = {"added": added, "points_fnv64": fnv64(m.points())}
    out["layouts"] = {name: {"xyz_fnv64": fnv64(orc.decode_pc2(msg))} for name, msg in layout_messages().items()}
    print(out["calibration"]["costs"], out["map"]["added"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_neighbours_v1.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
