#!/usr/bin/env python3
"""Generates tests/golden/golden_v<SPEC_VERSION>.json (never overwrites an existing version without --force: a change of the normative
arithmetic gets a NEW file and an entry in its spec_history; older files stay frozen and keep being tested for every field the change
did not touch — tests/test_golden.py).

The reference holds no golden vectors for this path (gpu_6dslam is an empty submodule, the
tree has no tests), so these vectors are produced by THIS repo's CPU oracle (oracle/m3d_oracle.c) on
seeded synthetic inputs and frozen here: they pin the oracle against regressions (CPU suite) and the HIP
path against the frozen oracle (GPU suite). Inputs are regenerated from the seeds by
mandala_mapping_amd.synth, so only checksums and poses are stored.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mandala_mapping_amd import abi, synth  # noqa: E402
from oracle import orc  # noqa: E402


SPEC_VERSION = 2   # golden_v2.json: normals v2 (bea73c6) + solve v2 (a025d4d); golden_v1.json = spec v1, frozen


def fnv64(a):
    """FNV-1a over the little-endian bytes of an array (uint64 hex)."""
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


CASES = [
    # name, generator, params
    ("config1_pt2plane_seed42", lambda: synth.config1(6000),
     dict(leaf=0.25, iterations=12, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)),
    ("config1_pt2pt_seed42", lambda: synth.config1(6000),
     dict(leaf=0.25, iterations=12, max_corr_dist=0.5, metric=abi.POINT_TO_POINT)),
    ("hdl32_small_seed43", lambda: synth.hdl32_pair(500, 43, 143, dx=0.25, dy=0.05, dyaw_deg=2.0),
     dict(leaf=0.2, iterations=15, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)),
    ("hdl32_multires_seed44", lambda: synth.hdl32_pair(400, 44, 144, dx=0.4, dy=-0.1, dyaw_deg=3.0),
     dict(leaf=(0.4, 0.2), iterations=(8, 10), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.5)),
]


def run_case(gen, pk):
    src, tgt, Tgt = gen()
    p = abi.Params.make(**pk)
    cs, ct = orc.Cloud(p, src), orc.Cloud(p, tgt)
    T, st, tr = orc.align(p, cs, ct, trace_cap=64)
    levels = []
    for l in range(p.n_levels):
        g, e = ct.grid_info(l), ct.export(l)
        levels.append({"dims": list(g.dims), "bits": list(g.bits), "n_cells": g.n_cells, "n_valid": g.n_valid,
                       "keys_fnv64": fnv64(e["keys"]), "perm_fnv64": fnv64(e["perm"]),
                       "normals_fnv64": fnv64(e["normals"][: g.n_valid]) if g.has_normals else None})
    return {"n_src": len(src), "n_tgt": len(tgt), "input_fnv64": [fnv64(src), fnv64(tgt)], "target_levels": levels,
            "pose_colmajor_f32": [float(v) for v in np.asarray(T, np.float64).T.reshape(16)],
            "trace_fnv64": fnv64(tr), "status": st.status, "iterations": st.iterations, "n_corr": st.n_corr,
            "rms": st.rms, "pose_error_deg_m": list(synth.pose_error(T, Tgt))}


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"golden_v{SPEC_VERSION}.json")
    if os.path.exists(path) and "--force" not in sys.argv:
        raise SystemExit(f"{path} exists: a spec change gets a new SPEC_VERSION (and a spec_history entry), not a regenerated file")
    out = {"version": SPEC_VERSION, "generator": "tests/golden/make_golden.py (CPU oracle; the reference has no vectors for this path)", "cases": {}}
    if os.path.exists(path):
        out["spec_history"] = json.load(open(path)).get("spec_history", [])
    for name, gen, pk in CASES:
        out["cases"][name] = {"params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in pk.items()}, **run_case(gen, pk)}
        print(name, out["cases"][name]["pose_error_deg_m"], out["cases"][name]["iterations"])
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
