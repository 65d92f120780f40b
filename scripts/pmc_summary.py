#!/usr/bin/env python3
"""Turns rocprofv3 --pmc CSVs (one pass per counter group, collected by scripts/profile_gpu.sh on the GPU
box) into profiles/pmc_summary.json, which bench.py reads for `roofline.traffic`.

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so the guide's correction
(x2) is applied to the read side and stated in the output ("fetch_correction")."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if not k.startswith("k_"):
                continue
            for c, v in cs.items():
                out.setdefault(k, {})[c] = sum(v) / len(v)
                out[k]["launches_" + c] = len(v)
for k, v in out.items():
    if "FETCH_SIZE" in v:
        v["fetch_correction"] = 2.0
        v["hbm_read_bytes_per_launch"] = v["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in v:
        v["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024.0
    if "hbm_read_bytes_per_launch" in v:
        v["hbm_bytes_per_launch"] = v["hbm_read_bytes_per_launch"] + v.get("hbm_write_bytes_per_launch", 0.0)
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
# per Gauss-Newton ITERATION (k_nn_tiles is launched in the first iterations of a level only, k_nn_iter in every one): what bench.py's
# roofline.traffic sums over the two kernels of the correspondence step
it = out.get("k_nn_iter", {}).get("launches_FETCH_SIZE", 0)
for k in ("k_nn_iter", "k_nn_tiles"):
    v = out.get(k, {})
    if "hbm_bytes_per_launch" in v and it:
        v["hbm_bytes_per_iteration"] = v["hbm_bytes_per_launch"] * v.get("launches_FETCH_SIZE", 0) / it
        v["hbm_bytes_per_iteration_uncorrected"] = (v.get("FETCH_SIZE", 0.0) * 1024.0 + v.get("hbm_write_bytes_per_launch", 0.0)) * v.get("launches_FETCH_SIZE", 0) / it
json.dump(out, open(os.path.join("profiles", "pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: out[k][c] for c in out[k] if c in ("hbm_bytes_per_launch", "hbm_bytes_per_iteration", "l2_hit_rate", "FETCH_SIZE", "WRITE_SIZE")} for k in out}, indent=1))
