"""Per-iteration kernel timeline of the LAST bench step from a rocprofv3 --kernel-trace csv.
usage: python scripts/iter_profile.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
from collections import defaultdict
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
idx = [i for i, r in enumerate(rows) if nm(r) == "k_nn_iter"]
i0 = idx[-1]
j = i0
while j > 0 and nm(rows[j]) != "k_decode_aabb":
    j -= 1
agg = defaultdict(float)
for r in rows[j:i0]:
    agg[nm(r)] += dur(r)
print("bucketing: busy %.1f us, span %.1f us" % (sum(agg.values()), (int(rows[i0]["Start_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3))
for k, v in sorted(agg.items(), key=lambda x: -x[1]):
    print(f"  {k:40s} {v:8.1f}")
cur, it, t_prev = [], 0, None
for r in rows[i0:]:
    cur.append((nm(r), dur(r)))
    if nm(r) in ("k_solve_update", "k_accumulate_matches"):
        if nm(r) == "k_accumulate_matches" and any(nm(x) == "k_solve_update" for x in rows[i0:]):
            continue
        end = int(r["End_Timestamp"])
        span = (end - t_prev) / 1e3 if t_prev else 0.0
        t_prev = end
        print(f"  it{it:2d} " + " ".join(f"{a[2:9]}={b:.1f}" for a, b in cur) + "  busy=%.1f span=%.1f" % (sum(b for a, b in cur), span))
        cur = []
        it += 1
