#!/usr/bin/env python3
"""Round 6: randomised comparison of the HIP path with the CPU oracle, far outside the shapes the suite's fixtures have.
Per case (a seed): two clouds of 1 ... 60 000 points out of a random scene — walls and floors, gaussian blobs denser than a voxel, thin lines, a uniform box, exact
duplicates, a share of NaN / inf points, a random extent of 1 ... 80 m —, the source a perturbed copy, another view of the scene, or unrelated; random parameters (one
to three levels of 0.05 ... 1 m, 1 ... 30 iterations per level, both metrics, d_max 1 ... 4 leaves, convergence thresholds on or off, a random initial pose).
Compared: error codes of the clouds; the pose (all 16 floats), status, iterations, n_corr and rms of the registration, bit for bit; for target clouds the exported sort
(keys, permutation, points, normals) against the oracle's and the tile images against the cloud itself (mandala_mapping_amd/diag.py).
usage: [M3DREG_LIB=.../libm3dreg_jitter.so] python scripts/r6_fuzz.py [seconds=300] [first_seed=0]      exit code 1 on any difference."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mandala_mapping_amd import abi, binding, diag, synth   # noqa: E402
from oracle import orc   # noqa: E402  (test infrastructure: this script is a test)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
orc.build()


def scene(rng, n, extent):
    """n points of a random scene inside +-extent"""
    parts, left = [], n
    kinds = rng.choice(["wall", "floor", "blob", "line", "box", "dup"], size=rng.integers(1, 6))
    for i, kind in enumerate(kinds):
        m = left if i == len(kinds) - 1 else int(rng.integers(0, left + 1))
        left -= m
        if m == 0:
            continue
        c = rng.uniform(-extent, extent, 3)
        if kind == "wall":
            p = np.stack([rng.uniform(-extent, extent, m), np.full(m, c[1]) + rng.normal(0, 0.005, m), rng.uniform(-2, 3, m)], 1)
            if rng.random() < 0.5:
                p = p[:, [1, 0, 2]]
        elif kind == "floor":
            p = np.stack([rng.uniform(-extent, extent, m), rng.uniform(-extent, extent, m), np.full(m, c[2] * 0.05) + rng.normal(0, 0.004, m)], 1)
        elif kind == "blob":   # far more points than a voxel holds: crowded voxels, big buckets, multi-image / oversize tiles
            # (the tightest blobs are capped: the ORACLE compares every query with every point of its 27 voxels, a 60 000-point blob is minutes per registration)
            sg = float(rng.choice([0.01, 0.05, 0.3]))
            p = c + rng.normal(0, sg, (m, 3))
            if sg < 0.3 and m > 4000:
                p[4000:] = c + rng.normal(0, 0.3 * extent, (m - 4000, 3))
        elif kind == "line":
            t = rng.uniform(0, 1, m)[:, None]
            p = c + t * rng.uniform(-extent, extent, 3) + rng.normal(0, 0.002, (m, 3))
        elif kind == "box":
            p = rng.uniform(-extent, extent, (m, 3))
        else:   # exact duplicates of a few points
            base = rng.uniform(-extent, extent, (max(1, m // 50), 3))
            p = base[rng.integers(0, len(base), m)]
        parts.append(p)
    return np.concatenate(parts).astype(np.float32) if parts else np.zeros((0, 3), np.float32)


def case(seed):
    rng = np.random.default_rng(seed)
    extent = float(rng.choice([1.0, 5.0, 20.0, 80.0]))
    n_t = int(rng.choice([1, 7, 300, 4000, 20000, 60000]) * rng.uniform(0.5, 1.0)) + 1
    tgt = scene(rng, n_t, extent)
    mode = rng.choice(["copy", "view", "other"], p=[0.6, 0.25, 0.15])
    Tgt = synth.small_pose(rng, deg=float(rng.uniform(0, 3)), trans=float(rng.uniform(0, 0.15))) if hasattr(synth, "small_pose") else np.eye(4)
    if mode == "copy":
        idx = rng.permutation(len(tgt))[: max(1, int(len(tgt) * rng.uniform(0.3, 1.0)))]
        src = (tgt[idx].astype(np.float64) @ Tgt[:3, :3].T + Tgt[:3, 3] + rng.normal(0, 0.003, (len(idx), 3))).astype(np.float32)
    elif mode == "view":
        src = scene(np.random.default_rng(seed + 7), int(rng.integers(1, 30000)), extent)
    else:
        src = scene(rng, int(rng.integers(1, 5000)), extent * 0.5)
    for a in (src, tgt):   # non-finite points
        if rng.random() < 0.3 and len(a) > 3:
            k = rng.integers(0, len(a), max(1, len(a) // 40))
            a[k, rng.integers(0, 3, len(k))] = rng.choice([np.nan, np.inf, -np.inf], len(k))
    if rng.random() < 0.03:
        tgt[:] = np.nan   # an empty cloud
    levels = int(rng.choice([1, 1, 1, 2, 3]))
    fine = float(rng.choice([0.05, 0.1, 0.2, 0.5, 1.0]))
    leaf = tuple(fine * 2 ** (levels - 1 - l) for l in range(levels))
    its = tuple(int(rng.integers(1, 31)) for _ in range(levels))
    dmx = tuple(l * float(rng.uniform(1.0, 4.0)) for l in leaf)
    metric = int(rng.choice([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
    eps = float(rng.choice([0.0, 1e-5]))
    p = abi.Params.make(leaf=leaf, iterations=its, max_corr_dist=dmx, metric=metric, normal_leaf=leaf[-1] * float(rng.choice([2.0, 4.0])), eps_rot=eps, eps_trans=eps,
                        min_correspondences=int(rng.choice([3, 10, 200])))
    init = None if rng.random() < 0.5 else np.linalg.inv(Tgt)
    return p, src, tgt, init, f"seed {seed}: {mode}, src {len(src)}, tgt {len(tgt)}, extent {extent}, leaf {leaf}, its {its}, metric {metric}, eps {eps}"


def small_pose(rng, deg, trans):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax) + 1e-12
    a = np.deg2rad(deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    T[:3, 3] = rng.normal(size=3) * trans
    return T


if not hasattr(synth, "small_pose"):
    synth.small_pose = small_pose


def outcome_hip(R, p, src, tgt, init):
    try:
        cs, ct = R.clouds([src, tgt], source_only=[True, False])
    except abi.M3dregError as e:
        return ("cloud_error", e.code), None
    try:
        T, st = R.align(cs, ct, init)
        out = ("ok", np.asarray(T, np.float64).tobytes(), st.status, st.iterations, st.n_corr, float(st.rms))
    except abi.M3dregError as e:
        out = ("align_error", e.code)
    return out, (cs, ct)


def outcome_orc(p, src, tgt, init):
    try:
        os_, ot = orc.Cloud(p, src, source_only=True), orc.Cloud(p, tgt)
    except abi.M3dregError as e:
        return ("cloud_error", e.code), None
    try:
        T, st, _ = orc.align(p, os_, ot, init)
        out = ("ok", np.asarray(T, np.float64).tobytes(), st.status, st.iterations, st.n_corr, float(st.rms))
    except abi.M3dregError as e:
        out = ("align_error", e.code)
    return out, (os_, ot)


def batch_case(seed):
    """FUZZ_BATCH=1: one m3dreg_align_batch of 2 ... 12 pairs of very different sizes (one parameter set), pipelined like bench.py (clouds created without a wait, the
    asynchronous call), every pair against the oracle"""
    rng = np.random.default_rng(seed)
    p, _, _, _, what = case(seed)
    B = int(rng.integers(2, 13))
    pairs = []
    for b in range(B):
        _, src, tgt, init, _ = case(seed * 131 + b + 1)
        pairs.append((src, tgt, init))
    return p, pairs, f"seed {seed}: batch of {B} pairs, sizes {[(len(s_), len(t_)) for s_, t_, _ in pairs]}, {what.split(', leaf')[1] if ', leaf' in what else what}"


def run_batch(R, p, pairs):
    arrays, so = [], []
    for s_, t_, _ in pairs:
        arrays += [s_, t_]; so += [True, False]
    try:
        cl = R.clouds(arrays, wait=False, source_only=so)
    except abi.M3dregError as e:
        return [("cloud_error", e.code)] * len(pairs), None
    T, st = R.align_batch([(cl[2 * j], cl[2 * j + 1], pairs[j][2]) for j in range(len(pairs))])
    out = [("ok", np.asarray(T[j], np.float64).tobytes(), st[j].status, st[j].iterations, st[j].n_corr, float(st[j].rms)) for j in range(len(pairs))]
    return out, cl


t0, n_cases, bad, regs = time.time(), 0, 0, {}
seed = seed0
stats = {"ok": 0, "cloud_error": 0, "align_error": 0}
BATCH = os.environ.get("FUZZ_BATCH", "0") == "1"
while BATCH and time.time() - t0 < budget:
    p, pairs, what = batch_case(seed)
    key = bytes(p)
    R = regs.get(key)
    if R is None:
        if len(regs) > 12:
            for r in regs.values():
                r.close()
            regs.clear()
        R = regs[key] = binding.Registrar(p)
    hs, cl = run_batch(R, p, pairs)
    msgs = []
    for j, (src, tgt, init) in enumerate(pairs):
        o, _ = outcome_orc(p, src, tgt, init)
        h = hs[j]
        if o[0] == "cloud_error":   # (a bad cloud in a batch created without a wait: the pair comes back M3DREG_BAD_CLOUD with the initial pose)
            if not (h[0] == "ok" and h[2] == abi.BAD_CLOUD) and h[0] != "cloud_error":
                msgs.append(f"pair {j}: the oracle refuses a cloud ({o[1]}), HIP returns {h[0], h[2:] if h[0] == 'ok' else h[1]}")
        elif h != o:
            msgs.append(f"pair {j}: HIP {h[0], h[2:] if h[0] == 'ok' else h[1]} oracle {o[0], o[2:] if o[0] == 'ok' else o[1]}" +
                        (f", max |dT| {np.abs(np.frombuffer(h[1], np.float64) - np.frombuffer(o[1], np.float64)).max():.3e}" if h[0] == o[0] == "ok" else ""))
    if cl is not None:
        for c in cl:
            c.free()
    stats[hs[0][0]] += 1
    n_cases += 1
    if msgs:
        bad += 1
        print(f"DIFF {what}\n     " + "\n     ".join(msgs), flush=True)
    if n_cases % 20 == 0:
        print(f"{n_cases} batches, {bad} with differences, {time.time() - t0:.0f} s", flush=True)
    seed += 1
while not BATCH and time.time() - t0 < budget:
    t_case = time.time()
    p, src, tgt, init, what = case(seed)
    key = bytes(p)
    R = regs.get(key)
    if R is None:
        if len(regs) > 24:   # (handles are cheap, their pools are not)
            for r in regs.values():
                r.close()
            regs.clear()
        R = regs[key] = binding.Registrar(p)
    h, hc = outcome_hip(R, p, src, tgt, init)
    o, oc = outcome_orc(p, src, tgt, init)
    msgs = []
    if h != o:
        if h[0] == "ok" and o[0] == "ok":
            Th, To = np.frombuffer(h[1], np.float64), np.frombuffer(o[1], np.float64)
            msgs.append(f"registration differs: max |dT| {np.abs(Th - To).max():.3e}, HIP {h[2:]} oracle {o[2:]}")
        else:
            msgs.append(f"outcome differs: HIP {h[0], h[1] if h[0] != 'ok' else ''} oracle {o[0], o[1] if o[0] != 'ok' else ''}")
    if hc is not None and oc is not None:
        for lvl in range(p.n_levels):
            eh, eo = hc[1].export(lvl), oc[1].export(lvl)
            for k in ("keys", "sorted_keys", "perm", "sorted_xyz", "normals"):
                if eh[k] is None or eo[k] is None:
                    continue
                if not np.array_equal(eh[k].view(np.uint32) if eh[k].dtype == np.float32 else eh[k], eo[k].view(np.uint32) if eo[k].dtype == np.float32 else eo[k]):
                    msgs.append(f"target level {lvl}: {k} differs from the oracle's")
        try:
            pr = diag.tile_image_problems(hc[1], level=p.n_levels - 1)
            msgs += [f"tile image: {m}" for m in pr[:3]]
        except abi.M3dregError as e:
            if e.code != abi.ERR_LEVEL_MISMATCH:   # (a handle without tiles)
                msgs.append(f"tile image check: {e}")
    if hc is not None:
        for c in hc:
            c.free()
    stats[h[0]] += 1
    n_cases += 1
    if msgs:
        bad += 1
        print(f"DIFF {what}\n     " + "\n     ".join(msgs), flush=True)
    if time.time() - t_case > 8.0:
        print(f"(slow case, {time.time() - t_case:.0f} s: {what})", flush=True)
    if n_cases % 50 == 0:
        print(f"{n_cases} cases, {bad} with differences, {stats}, {time.time() - t0:.0f} s", flush=True)
    seed += 1
chk = None
try:
    chk = next(iter(regs.values())).checks()
except Exception:   # noqa: BLE001
    pass
print(f"done: seeds {seed0}..{seed - 1}: {n_cases} cases, {bad} with differences, outcomes {stats}" + (f", index checks {chk}" if chk else ""), flush=True)
raise SystemExit(1 if bad or (chk and (chk['icp'][0] or chk['bucket'][0])) else 0)
