"""Loop-closure candidate generation (second half): m3dloop_* (csrc/loop.hip) against oracle/m3d_loop_oracle.c, the frozen
fixture tests/golden/loop_v1.json and an independent numpy restatement — and the candidates feeding the batch path unchanged
(m3dreg_align_batch on the keyframes' resident clouds, m3dreg_multi_align on their payload descriptors)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from mandala_mapping_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "loop_v1.json")))


def _fnv64(a):
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def _record(c):
    return {"source": c.source, "target": c.target, "overlap": c.overlap, "pop_source": c.pop_source, "pop_target": c.pop_target,
            "dist2_bits": f"{np.float32(c.dist2).view(np.uint32):08x}", "init_T_bits": np.asarray(list(c.init_T), np.float32).view(np.uint32).tobytes().hex()}


def _numpy_signature(xyz, T, leaf, log2_bits):
    """independent restatement: float64 emulation of the float fma chain (a product of two floats is exact in double; the sums round once more,
    which can differ from a true fma only within 2^-29 relative of a rounding tie — never on these seeded clouds), Python integers for the hash"""
    T32 = np.asarray(T, np.float64).astype(np.float32)
    p = xyz[np.isfinite(xyz).all(axis=1)].astype(np.float64)
    u = np.empty_like(p, dtype=np.float32)
    for r in range(3):
        acc = np.float64(T32[r, 3])
        acc = (np.float64(T32[r, 2]) * p[:, 2] + acc).astype(np.float32)
        acc = (np.float64(T32[r, 1]) * p[:, 1] + acc.astype(np.float64)).astype(np.float32)
        acc = (np.float64(T32[r, 0]) * p[:, 0] + acc.astype(np.float64)).astype(np.float32)
        u[:, r] = acc
    inv = np.float32(1.0) / np.float32(leaf)
    v = np.floor(u * inv).astype(np.int64)
    bits = set()
    for vx, vy, vz in np.unique(v, axis=0):
        key = ((int(vx) + 1048576) << 42) | ((int(vy) + 1048576) << 21) | (int(vz) + 1048576)
        bits.add(((key * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF) >> (64 - log2_bits))
    w = np.zeros(1 << (log2_bits - 5), np.uint32)
    for b in bits:
        w[b >> 5] |= np.uint32(1 << (b & 31))
    return w


def _numpy_candidates(sigs, poses, P):
    pops = [int(np.unpackbits(w.view(np.uint8)).sum()) for w in sigs]
    thr = int(np.rint(np.float32(P.min_overlap) * np.float32(65536.0)))
    r2 = np.float32(P.radius) * np.float32(P.radius)
    out = []
    for i in range(len(sigs)):
        ti = np.asarray(poses[i], np.float64).astype(np.float32)[:3, 3]
        row = []
        for j in range(0, i - P.min_gap + 1):
            tj = np.asarray(poses[j], np.float64).astype(np.float32)[:3, 3]
            d = ti - tj
            d2 = np.float32(np.float64(d[2]) * np.float64(d[2]) + np.float64(np.float32(np.float64(d[1]) * np.float64(d[1]) + np.float64(d[0] * d[0]))))
            if not d2 <= r2:
                continue
            ov = int(np.unpackbits((sigs[i] & sigs[j]).view(np.uint8)).sum())
            if (ov << 16) >= thr * min(pops[i], pops[j]):
                row.append((-ov, j))
        out += [(i, j, -o) for o, j in sorted(row)[: P.top_k]]
    return out


@pytest.mark.parametrize("case", sorted(FIX["cases"]))
def test_oracle_matches_the_frozen_fixture_and_an_independent_restatement(orc, case):
    fx = FIX["cases"][case]
    tr = synth.loop_trajectory(**fx["spec"]["traj"])
    P = abi.LoopParams.make(**fx["spec"]["loop"])
    L = orc.Loop(P)
    for cloud, _, T_odo in tr:
        L.add_keyframe(cloud, T_odo)
    sigs = [L.signature(k) for k in range(len(tr))]
    assert [p for _, p in sigs] == fx["pops"] and [_fnv64(w) for w, _ in sigs] == fx["signature_fnv64"]
    got = L.candidates()
    assert [_record(c) for c in got] == fx["candidates"]
    for k in (0, len(tr) // 2, len(tr) - 1):
        assert np.array_equal(_numpy_signature(tr[k][0], tr[k][2], P.sig_leaf, P.sig_log2_bits), sigs[k][0]), k
    assert _numpy_candidates([w for w, _ in sigs], [t[2] for t in tr], P) == [(c.source, c.target, c.overlap) for c in got]
    # rows asked one at a time (what a node does with the keyframe it has just added) = the rows of the whole table
    rows = [c for i in range(len(tr)) for c in L.candidates(i, 1)]
    assert [_record(c) for c in rows] == fx["candidates"]
    # init_T is inv(T_target) * T_source of the poses given
    c = got[-1]
    rel = np.linalg.inv(np.asarray(tr[c.target][2], np.float64).astype(np.float32).astype(np.float64)) @ np.asarray(tr[c.source][2], np.float64).astype(np.float32).astype(np.float64)
    assert np.allclose(np.asarray(list(c.init_T)).reshape(4, 4).T, rel, atol=1e-5)


def test_oracle_rules_gap_radius_threshold_and_ties(orc):
    """Hand-made keyframes: identical clouds at chosen positions — overlap ties go to the OLDER keyframe, the gap and the radius rule pairs out, a
    threshold above the overlap ratio empties the row, non-finite points and a full database are handled."""
    rng = np.random.default_rng(3)
    cloud = rng.uniform(-6, 6, (3000, 3)).astype(np.float32)
    cloud[::50] = np.nan
    P = abi.LoopParams.make(sig_leaf=1.0, sig_log2_bits=12, radius=3.0, min_gap=2, top_k=3, min_overlap=0.9, max_keyframes=6)
    L = orc.Loop(P)
    T = lambda x: synth.make_T(np.eye(3), np.array([x, 0.0, 0.0]))
    for x in (0.0, 0.0, 0.0, 100.0, 0.0, 2.0):      # 0, 1, 2, 4 identical; 3 far away; 5 shifted by two voxels
        assert L.add_keyframe(cloud, T(x)) == len(L) - 1
    assert L.add_keyframe(cloud, T(0.0)) == -1       # full
    got = [(c.source, c.target) for c in L.candidates()]
    assert got == [(2, 0), (4, 0), (4, 1), (4, 2)], got          # row 3: nobody within 3 m; row 5: shifted, overlap below 0.9; ties: older first
    P2 = abi.LoopParams.make(sig_leaf=1.0, sig_log2_bits=12, radius=3.0, min_gap=2, top_k=3, min_overlap=0.0, max_keyframes=6)
    L2 = orc.Loop(P2)
    for x in (0.0, 0.0, 0.0, 100.0, 0.0, 2.0):
        L2.add_keyframe(cloud, T(x))
    row5 = [(c.source, c.target) for c in L2.candidates(5, 1)]
    assert row5 == [(5, 0), (5, 1), (5, 2)], row5               # (3 is 98 m away; 4 is inside the gap)


def test_abi_structs_have_the_header_layout():
    assert C.sizeof(abi.LoopParams) == 32 and C.sizeof(abi.LoopCandidate) == 24 + 64


# ---------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(FIX["cases"]))
def test_device_candidates_equal_the_oracle_and_the_fixture(reg, orc, case):
    fx = FIX["cases"][case]
    tr = synth.loop_trajectory(**fx["spec"]["traj"])
    P = abi.LoopParams.make(**fx["spec"]["loop"])
    p = abi.Params.make(leaf=0.2, iterations=12, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    clouds = R.clouds([t[0] for t in tr])
    G, O = reg.LoopCloser(R, P), orc.Loop(P)
    for (cloud, _, T_odo), c in zip(tr, clouds):
        assert G.add_keyframe(c, T_odo) == O.add_keyframe(cloud, T_odo)
    for k in range(len(tr)):
        wg, pg = G.signature(k)
        wo, po = O.signature(k)
        assert np.array_equal(wg, wo) and pg == po == fx["pops"][k], k
    got = [_record(c) for c in G.candidates()]
    assert got == [_record(c) for c in O.candidates()] == fx["candidates"]
    rows = [_record(c) for i in range(len(tr)) for c in G.candidates(i, 1)]     # the node's use: one new row at a time
    assert rows == fx["candidates"]
    assert [_record(c) for c in G.candidates(7, 30)] == [_record(c) for c in O.candidates(7, 30)]
    ms, nbytes = G.last_profile()
    assert ms > 0.0 and nbytes > 0
    # a pose graph moves a keyframe: its signature follows
    k = len(tr) - 3
    T2 = tr[k][2] @ synth.make_T(synth.rot_z(np.radians(4.0)), np.array([0.7, -0.4, 0.05]))
    G.update_pose(k, T2); O.update_pose(k, tr[k][0], T2)
    assert np.array_equal(G.signature(k)[0], O.signature(k)[0])
    assert [_record(c) for c in G.candidates()] == [_record(c) for c in O.candidates()]


@pytest.mark.gpu
def test_candidates_feed_the_batch_path_unchanged(reg, orc):
    """The closed trajectory end to end: candidates -> m3dreg_pair[] -> m3dreg_align_batch -> gate. Every accepted loop closure recovers the TRUE
    relative pose of its two keyframes (the odometry's guess is off by the accumulated drift); the same candidates as m3dreg_pair_desc[] through
    m3dreg_multi_align (target groups: one bucketing per target keyframe) give the same bits; and the registration equals the oracle's."""
    tr = synth.loop_trajectory(n_keyframes=50, per_lap=40, n_azimuth=900, seed=9200, step_noise_deg=0.2, step_noise_m=0.02)
    p = abi.Params.make(leaf=(0.4, 0.15), iterations=(15, 15), max_corr_dist=(1.2, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    P = abi.LoopParams.make(radius=3.0, min_gap=15, top_k=2, min_overlap=0.5, max_keyframes=64)
    R = reg.Registrar(p)
    from mandala_mapping_amd.pointcloud2 import encode_xyz
    msgs = [encode_xyz(t[0]) for t in tr]
    bufs = [np.frombuffer(m.data, np.uint8) for m in msgs]
    clouds = R.clouds(msgs)
    G = reg.LoopCloser(R, P)
    for k, (t, c) in enumerate(zip(tr, clouds)):
        d = abi.CloudDesc()
        d.data, d.n, d.point_step, d.off_x, d.off_y, d.off_z, d.data_is_device, d.source_only = bufs[k].ctypes.data, msgs[k].n, 16, 0, 4, 8, 0, 0
        G.add_keyframe(c, t[2], payload=d)
    cands = G.candidates()
    n = len(cands)
    assert 10 <= n <= 40 and all(c.source - c.target >= 15 for c in cands)
    out = np.zeros((n, 16), np.float32)
    st = (abi.Stats * n)()
    R._check(reg.lib().m3dreg_align_batch(R._h, G.pairs(cands), n, out.ctypes.data_as(C.POINTER(C.c_float)), st), "align_batch")
    acc = G.gate(cands, list(st), min_corr=2000, max_rms=0.05)
    assert sum(acc) >= n - 2
    drift = []
    for i, c in enumerate(cands):
        T = out[i].reshape(4, 4).T.astype(np.float64)
        Ttrue = synth.inv_T(tr[c.target][1]) @ tr[c.source][1]
        rot, tra = synth.pose_error(T, Ttrue)
        rot0, tra0 = synth.pose_error(np.asarray(list(c.init_T), np.float64).reshape(4, 4).T, Ttrue)
        drift.append(tra0)
        if acc[i]:
            assert rot < 0.15 and tra < 0.02, (c.source, c.target, rot, tra)
    assert max(drift) > 0.05            # (the guesses really were off: the closures corrected something)
    # one pair against the oracle, bit for bit
    c = cands[0]
    To, sto, _ = orc.align(p, orc.Cloud(p, tr[c.source][0]), orc.Cloud(p, tr[c.target][0]), np.asarray(list(c.init_T), np.float64).reshape(4, 4).T)
    assert np.array_equal(out[0].reshape(4, 4).T.astype(np.float64), To) and st[0].n_corr == sto.n_corr
    # the same candidates through the one-process multi-device call: payload descriptors, target groups
    M = reg.MultiRegistrar(p, devices=(0, 0))
    Tm, stm, dev = M.align_described(G.pair_descs(cands))
    assert np.array_equal(np.stack([out[i].reshape(4, 4).T for i in range(n)]).astype(np.float64), Tm.astype(np.float64))
    groups = {c.target for c in cands}
    assert M.clouds_bucketed() == n + len(groups)     # every source once, every target keyframe once
    for t in groups:
        assert len({int(dev[i]) for i, c in enumerate(cands) if c.target == t}) == 1


@pytest.mark.gpu
def test_loop_edge_cases(reg, orc):
    p = abi.Params.make(leaf=0.25, iterations=3, max_corr_dist=0.5, metric=abi.POINT_TO_POINT)
    R = reg.Registrar(p)
    with pytest.raises(abi.M3dregError):
        reg.LoopCloser(R, abi.LoopParams.make(sig_log2_bits=19))
    with pytest.raises(abi.M3dregError):
        reg.LoopCloser(R, abi.LoopParams.make(top_k=0))
    rng = np.random.default_rng(3)
    cloud = rng.uniform(-6, 6, (3000, 3)).astype(np.float32)
    cloud[::50] = np.nan
    T = lambda x: synth.make_T(np.eye(3), np.array([x, 0.0, 0.0]))
    for bits in (10, 12, 18):
        P = abi.LoopParams.make(sig_leaf=1.0, sig_log2_bits=bits, radius=3.0, min_gap=2, top_k=3, min_overlap=0.9, max_keyframes=6)
        G, O = reg.LoopCloser(R, P), orc.Loop(P)
        assert len(G.candidates()) == 0                      # empty database
        c = R.cloud(cloud)
        for x in (0.0, 0.0, 0.0, 100.0, 0.0, 2.0):
            G.add_keyframe(c, T(x)); O.add_keyframe(cloud, T(x))
        with pytest.raises(abi.M3dregError):
            G.add_keyframe(c, T(0.0))                        # full
        with pytest.raises(abi.M3dregError):
            G.update_pose(2, np.full((4, 4), np.nan))
        assert [_record(x) for x in G.candidates()] == [_record(x) for x in O.candidates()]
        assert [(x.source, x.target) for x in G.candidates()] == [(2, 0), (4, 0), (4, 1), (4, 2)]
        for k in range(6):
            assert np.array_equal(G.signature(k)[0], O.signature(k)[0])
        G.clear()
        assert len(G) == 0 and len(G.candidates()) == 0


@pytest.mark.gpu
def test_many_keyframes_several_launch_chunks(reg, orc):
    """700 keyframes (more than the 256 rows of one scoring launch; row tiles, column splits and the chunk loop all in play): small random clouds
    on a random walk, every candidate equal to the oracle's."""
    rng = np.random.default_rng(11)
    p = abi.Params.make(leaf=0.5, iterations=2, max_corr_dist=1.0, metric=abi.POINT_TO_POINT)
    R = reg.Registrar(p)
    P = abi.LoopParams.make(sig_leaf=1.5, sig_log2_bits=13, radius=5.0, min_gap=7, top_k=4, min_overlap=0.2, max_keyframes=700)
    G, O = reg.LoopCloser(R, P), orc.Loop(P)
    base = [rng.normal(0, 4.0, (400, 3)).astype(np.float32) for _ in range(16)]
    cl = R.clouds(base)
    pos = np.zeros(3)
    for k in range(700):
        pos = pos + rng.normal(0, 0.8, 3) * np.array([1.0, 1.0, 0.1])
        Tk = synth.make_T(synth.rot_z(rng.uniform(-3, 3)), pos)
        G.add_keyframe(cl[k % 16], Tk); O.add_keyframe(base[k % 16], Tk)
    got, want = G.candidates(), O.candidates()
    assert len(got) == len(want) > 500
    assert [_record(c) for c in got] == [_record(c) for c in want]
    assert [_record(c) for c in G.candidates(250, 300)] == [_record(c) for c in O.candidates(250, 300)]
