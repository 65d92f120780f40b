"""HIP path vs CPU oracle on the same seeded inputs — BIT-EXACT at every stage (no tolerance):
voxel keys, stable permutation, cell-sorted points, normals, NN indices and distances, the 29
fixed-point sums, every per-iteration pose and the final pose/stats. Everything goes through the
C ABI (ctypes). Run on the GPU box with `pytest -m gpu`."""
import numpy as np
import pytest

from mandala_mapping_amd import abi, synth
from mandala_mapping_amd import pointcloud2 as pc2

pytestmark = pytest.mark.gpu


def _params(**kw):
    return abi.Params.make(**kw)


def _same_stats(a, b):
    assert (a.status, a.iterations, a.n_corr) == (b.status, b.iterations, b.n_corr)
    assert a.rms == b.rms and a.last_rot == b.last_rot and a.last_trans == b.last_trans


def _check_bucketing(reg_cloud, orc_cloud, levels):
    for l in range(levels):
        g, go = reg_cloud.grid_info(l), orc_cloud.grid_info(l)
        assert bytes(g) == bytes(go), (g.as_dict(), go.as_dict())
        e, eo = reg_cloud.export(l), orc_cloud.export(l)
        nv = g.n_valid
        assert np.array_equal(e["keys"], eo["keys"])
        assert np.array_equal(e["sorted_keys"], eo["sorted_keys"])
        assert np.array_equal(e["perm"], eo["perm"])
        assert np.array_equal(e["sorted_xyz"][:nv].view(np.uint32), eo["sorted_xyz"][:nv].view(np.uint32))
        if g.has_normals:
            assert np.array_equal(e["normals"][:nv].view(np.uint32), eo["normals"][:nv].view(np.uint32))


def test_large_cloud_takes_the_separate_scan_of_the_radix_sort(reg, orc):
    """More than 128 sort tiles (262 144 points): the counter scan is its own launch again (bucket.hip: RS_FUSED_TILES); a batch that
    mixes such a cloud with a small one takes that path for both. Bucketing and normals bit-exact against the oracle."""
    rng = np.random.default_rng(77)
    big = np.concatenate([synth.planes_cloud(280000, 78, sigma=0.02, size=40.0), rng.uniform(-5, 45, (20000, 3)).astype(np.float32)])
    small = synth.planes_cloud(5000, 79, sigma=0.01, size=8.0)
    p = _params(leaf=0.25, iterations=2, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    cb, cs = R.clouds([big, small])
    _check_bucketing(cb, orc.Cloud(p, big), 1)
    _check_bucketing(cs, orc.Cloud(p, small), 1)
    _check_bucketing(R.cloud(small), orc.Cloud(p, small), 1)   # (alone: the fused path)


@pytest.mark.parametrize("leaf,normal_leaf", [(0.25, 0.07), (0.1, 1.3), (0.2, 0.2), ((0.6, 0.15), 0.33)])
def test_normals_for_any_ratio_of_normal_leaf_to_leaf(reg, orc, leaf, normal_leaf):
    """Round 5: the normal-estimation grid is no longer sorted — its voxels are a hash table filled from the finest level's sorted order (bucket.hip:
    k_finalize_level / k_post_finalize / k_tiles_normals / k_nrm_handout). A normal voxel is then one short run of that order only when normal_leaf is a
    multiple of the leaf; here it is smaller than a voxel (every voxel of the level holds several normal voxels, interleaved: many runs per voxel, the
    atomic path), much larger, equal, and incommensurable under a pyramid (every level's order gets the normals) — with non-finite points, a crowded
    patch and a sparse far field in the cloud. Keys, order and normals bit for bit against the oracle."""
    rng = np.random.default_rng(5)
    base = synth.planes_cloud(30000, 91, sigma=0.01, size=12.0)
    crowd = (rng.normal(0, 0.03, (6000, 3)) + np.float32([1.0, 1.0, 0.0])).astype(np.float32)          # thousands of points in a few voxels
    far = rng.uniform(-30, 30, (3000, 3)).astype(np.float32)                                            # a voxel per point
    cloud = np.concatenate([base, crowd, far]).astype(np.float32)
    cloud = cloud[rng.permutation(len(cloud))]
    cloud[::977] = np.nan
    levels = len(leaf) if isinstance(leaf, tuple) else 1
    p = _params(leaf=leaf, iterations=(1,) * levels if levels > 1 else 1, max_corr_dist=(0.5,) * levels if levels > 1 else 0.5, metric=abi.POINT_TO_PLANE, normal_leaf=normal_leaf)
    R = reg.Registrar(p)
    c1, c2 = R.clouds([cloud, base])        # (a batch of two: the second cloud's tables lie behind the first's in the workspace)
    _check_bucketing(c1, orc.Cloud(p, cloud), levels)
    _check_bucketing(c2, orc.Cloud(p, base), levels)
    e = c1.export(levels - 1)
    assert np.any(e["normals"][: c1.grid_info(levels - 1).n_valid] != 0)


@pytest.mark.parametrize("metric", [abi.POINT_TO_POINT, abi.POINT_TO_PLANE])
def test_config1_every_stage_bit_exact(reg, orc, metric):
    src, tgt, Tgt = synth.config1()
    p = _params(leaf=0.25, iterations=12, max_corr_dist=0.5, metric=metric, normal_leaf=0.5)
    R = reg.Registrar(p)
    cs, ct = R.cloud(src), R.cloud(tgt)
    os_, ot = orc.Cloud(p, src), orc.Cloud(p, tgt)
    _check_bucketing(ct, ot, 1)
    _check_bucketing(cs, os_, 1)
    # NN of the transformed source
    q = synth.apply_T(Tgt, src).astype(np.float32)
    i1, d1 = ct.nn(q, 0.5)
    i2, d2 = ot.nn(q, 0.5)
    assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
    # one linearisation at identity and at the ground truth
    for T in (np.eye(4), Tgt):
        s1, e1 = R.accumulate(cs, ct, T)
        s2, e2 = orc.accumulate(p, os_, ot, T)
        assert np.array_equal(e1, e2)
        assert np.array_equal(s1, s2), (s1 - s2)
    # full run
    T1, st1 = R.align(cs, ct)
    tr1 = R.trace()
    T2, st2, tr2 = orc.align(p, os_, ot, trace_cap=64)
    assert np.array_equal(tr1, tr2)
    assert np.array_equal(T1, T2)
    _same_stats(st1, st2)
    rot, tra = synth.pose_error(T1, Tgt)
    if metric == abi.POINT_TO_PLANE:
        assert rot < 0.05 and tra < 0.005


def test_nonfinite_points_and_odd_layout(reg, orc):
    src, tgt, _ = synth.config1(5000)
    tgt = tgt.copy()
    tgt[::97] = np.nan
    tgt[5] = [np.inf, 0, 0]
    src = src.copy()
    src[::131, 1] = np.nan
    p = _params(leaf=0.25, iterations=6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    mt = pc2.encode_xyz(tgt, point_step=32, offsets=(4, 12, 20))
    ms = pc2.encode_xyz(src, point_step=20, offsets=(8, 0, 4))
    ct, cs = R.cloud(mt), R.cloud(ms)
    ot = orc.Cloud(p, mt.data, mt.n, 32, (4, 12, 20))
    os_ = orc.Cloud(p, ms.data, ms.n, 20, (8, 0, 4))
    _check_bucketing(ct, ot, 1)
    T1, st1 = R.align(cs, ct)
    T2, st2, _ = orc.align(p, os_, ot)
    assert np.array_equal(T1, T2)
    _same_stats(st1, st2)


def test_unaligned_payload_is_repacked(reg, orc):
    xyz = synth.planes_cloud(2000, 9)
    m = pc2.encode_xyz(xyz, point_step=13, offsets=(1, 5, 9))
    p = _params(leaf=0.25, metric=abi.POINT_TO_POINT)
    R = reg.Registrar(p)
    c = R.cloud(m)
    o = orc.Cloud(p, m.data, m.n, 13, (1, 5, 9))
    _check_bucketing(c, o, 1)


def test_multiresolution_hdl32_bit_exact(reg, orc):
    src, tgt, Tgt = synth.hdl32_pair(700, 100, 101)   # 22 400 rays: the oracle finishes in about a second
    p = _params(leaf=(0.4, 0.2), iterations=(8, 8), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    cs, ct = R.cloud(src), R.cloud(tgt)
    os_, ot = orc.Cloud(p, src), orc.Cloud(p, tgt)
    _check_bucketing(ct, ot, 2)
    T1, st1 = R.align(cs, ct)
    tr1 = R.trace()
    T2, st2, tr2 = orc.align(p, os_, ot, trace_cap=64)
    assert np.array_equal(tr1, tr2) and np.array_equal(T1, T2)
    _same_stats(st1, st2)
    rot, tra = synth.pose_error(T1, Tgt)
    assert rot < 0.1 and tra < 0.03, (rot, tra)


def test_batch_equals_single_and_oracle(reg, orc):
    p = _params(leaf=0.25, iterations=8, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    pairs, ref = [], []
    for k in range(5):
        tgt = synth.planes_cloud(3000 + 500 * k, 50 + k)
        Tg = synth.random_T(np.random.default_rng(k), 2.0, 0.1)
        src = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(2500 + 300 * k, 80 + k)).astype(np.float32)
        pairs.append((R.cloud(src), R.cloud(tgt), None))
        ref.append(orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt)))
    Tb, stb = R.align_batch(pairs)
    for k in range(5):
        assert np.array_equal(Tb[k], ref[k][0])
        _same_stats(stb[k], ref[k][1])
        T1, st1 = R.align(pairs[k][0], pairs[k][1])
        assert np.array_equal(T1, Tb[k])


def test_synchronous_and_asynchronous_calls_give_the_same_bits(reg, orc):
    """m3dreg_align_batch (waits; convergence-terminated batches are enqueued a few iterations ahead of the device and a level is cut short as soon as the device
    reports it finished) and m3dreg_align_batch_async + m3dreg_batch_wait (enqueues everything; launches behind a finished level leave at once): batches of 9 ... 1
    pairs that finish after different numbers of iterations, fixed iteration counts and eps-terminated, one level and a pyramid, the bucketing of the clouds still in
    flight when the call is made — the oracle's bits from either call, and the handle stays usable for single registrations in between."""
    for levels in (dict(leaf=0.25, iterations=12), dict(leaf=(0.5, 0.25), iterations=(6, 10), max_corr_dist=(1.5, 0.6))):
        for eps in (0.0, 1e-5):
            p = _params(metric=abi.POINT_TO_PLANE, normal_leaf=0.5, eps_rot=eps, eps_trans=eps, **levels)
            R = reg.Registrar(p)
            raw, ref = [], []
            for k in range(9):
                tgt = synth.planes_cloud(3000 + 400 * k, 150 + k)
                Tg = synth.random_T(np.random.default_rng(100 + k), 1.0 + 0.2 * k, 0.1)
                src = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(2500 + 300 * k, 180 + k)).astype(np.float32)
                raw.append((src, tgt))
                ref.append(orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt)))
            for n in (9, 5, 2, 1):
                cl = R.clouds([a for s_, t_ in raw[:n] for a in (s_, t_)], wait=False)      # bucketing still in flight when the call is made
                pairs = [(cl[2 * i], cl[2 * i + 1], None) for i in range(n)]
                Tb, stb = R.align_batch(pairs)
                R.align_batch_async(R._pairs(pairs), n)
                Ta, sta = R.batch_wait(n)
                for k in range(n):
                    assert np.array_equal(Tb[k], ref[k][0]) and np.array_equal(Ta[k], ref[k][0]), (levels, eps, n, k)
                    _same_stats(stb[k], ref[k][1]); _same_stats(sta[k], ref[k][1])
                T1, _ = R.align(pairs[n - 1][0], pairs[n - 1][1])
                assert np.array_equal(T1, Tb[n - 1])
                for c in cl:
                    c.free()


def test_dense_level_schedule_is_a_function_of_the_batch(reg, orc):
    """ABI 8: which search kernels a level launches (k_nn_iter alone / k_nn_iter + k_nn_coop / k_nn_coop alone) is decided per batch — from the clouds' own counts
    where the host has read them back (synchronous creation), "launch both, the device decides" where it has not (enqueue-only creation) — and no longer from what
    the handle's previous batch looked like. A dense pair (60 000-point target, hundreds of points per coarse voxel) and an ordinary one, alone and mixed, through
    both creation paths, in every order on ONE handle, and on a fresh handle each: always the oracle's bits."""
    p = _params(leaf=(0.4, 0.1), iterations=(8, 6), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    dense_t = synth.planes_cloud(60000, 3100, sigma=0.01, size=3.0)
    Tg = synth.make_T(synth.rot_z(np.radians(2.0)) @ synth.rot_x(np.radians(-1.0)), np.array([0.12, -0.08, 0.05]))
    dense_s = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(20000, 3101, sigma=0.01, size=3.0).astype(np.float64)).astype(np.float32)
    plain_s, plain_t, _ = synth.hdl32_pair(500, 71, 72, dx=0.2, dy=0.05, dyaw_deg=1.5)
    data = {"dense": (dense_s, dense_t), "plain": (plain_s, plain_t)}
    ref = {k: orc.align(p, orc.Cloud(p, s, omp=True, source_only=True), orc.Cloud(p, t, omp=True)) for k, (s, t) in data.items()}
    shared = reg.Registrar(p)
    for order in (("dense",), ("plain",), ("dense", "plain"), ("plain", "plain"), ("dense", "dense"), ("plain", "dense"), ("dense",)):
        for wait in (True, False):
            for R in (shared, reg.Registrar(p)):
                cl = R.clouds([a for k in order for a in data[k]], wait=wait, source_only=[True, False] * len(order))
                T, st = R.align_batch([(cl[2 * i], cl[2 * i + 1], None) for i in range(len(order))])
                for i, k in enumerate(order):
                    assert np.array_equal(T[i], ref[k][0]), (order, wait, R is shared, i)
                    _same_stats(st[i], ref[k][1])
                for c in cl:
                    c.free()


def test_latency_mode_gives_the_same_bits(reg, orc):
    """ABI 7, m3dreg_set_latency_mode: a serial caller's statement that its batches have the GPU to themselves changes launch grids (the reduction pass's
    workgroups per pair), never a bit: batches and single registrations with the mode on, off and toggled between batches equal the oracle; the call is refused
    between an asynchronous call and its wait."""
    p = _params(leaf=0.25, iterations=10, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    raw, ref = [], []
    for k in range(8):
        tgt = synth.planes_cloud(9000 + 1500 * k, 400 + k)
        Tg = synth.random_T(np.random.default_rng(500 + k), 2.0, 0.1)
        src = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(8000 + 1000 * k, 430 + k)).astype(np.float32)
        raw.append((src, tgt))
        ref.append(orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt)))
    for on, n in ((True, 8), (False, 8), (True, 3), (True, 1), (False, 1), (True, 8)):
        R.set_latency_mode(on)
        cl = R.clouds([a for s_, t_ in raw[:n] for a in (s_, t_)])
        pairs = [(cl[2 * i], cl[2 * i + 1], None) for i in range(n)]
        Tb, stb = R.align_batch(pairs)
        for k in range(n):
            assert np.array_equal(Tb[k], ref[k][0]), (on, n, k)
            _same_stats(stb[k], ref[k][1])
        if n == 3:
            R.align_batch_async(R._pairs(pairs), n)
            with pytest.raises(Exception):
                R.set_latency_mode(False)
            Ta, _ = R.batch_wait(n)
            assert np.array_equal(Ta, Tb)
        for c in cl:
            c.free()


def test_status_codes(reg, orc):
    p = _params(leaf=0.5, iterations=5, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    rng = np.random.default_rng(3)
    plane = np.c_[rng.uniform(0, 10, size=(4000, 2)), np.zeros(4000)].astype(np.float32)
    T, st = R.align(R.cloud(plane + np.float32([0, 0, 0.02])), R.cloud(plane))
    assert st.status == abi.RANK_DEFICIENT and np.array_equal(T, np.eye(4))
    far = R.cloud(plane + np.float32(60.0))
    T, st = R.align(far, R.cloud(plane))
    assert st.status == abi.TOO_FEW_CORR and st.iterations == 1 and st.n_corr == 0
    with pytest.raises(abi.M3dregError) as ei:
        R.cloud(np.full((8, 3), np.nan, np.float32))
    assert ei.value.code == abi.ERR_EMPTY_CLOUD
    Rf = reg.Registrar(_params(leaf=0.001, metric=abi.POINT_TO_POINT))
    with pytest.raises(abi.M3dregError) as ei:
        Rf.cloud(np.array([[0, 0, 0], [5000, 5000, 5000]], np.float32))
    assert ei.value.code == abi.ERR_GRID_TOO_LARGE


def test_node_surface_pointcloud2_in_pose_out(reg, orc):
    """m3dreg_set_target_xyz / m3dreg_align with the aggregator's exact message layout."""
    src, tgt, Tgt = synth.config1(6000)
    p = _params(leaf=0.25, iterations=15, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    node = reg.Gpu6dSlamNode(p)
    pose0, st0 = node.on_cloud(pc2.encode_xyz(tgt))
    assert st0 is None and np.array_equal(pose0, np.eye(4))
    pose1, st1 = node.on_cloud(pc2.encode_xyz(src))
    T2, st2, _ = orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt))
    assert np.array_equal(pose1, T2)
    _same_stats(st1, st2)


def test_full_size_properties_config3(reg):
    """BASELINE config 3 at full size (100k points): properties that need no oracle run."""
    src, tgt, Tgt = synth.config3()
    p = _params(leaf=0.1, iterations=30, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    R = reg.Registrar(p)
    cs, ct = R.cloud(src), R.cloud(tgt)
    e = ct.export()
    assert (np.diff(e["sorted_keys"].astype(np.int64)) >= 0).all()           # sortedness
    assert np.array_equal(np.sort(e["perm"]), np.arange(len(tgt)))           # a permutation
    assert np.array_equal(e["sorted_xyz"], tgt[e["perm"]])                   # gather is exact
    T, st = R.align(cs, ct)
    rot, tra = synth.pose_error(T, Tgt)
    assert st.status == abi.CONVERGED and rot < 0.02 and tra < 0.005, (rot, tra, st.as_dict())
    # idempotence: restarting from the answer stays there
    T2, st2 = R.align(cs, ct, T)
    assert synth.pose_error(T2, T)[0] < 1e-3 and synth.pose_error(T2, T)[1] < 1e-4
    # source order does not change a single bit of the result
    perm = np.random.default_rng(1).permutation(len(src))
    T3, st3 = R.align(R.cloud(src[perm]), ct)
    assert np.array_equal(T3, T) and st3.n_corr == st.n_corr
    # run-to-run determinism
    T4, _ = R.align(cs, ct)
    assert np.array_equal(T4, T)


def test_tile_search_and_global_walk_are_bit_identical(reg, orc, monkeypatch):
    """The LDS-staged tile search (k_nn_tiles, default) and the global walk (M3DREG_TILES=0) must agree with each other and
    with the oracle on every bit; the big yaw leaves many queries without an occupied home bucket (global-walk list)."""
    src, tgt, Tgt = synth.hdl32_pair(900, 300, 301, dx=0.3, dy=-0.2, dyaw_deg=25.0)
    p = _params(leaf=0.1, iterations=6, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    T0 = synth.perturb(Tgt, np.random.default_rng(5), 1.0, 0.1)
    out = []
    for tiles in ("0", "1"):
        monkeypatch.setenv("M3DREG_TILES", tiles)
        R = reg.Registrar(p)
        cs, ct = R.cloud(src), R.cloud(tgt)
        s, e = R.accumulate(cs, ct, T0)
        T, st = R.align(cs, ct, T0)
        out.append((s, e, T, R.trace(), st, R.counters()))
    assert out[0][5][0] == 0 and out[1][5][0] > 0            # the tile path really ran in the second build (and only there)
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1])
        assert np.array_equal(out[0][3], o[3]) and np.array_equal(out[0][2], o[2])
    To, sto, tro = orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt), T0, trace_cap=16)
    assert np.array_equal(out[1][3], tro) and np.array_equal(out[1][2], To)
    _same_stats(out[1][4], sto)


def test_tall_grid_more_than_2048_voxels_in_z(reg, orc, monkeypatch):
    """A grid of more than 2048 voxels along z (11 bits of bucket coordinate: ADVICE r2 — k_tile_build once packed the own buckets'
    coordinates into 11 / 11 / 10-bit fields and staged the wrong neighbours above bucket 1023): two patches 205 m apart in z, the upper
    one straddling voxel 2048. Tile search, global walk and oracle agree on every bit; the tile path really ran."""
    def tower(seed):
        lo = synth.planes_cloud(6000, seed, sigma=0.01, size=6.0)
        hi = synth.planes_cloud(6000, seed + 1, sigma=0.01, size=6.0) + np.array([0.0, 0.0, 203.0], dtype=np.float32)
        return np.concatenate([lo, hi]).astype(np.float32)
    tgt = tower(900)
    Tg = synth.make_T(synth.rot_z(np.radians(0.5)), np.array([0.05, -0.03, 0.04]))
    src = synth.apply_T(synth.inv_T(Tg), tower(910).astype(np.float64)).astype(np.float32)
    p = _params(leaf=0.1, iterations=6, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    ot, os_ = orc.Cloud(p, tgt), orc.Cloud(p, src)
    assert ot.grid_info(0).dims[2] > 2048
    To, sto, tro = orc.align(p, os_, ot, trace_cap=16)
    q = synth.apply_T(Tg, src).astype(np.float32)
    io, do = ot.nn(q, 0.5)
    for tiles in ("1", "0"):
        monkeypatch.setenv("M3DREG_TILES", tiles)
        R = reg.Registrar(p)
        cs, ct = R.cloud(src), R.cloud(tgt)
        _check_bucketing(ct, ot, 1)
        i1, d1 = ct.nn(q, 0.5)
        assert np.array_equal(i1, io) and np.array_equal(d1.view(np.uint32), do.view(np.uint32))
        T, st = R.align(cs, ct)
        assert np.array_equal(R.trace(), tro) and np.array_equal(T, To)
        _same_stats(st, sto)
        searched_in_tiles, _ = R.counters()
        assert (searched_in_tiles > 0) == (tiles == "1")


@pytest.mark.parametrize("metric", [abi.POINT_TO_POINT, abi.POINT_TO_PLANE])
def test_fused_late_iterations_are_bit_identical(reg, orc, monkeypatch, metric):
    """From the 12th iteration of a level on, search and reduction run as ONE launch (k_icp_late, M3DREG_FUSE_FROM); with 0 every
    iteration is the two- / three-launch chain. 20 fixed iterations, two levels, a batch of three pairs: same trace, pose and
    statistics either way, and equal to the oracle's."""
    pairs = [synth.hdl32_pair(700, 400 + k, 500 + k, dx=0.25 - 0.1 * k, dy=0.05 * k, dyaw_deg=2.0 + k) for k in range(3)]
    p = _params(leaf=(0.4, 0.2), iterations=(16, 20), max_corr_dist=(1.0, 0.5), metric=metric, normal_leaf=0.5, eps_rot=0.0, eps_trans=0.0)
    runs = []
    for fuse in ("0", "12", "10"):
        monkeypatch.setenv("M3DREG_FUSE_FROM", fuse)
        R = reg.Registrar(p)
        clouds = [tuple(R.clouds([s, t], source_only=[True, False])) for s, t, _ in pairs]
        T, st = R.align_batch([(cs, ct, None) for cs, ct in clouds])
        T1, st1 = R.align(clouds[0][0], clouds[0][1])
        runs.append((T, [(x.status, x.iterations, x.n_corr, x.rms) for x in st], T1, R.trace()))
    for r in runs[1:]:
        assert np.array_equal(runs[0][0], r[0]) and runs[0][1] == r[1]
        assert np.array_equal(runs[0][2], r[2]) and np.array_equal(runs[0][3], r[3])
    s, t, _ = pairs[0]
    To, sto, tro = orc.align(p, orc.Cloud(p, s, source_only=True), orc.Cloud(p, t), trace_cap=64)
    assert np.array_equal(runs[1][2], To) and np.array_equal(runs[1][3], tro)
    assert runs[1][1][0][1] == 36 and sto.iterations == 36


def test_batched_bucketing_equals_single(reg, orc):
    """m3dreg_cloud_create_batch (one pipeline for many clouds of different sizes, some with non-finite
    points) produces exactly the clouds the one-at-a-time path and the oracle produce."""
    p = _params(leaf=(0.5, 0.25), iterations=(3, 3), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    arrays = []
    for k in range(5):
        a = synth.planes_cloud(1500 + 700 * k, 20 + k).copy()
        if k % 2:
            a[::53] = np.nan
        arrays.append(a)
    batch = R.clouds(arrays)
    for a, c in zip(arrays, batch):
        o = orc.Cloud(p, a)
        _check_bucketing(c, o, 2)
        _check_bucketing(R.cloud(a), o, 2)


CONFIG2 = dict(leaf=(0.8, 0.4, 0.2), iterations=(30, 30, 150), max_corr_dist=(2.0, 0.6, 0.2), metric=abi.POINT_TO_POINT)   # == bench.py's config-2 leg


def test_config2_hdl32_point_to_point(reg, orc):
    """BASELINE config 2: single HDL-32 scan pair (70 016 rays), point-to-point, 1 x MI355X, FROM IDENTITY (0.5 m / 3 deg apart) — bit-exact
    vs the oracle, and inside the acceptance bar BASELINE.md states for it: <= 0.1 deg / 2 cm (measured 0.076 deg / 1.0 cm; the single
    0.2 m level of round 2 ended 0.47 m off from identity). Point-to-point on ring-structured sweeps needs the pyramid to get there and a
    correspondence distance below the ring spacing at the end (with 1.0 m the ground rings pull the translation 12 cm short); its
    rotation floor at sigma = 2 cm range noise is what remains — point-to-plane reaches 0.007 deg / 0.8 mm on the same pair (below)."""
    src, tgt, Tgt = synth.config2()
    p = _params(**CONFIG2)
    R = reg.Registrar(p)
    cs, ct = R.clouds([src, tgt], source_only=[True, False])
    T1, st1 = R.align(cs, ct)
    T2, st2, tr2 = orc.align(p, orc.Cloud(p, src, omp=True, source_only=True), orc.Cloud(p, tgt, omp=True), trace_cap=256)
    assert np.array_equal(R.trace(), tr2) and np.array_equal(T1, T2)
    _same_stats(st1, st2)
    rot, tra = synth.pose_error(T1, Tgt)
    assert st1.status == abi.CONVERGED and rot <= 0.1 and tra <= 0.02, (rot, tra, st1.as_dict())
    # the library's DEFAULT parameters (what a node launched without parameters runs: point-to-plane, 0.4 m -> 0.1 m) on the same pair
    pd = reg.default_params()
    Rd = reg.Registrar(pd)
    Td, std = Rd.align(*Rd.clouds([src, tgt], source_only=[True, False]))
    To, sto, _ = orc.align(pd, orc.Cloud(pd, src, omp=True, source_only=True), orc.Cloud(pd, tgt, omp=True))
    assert np.array_equal(Td, To)
    _same_stats(std, sto)
    rot, tra = synth.pose_error(Td, Tgt)
    assert std.status == abi.CONVERGED and rot <= 0.05 and tra <= 0.005, (rot, tra)


def test_config2_as_stated_in_the_survey_single_level(reg, orc):
    """SURVEY §8(d) states config 2 as ONE level: 70 016 rays per sweep, point-to-point, leaf 0.2 m, d_max 1.0 m, eps 1e-5 or 30 iterations, from identity. Run exactly
    so, as a parity case: every per-iteration pose, the final pose and the statistics equal the oracle's bit for bit. It does NOT converge to the ground truth
    — point-to-point with a 1 m gate on ring-structured sweeps 0.5 m / 3 deg apart slides along the rings (it ends ~0.47 m off) — which is why the bench's
    config-2 leg and test_config2_hdl32_point_to_point use the 0.8 / 0.4 / 0.2 m pyramid; the stated parameters are kept here so that they stay exercised as written."""
    src, tgt, Tgt = synth.config2()
    assert len(src) > 60000 and len(tgt) > 60000
    p = _params(leaf=0.2, iterations=30, max_corr_dist=1.0, metric=abi.POINT_TO_POINT, eps_rot=1e-5, eps_trans=1e-5)
    R = reg.Registrar(p)
    cs, ct = R.clouds([src, tgt], source_only=[True, False])
    _check_bucketing(ct, orc.Cloud(p, tgt, omp=True), 1)
    T1, st1 = R.align(cs, ct)
    T2, st2, tr2 = orc.align(p, orc.Cloud(p, src, omp=True, source_only=True), orc.Cloud(p, tgt, omp=True), trace_cap=32)
    assert st2.iterations >= 5 and np.array_equal(R.trace(), tr2) and np.array_equal(T1, T2)
    _same_stats(st1, st2)
    rot, tra = synth.pose_error(T1, Tgt)
    assert tra > 0.05            # (documents the statement above: if this ever converges, BASELINE.md §3 and the bench's config-2 leg should go back to the stated level)


def test_config5_dense_map_multiresolution(reg, orc):
    """BASELINE config 5 (reduced to 5 sweeps = ~0.5 M map points so the oracle finishes in seconds):
    live scan against a merged map, leaf 0.4 -> 0.2 -> 0.1, bit-exact vs the oracle."""
    live, mp, Tgt, T0 = synth.config5(n_scans=5)
    p = _params(leaf=(0.4, 0.2, 0.1), iterations=(10, 10, 10), max_corr_dist=(1.0, 0.5, 0.3), metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    R = reg.Registrar(p)
    cs, ct = R.clouds([live, mp])
    _check_bucketing(ct, orc.Cloud(p, mp, omp=True), 3)
    T1, st1 = R.align(cs, ct, T0)
    T2, st2, tr2 = orc.align(p, orc.Cloud(p, live, omp=True), orc.Cloud(p, mp, omp=True), T0, trace_cap=64)
    assert np.array_equal(R.trace(), tr2) and np.array_equal(T1, T2)
    _same_stats(st1, st2)
    rot, tra = synth.pose_error(T1, Tgt)
    assert rot < 0.05 and tra < 0.01, (rot, tra)
    # the other schedule: clouds out of the enqueue-only bucketing, whose counts the host has not read back — both search kernels are launched on every level and the
    # device decides pair by pair (ABI 8: the schedule is a function of the batch, not of the handle's previous one) — the same bits
    ca, cta = R.clouds([live, mp], wait=False, source_only=[True, False])
    for _ in range(2):
        T3, st3 = R.align(ca, cta, T0)
        assert np.array_equal(R.trace(), tr2) and np.array_equal(T3, T2)
        _same_stats(st3, st2)


def test_exact_distance_ties_go_to_the_lowest_input_index(reg, orc, monkeypatch):
    """Spec: ties of the squared distance go to the lowest INPUT index. Two lattices of exactly representable coordinates, the source half a spacing off:
    at the first iteration EVERY query is exactly equally far from two (or four) target points that lie in DIFFERENT voxels, and the target's input order is
    shuffled, so sorted position and input index disagree about who comes first. The LDS tile search, the global walk and the oracle must agree on every pose.
    (Round 3 tried staged points that carry their sorted position — the winner's position being the answer, input indices fetched only for such ties — to
    spare the search its last dependent load; exact, by this test, and 19 % slower: the tie check sits in the innermost loop.)"""
    rng = np.random.default_rng(12)
    ii, jj, kk = np.meshgrid(np.arange(40), np.arange(40), np.arange(3), indexing="ij")
    tgt = np.stack([0.5 * ii, 0.5 * jj, 1.0 * kk], -1).reshape(-1, 3).astype(np.float32)
    tgt = tgt[rng.permutation(len(tgt))]
    src = (tgt + np.array([0.25, 0.25, 0.0], np.float32))[rng.permutation(len(tgt))][:3000]
    p = _params(leaf=0.25, iterations=4, max_corr_dist=0.6, metric=abi.POINT_TO_POINT, eps_rot=0.0, eps_trans=0.0)
    ot, os_ = orc.Cloud(p, tgt), orc.Cloud(p, src)
    q = src[:500]
    io, do = ot.nn(q, 0.6)
    e = q[:, None, :] - tgt[None, :, :]
    d2 = (e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]
    assert ((d2 == d2.min(1, keepdims=True)).sum(1) >= 2).all()              # every query really has an exact tie
    assert np.array_equal(io, np.argmax(d2 == d2.min(1, keepdims=True), axis=1))   # and the oracle takes the lowest input index
    To, sto, tro = orc.align(p, os_, ot, trace_cap=8)
    for tiles in ("1", "0"):
        monkeypatch.setenv("M3DREG_TILES", tiles)
        R = reg.Registrar(p)
        cs, ct = R.cloud(src), R.cloud(tgt)
        i1, d1 = ct.nn(q, 0.6)
        assert np.array_equal(i1, io) and np.array_equal(d1.view(np.uint32), do.view(np.uint32))
        s1, e1 = R.accumulate(cs, ct, np.eye(4))
        s2, e2 = orc.accumulate(p, os_, ot, np.eye(4))
        assert np.array_equal(s1, s2) and np.array_equal(e1, e2)
        T, st = R.align(cs, ct)
        assert np.array_equal(R.trace(), tro) and np.array_equal(T, To)
        _same_stats(st, sto)
        assert (R.counters()[0] > 0) == (tiles == "1")


def test_crowded_coarse_level_is_walked_cooperatively_and_stays_exact(reg, orc):
    """A pyramid whose coarse level holds hundreds of points per voxel (a dense map): that level is sorted from the finest level's order, has no
    tiles, and k_patch_jobs sends every one of its searches to the eight-lanes-per-query walk (M3dJob::coop_always, more than 48 points per occupied
    voxel) — exports in the spec's order, every per-iteration pose and the statistics equal to the oracle's; point-to-point and point-to-plane."""
    tgt = synth.planes_cloud(60000, 3100, sigma=0.01, size=3.0)
    Tg = synth.make_T(synth.rot_z(np.radians(2.0)) @ synth.rot_x(np.radians(-1.0)), np.array([0.12, -0.08, 0.05]))
    src = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(20000, 3101, sigma=0.01, size=3.0).astype(np.float64)).astype(np.float32)
    for metric in (abi.POINT_TO_PLANE, abi.POINT_TO_POINT):
        p = _params(leaf=(0.4, 0.1), iterations=(10, 6), max_corr_dist=(1.0, 0.5), metric=metric, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
        R = reg.Registrar(p)
        cs, ct = R.clouds([src, tgt], source_only=[True, False])
        ot = orc.Cloud(p, tgt, omp=True)
        g0 = ct.grid_info(0)
        assert g0.n_valid > 48 * g0.n_cells                       # the coarse level IS crowded
        _check_bucketing(ct, ot, 2)
        T, st = R.align(cs, ct)
        To, sto, tro = orc.align(p, orc.Cloud(p, src, omp=True, source_only=True), ot, trace_cap=32)
        assert np.array_equal(R.trace(), tro) and np.array_equal(T, To)
        _same_stats(st, sto)
        e0, e1 = synth.pose_error(np.eye(4), Tg), synth.pose_error(T, Tg)
        assert e1[0] < e0[0] and e1[1] < e0[1], (e0, e1)        # (parity is the point here; 16 iterations on three 3 m planes only have to go the right way)


@pytest.mark.parametrize("n_pairs", [3, 8])
def test_batch_mixing_crowded_and_ordinary_coarse_levels(reg, orc, n_pairs):
    """One batch, two levels: some pairs' coarse level is crowded (k_nn_coop answers them, k_nn_iter's workgroups leave), the others' is not (the other way
    round) — decided per pair on the device. 3 pairs (block -> pair map by division) and 8 (one pair per XCD): every pose equals the pair registered alone
    and the oracle's."""
    p = _params(leaf=(0.4, 0.1), iterations=(6, 5), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    R = reg.Registrar(p)
    data = []
    for k in range(n_pairs):
        if k % 2 == 0:   # dense planes: hundreds of points per 0.4 m voxel
            tgt = synth.planes_cloud(40000 + 3000 * k, 3300 + k, sigma=0.01, size=3.0)
            Tg = synth.make_T(synth.rot_z(np.radians(1.0 + 0.2 * k)), np.array([0.08, -0.05, 0.03]))
            src = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(12000, 3400 + k, sigma=0.01, size=3.0).astype(np.float64)).astype(np.float32)
        else:            # an HDL-32-shaped sweep: a handful of points per voxel
            src, tgt, Tg = synth.hdl32_pair(500 + 40 * k, 3500 + k, 3600 + k, dx=0.2, dy=0.05, dyaw_deg=1.5)
        data.append((src, tgt))
    clouds = [tuple(R.clouds([s_, t_], source_only=[True, False])) for s_, t_ in data]
    dens = [c[1].grid_info(0).n_valid / max(1, c[1].grid_info(0).n_cells) for c in clouds]
    assert max(dens) > 48 and min(dens) < 48, dens
    Tb, stb = R.align_batch([(cs, ct, None) for cs, ct in clouds])
    for k, (src, tgt) in enumerate(data):
        T1, st1 = R.align(clouds[k][0], clouds[k][1])
        assert np.array_equal(T1, Tb[k]) and st1.n_corr == stb[k].n_corr, k
        if k < 3:
            To, sto, _ = orc.align(p, orc.Cloud(p, src, omp=True, source_only=True), orc.Cloud(p, tgt, omp=True))
            assert np.array_equal(Tb[k], To), k
            _same_stats(stb[k], sto)
    # the dense pairs as a batch of their own: EVERY pair has a dense coarse level and the host knows it (synchronous creation read the counts back), so
    # k_nn_coop is launched alone there (no classifying launch) — several pairs in that schedule, the same bits
    dense = [k for k in range(n_pairs) if k % 2 == 0]
    for _ in range(2):
        Td, std = R.align_batch([(clouds[k][0], clouds[k][1], None) for k in dense])
        for j, k in enumerate(dense):
            assert np.array_equal(Td[j], Tb[k]) and std[j].n_corr == stb[k].n_corr, k


def test_config5_full_size_properties(reg):
    """BASELINE config 5 at full size: 2 M-point map (2 066 481 points: 22 sweeps de-duplicated at 1 cm) vs 100 k live scan, multi-resolution voxel NN."""
    live, mp, Tgt, T0 = synth.config5()
    assert len(mp) >= 2_000_000
    p = _params(leaf=(0.4, 0.2, 0.1), iterations=(10, 10, 10), max_corr_dist=(1.0, 0.5, 0.3), metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    R = reg.Registrar(p)
    cs, ct = R.clouds([live, mp])
    e = ct.export(2)
    assert (np.diff(e["sorted_keys"].astype(np.int64)) >= 0).all()
    assert np.array_equal(np.sort(e["perm"]), np.arange(len(mp)))
    T, st = R.align(cs, ct, T0)
    rot, tra = synth.pose_error(T, Tgt)
    assert st.status in (abi.CONVERGED, abi.MAX_ITERATIONS) and rot < 0.05 and tra < 0.01, (rot, tra, st.as_dict())
    T2, _ = R.align(cs, ct, T0)
    assert np.array_equal(T, T2)


def test_degenerate_clouds(reg, orc):
    """All points identical (one voxel, one huge bucket), a single point, and two far-apart points."""
    p = _params(leaf=0.1, iterations=3, metric=abi.POINT_TO_POINT)
    R = reg.Registrar(p)
    same = np.tile(np.float32([1.0, 2.0, 3.0]), (70000, 1))     # > 65535 points in one bucket: the bigcum path
    c, o = R.cloud(same), orc.Cloud(p, same)
    _check_bucketing(c, o, 1)
    q = np.float32([[1.0, 2.0, 3.05], [1.2, 2.0, 3.0]])
    i1, d1 = c.nn(q, 0.5)
    i2, d2 = o.nn(q, 0.5)
    assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32)) and i1[0] == 0
    one = np.float32([[0.5, 0.5, 0.5]])
    _check_bucketing(R.cloud(one), orc.Cloud(p, one), 1)
    two = np.float32([[0, 0, 0], [50, 40, 3]])
    _check_bucketing(R.cloud(two), orc.Cloud(p, two), 1)
    T, st = R.align(R.cloud(two), R.cloud(two))
    assert st.status == abi.TOO_FEW_CORR


def test_randomised_stress_against_oracle(reg, orc):
    """24 random registrations (cloud sizes, leaf, metric, levels, noise, initial offsets, non-finite points,
    batch composition all drawn from a seeded RNG): every per-iteration pose must match the oracle bit for bit.
    Exercises the certificate / worklist / cooperative-search chain on many different fill patterns."""
    rng = np.random.default_rng(2026)
    for case in range(6):
        metric = int(rng.integers(0, 2))
        two_levels = bool(rng.integers(0, 2))
        leaf = float(rng.choice([0.15, 0.2, 0.3]))
        p = _params(leaf=(2 * leaf, leaf) if two_levels else leaf, iterations=(6, 9) if two_levels else 12,
                    max_corr_dist=(4 * leaf, 2.5 * leaf) if two_levels else 2.5 * leaf, metric=metric, normal_leaf=max(0.4, 2 * leaf),
                    eps_rot=1e-6, eps_trans=1e-6)
        R = reg.Registrar(p)
        pairs, refs = [], []
        for k in range(4):
            n_az = int(rng.integers(150, 500))
            src, tgt, Tgt = synth.hdl32_pair(n_az, int(rng.integers(1, 10**6)), int(rng.integers(1, 10**6)), dx=float(rng.uniform(-0.3, 0.3)),
                                             dy=float(rng.uniform(-0.2, 0.2)), dyaw_deg=float(rng.uniform(-3, 3)),
                                             base=(float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3)), float(rng.uniform(-180, 180))))
            if k == 1:
                src = src.copy(); src[:: 41] = np.nan
            T0 = synth.perturb(Tgt, rng, 0.8, 0.08) if k % 2 else np.eye(4)
            cs, ct = R.clouds([src, tgt])
            pairs.append((cs, ct, T0))
            refs.append(orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt), T0, trace_cap=32))
        Tb, stb = R.align_batch(pairs)
        for k in range(4):
            assert np.array_equal(Tb[k], refs[k][0]), (case, k)
            _same_stats(stb[k], refs[k][1])
            T1, st1 = R.align(*pairs[k])
            assert np.array_equal(R.trace(), refs[k][2]), (case, k)


@pytest.mark.parametrize("copies", [9, 16])
def test_identical_pairs_in_one_batch_give_identical_results(reg, orc, copies):
    """The same scan pair `copies` times in one batch: every copy must reproduce the single registration (and the
    oracle) bit for bit. Cloud sizes are chosen so that the last 256-query block of a pair is nearly empty — the
    per-pair worklist segments must not overlap (regression: stride rounded to 64 instead of 256)."""
    p = _params(leaf=0.2, iterations=10, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5, eps_rot=0.0, eps_trans=0.0)
    R = reg.Registrar(p)
    src, tgt, Tgt = synth.hdl32_pair(400, 7, 8, dx=0.25, dy=-0.1, dyaw_deg=2.0)
    keep = (len(src) // 256) * 256 + 3          # 3 queries in the pair's last block
    src = src[:keep] if keep <= len(src) else src[: (len(src) // 256 - 1) * 256 + 3]
    cs, ct = R.clouds([src, tgt])
    assert cs.grid_info().n_valid % 256 == 3
    To, sto, _ = orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt))
    T1, st1 = R.align(cs, ct)
    assert np.array_equal(T1, To)
    Tb, stb = R.align_batch([(cs, ct, None)] * copies)
    for k in range(copies):
        assert np.array_equal(Tb[k], To), k
        _same_stats(stb[k], sto)


def test_clouds_bucketed_on_one_handle_register_on_another(reg, orc):
    """Bucketing no longer ends with a host synchronisation: a registration enqueued on ANOTHER handle's stream must order
    itself behind the bucketing of its clouds on the device (one event per bucketed batch)."""
    p = _params(leaf=0.2, iterations=8, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    A, B = reg.Registrar(p), reg.Registrar(p)
    for seed in range(4):
        src, tgt, Tgt = synth.hdl32_pair(600, 10 + seed, 20 + seed, dx=0.2, dy=0.1, dyaw_deg=2.0)
        cs, ct = A.clouds([src, tgt])          # returns while the bucketing is still in flight on A's stream
        T, st = B.align(cs, ct)                # B's stream waits for it on the device
        To, sto, _ = orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt))
        assert np.array_equal(T, To), seed
        _same_stats(st, sto)
        assert ct.grid_info().n_cells == orc.Cloud(p, tgt).grid_info().n_cells   # lazily fetched meta data


def test_async_bucketing_no_host_sync_and_device_side_errors(reg, orc):
    """m3dreg_cloud_create_batch_async: nothing of the bucketing is waited for — the grid geometry, the fixed-point exponents
    and the error state of every cloud reach the registration on the device (k_grid_params -> M3dLevelMeta -> k_patch_jobs).
    Good pairs must come out exactly as through the synchronous path and the oracle; a cloud without a finite point or with
    a grid beyond 31 key bits ends ITS registrations with BAD_CLOUD (pose = initial guess) and nobody else's."""
    p = _params(leaf=(0.5, 0.25), iterations=(4, 6), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    src, tgt, Tgt = synth.config1(5000)
    src2 = src.copy(); src2[::41] = np.nan                      # non-finite points: the finite count is a device-side quantity too
    empty = np.full((300, 3), np.nan, np.float32)
    huge = np.array([[0, 0, 0], [90000.0, 90000.0, 90000.0]], np.float32)   # 360 000 voxels per axis at leaf 0.25
    cl = R.clouds([src, tgt, src2, empty, huge], wait=False)
    T0 = synth.perturb(Tgt, np.random.default_rng(9), 1.0, 0.05)
    pairs = [(cl[0], cl[1], T0), (cl[3], cl[1], T0), (cl[2], cl[1], T0), (cl[0], cl[4], T0), (cl[0], cl[3], T0)]
    Ts, sts = R.align_batch(pairs)
    ot = orc.Cloud(p, tgt)
    for k, s in ((0, src), (2, src2)):
        To, sto, _ = orc.align(p, orc.Cloud(p, s), ot, T0)
        assert np.array_equal(Ts[k], To)
        _same_stats(sts[k], sto)
    for k in (1, 3, 4):
        assert sts[k].status == abi.BAD_CLOUD and sts[k].iterations == 0
        assert np.array_equal(Ts[k].astype(np.float32), T0.astype(np.float32))
    assert [c.status() for c in cl] == [0, 0, 0, abi.ERR_EMPTY_CLOUD, abi.ERR_GRID_TOO_LARGE]
    with pytest.raises(abi.M3dregError) as ei:
        cl[4].grid_info()
    assert ei.value.code == abi.ERR_GRID_TOO_LARGE
    _check_bucketing(cl[2], orc.Cloud(p, src2), 2)               # read back lazily, equal to the oracle's geometry bit for bit
    # the synchronous call still refuses such a batch up front
    with pytest.raises(abi.M3dregError) as ei:
        R.clouds([src, empty])
    assert ei.value.code == abi.ERR_EMPTY_CLOUD


def _crowded_cloud(seed, n_bg=5000):
    """A cloud with crowded voxels: a raster-ordered surface patch (hundreds of points per 10 cm voxel, input order sweeping it strip
    by strip, like a wall a metre from the sensor), a random-order blob, and a sparse background."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.linspace(0.0, 0.33, 70), np.linspace(0.0, 0.29, 55), indexing="ij")
    patch = np.stack([1.0 + u.ravel(), 0.5 + v.ravel(), 0.3 + 0.2 * u.ravel()], axis=1) + rng.normal(0, 0.002, (70 * 55, 3))
    blob = np.array([-1.0, 0.2, 0.7]) + rng.uniform(0, 0.21, (2500, 3))
    bg = synth.planes_cloud(n_bg, seed + 1, sigma=0.01, size=6.0)
    return np.concatenate([bg[: n_bg // 2], patch, blob, bg[n_bg // 2:]]).astype(np.float32)


@pytest.mark.parametrize("metric", [abi.POINT_TO_POINT, abi.POINT_TO_PLANE])
def test_crowded_voxels_chunk_boxes_stay_exact(reg, orc, metric):
    """Rows with more than M3D_LONG_ROW candidates are walked chunk by chunk, each chunk's exact box first (k_post_finalize): whatever
    the boxes let the search skip, every per-iteration pose still equals the oracle's exhaustive search bit for bit — raster-ordered
    points (nearly every chunk skipped), random-ordered points (hardly any), non-finite points in the cloud, seeded and certified
    later iterations."""
    tgt = _crowded_cloud(11)
    Tgt = synth.make_T(synth.rot_z(np.radians(1.5)) @ synth.rot_x(np.radians(-0.8)), np.array([0.03, -0.02, 0.015]))
    src = synth.apply_T(synth.inv_T(Tgt), _crowded_cloud(12).astype(np.float64)).astype(np.float32)
    src[::97] = np.nan
    tgt = tgt.copy(); tgt[5::211] = np.inf
    p = _params(leaf=0.1, iterations=8, max_corr_dist=0.3, metric=metric, normal_leaf=0.3)
    R = reg.Registrar(p)
    cs, ct = R.clouds([src, tgt])
    _check_bucketing(ct, orc.Cloud(p, tgt), 1)
    T1, st1 = R.align(cs, ct)
    T2, st2, tr2 = orc.align(p, orc.Cloud(p, src, omp=True), orc.Cloud(p, tgt, omp=True), trace_cap=16)
    assert np.array_equal(R.trace(), tr2) and np.array_equal(T1, T2)
    _same_stats(st1, st2)


@pytest.mark.parametrize("blob", [(30000, 0.12), (2600, 0.05)])   # many / few queries past the tiles: the reduction pass walks them one per lane / eight lanes per query
@pytest.mark.parametrize("lean", ["0", "1"])
def test_lean_and_full_correspondence_kernels_are_bit_identical(reg, orc, monkeypatch, lean, blob):
    """M3DREG_LEAN=1 (default): the tile iterations run k_nn_iter<true> (classify + bin only), and what it cannot bin is walked by the
    reduction pass's workgroups (k_accumulate_matches<.., true>: a workgroup with many pending queries one per lane, with few eight lanes per
    query); 0: the full k_nn_iter, which walks what it does not bin. A crowded pair (queries past the tiles) and an ordinary one in one batch,
    14 iterations (tiles, then fused late iterations), twice on one handle: same poses and statistics, equal to the oracle's."""
    monkeypatch.setenv("M3DREG_LEAN", lean)
    def blob_cloud(seed):   # a cube so full that some 20 cm bucket holds more points than a tile image (2048): its tile is flagged
        rng = np.random.default_rng(seed)
        return np.concatenate([_crowded_cloud(seed), np.array([2.0, -1.0, 0.4]) + rng.uniform(0, blob[1], (blob[0], 3))]).astype(np.float32)
    tgt_c = blob_cloud(21)
    Tc = synth.make_T(synth.rot_z(np.radians(1.0)), np.array([0.02, 0.01, -0.01]))
    src_c = synth.apply_T(synth.inv_T(Tc), blob_cloud(22).astype(np.float64)).astype(np.float32)
    src_o, tgt_o, _ = synth.hdl32_pair(600, 610, 611, dx=0.2, dy=0.05, dyaw_deg=1.5)
    p = _params(leaf=0.1, iterations=14, max_corr_dist=0.3, metric=abi.POINT_TO_PLANE, normal_leaf=0.3, eps_rot=0.0, eps_trans=0.0)
    R = reg.Registrar(p)
    cs_c, ct_c, cs_o, ct_o = R.clouds([src_c, tgt_c, src_o, tgt_o], source_only=[True, False, True, False])
    T, st = R.align_batch([(cs_c, ct_c, None), (cs_o, ct_o, None)])
    tile_searches, walked = R.counters()
    assert tile_searches > 0 and walked > 0   # (the flagged tile's queries were walked: by the reduction pass or by the full k_nn_iter)
    T2, st2 = R.align_batch([(cs_c, ct_c, None), (cs_o, ct_o, None)])   # (a handle's second batch is scheduled like its first)
    assert R.counters()[1] > 0
    for k, (s_, t_) in enumerate(((src_c, tgt_c), (src_o, tgt_o))):
        To, sto, _ = orc.align(p, orc.Cloud(p, s_, source_only=True, omp=True), orc.Cloud(p, t_, omp=True))
        assert np.array_equal(T[k], To) and np.array_equal(T2[k], To)
        _same_stats(st[k], sto)
        _same_stats(st2[k], sto)


@pytest.mark.parametrize("fuse_from", ["1", "8"])
def test_fused_launches_from_the_second_iteration_on(reg, orc, monkeypatch, fuse_from):
    """k_icp_late with its LDS worklist full: fuse_from 1 — the fused launch runs from the second iteration on, when nearly every query is uncertified and
    goes through the workgroup's worklist (up to 7 x 256 entries); 8: the shipped boundary. Same poses, traces and statistics as the oracle, bit for bit."""
    monkeypatch.setenv("M3DREG_TILE_ITERS", fuse_from)
    monkeypatch.setenv("M3DREG_FUSE_FROM", fuse_from)
    src, tgt, _ = synth.hdl32_pair(1500, 4100, 4101, dx=0.25, dy=-0.1, dyaw_deg=2.0)
    p = _params(leaf=0.1, iterations=12, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    R = reg.Registrar(p)
    cs, ct = R.clouds([src, tgt], source_only=[True, False])
    T1, st1 = R.align(cs, ct)
    T2, st2, tr2 = orc.align(p, orc.Cloud(p, src, omp=True, source_only=True), orc.Cloud(p, tgt, omp=True), trace_cap=16)
    assert np.array_equal(R.trace()[:12], tr2[:12]) and np.array_equal(T1, T2)
    _same_stats(st1, st2)


def test_batches_queued_behind_each_other_on_one_stream(reg, orc):
    """The bench's pipeline: several handles share ONE HIP stream, each holds a batch — bucketing (enqueue-only) and iterations of
    batch k+1 are queued behind batch k, nothing is waited for until every batch has been enqueued, and clouds go back to their
    handle's pool and are re-used while other batches are still in flight. Every pose must equal the oracle's, bit for bit."""
    import ctypes as C
    p = _params(leaf=0.2, iterations=7, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    first = reg.Registrar(p)
    stream = C.c_void_p(first.stream)
    handles = [first] + [reg.Registrar(p, stream=stream) for _ in range(2)]
    data, refs = [], []
    for k in range(6):
        src, tgt, Tgt = synth.hdl32_pair(300 + 60 * k, 40 + k, 50 + k, dx=0.15, dy=-0.1, dyaw_deg=1.5 + 0.3 * k)
        if k == 4:
            src = src.copy(); src[::29] = np.nan
        data.append((src, tgt))
        refs.append(orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt)))
    for rnd in range(2):                                       # the second round runs on recycled blocks
        pending = []
        for b in range(3):                                     # batch b = pairs 2b, 2b+1 on handle b; all three enqueued before any wait
            h = handles[b]
            cl = h.clouds([data[2 * b][0], data[2 * b][1], data[2 * b + 1][0], data[2 * b + 1][1]], wait=False)
            h.align_batch_async(h._pairs([(cl[0], cl[1], None), (cl[2], cl[3], None)]), 2)
            pending.append((h, cl))
        for b, (h, cl) in enumerate(pending):
            T, st = h.batch_wait(2)
            for j in range(2):
                assert np.array_equal(T[j], refs[2 * b + j][0]), (rnd, b, j)
                _same_stats(st[j], refs[2 * b + j][1])
            for c in cl:
                c.free()


def test_a_handle_holds_one_batch(reg):
    """m3dreg_align_batch_async twice without m3dreg_batch_wait would overwrite the first batch's pinned descriptors in flight: refused."""
    p = _params(leaf=0.25, iterations=3, max_corr_dist=0.5, metric=abi.POINT_TO_POINT)
    R = reg.Registrar(p)
    src, tgt, _ = synth.config1(3000)
    cs, ct = R.clouds([src, tgt])
    arr = R._pairs([(cs, ct, None)])
    R.align_batch_async(arr, 1)
    with pytest.raises(abi.M3dregError) as ei:
        R.align_batch_async(arr, 1)
    assert ei.value.code == abi.ERR_INVALID_ARG
    T, st = R.batch_wait(1)
    R.align_batch_async(arr, 1)
    T2, _ = R.batch_wait(1)
    assert np.array_equal(T, T2)


def test_config4_shard_full_size_properties(reg):
    """BASELINE config 4 at full size, one GPU's shard (8 pairs x 100 000 rays, clouds as the aggregator publishes them): properties that
    need no oracle run — the batch equals the eight single registrations bit for bit, in any batch order, through the enqueue-only
    bucketing as well; every pose lands on the generator's ground truth."""
    p = _params(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    R = reg.Registrar(p)
    data = [synth.config4_pair(k) for k in range(8)]
    flat = [c for src, tgt, _ in data for c in (src, tgt)]
    cl = R.clouds(flat, wait=False)
    pairs = [(cl[2 * k], cl[2 * k + 1], None) for k in range(8)]
    Tb, stb = R.align_batch(pairs)
    Tr, _ = R.align_batch(pairs[::-1])
    assert np.array_equal(Tb, Tr[::-1])                                     # batch composition / order never changes a bit
    for k in (0, 4, 7):                                                     # pair 4 stands next to an obstacle (crowded voxels)
        cs, ct = R.clouds([data[k][0], data[k][1]])                         # synchronous bucketing, registered alone
        T1, st1 = R.align(cs, ct)
        assert np.array_equal(T1, Tb[k]) and st1.n_corr == stb[k].n_corr, k
    for k in range(8):
        rot, tra = synth.pose_error(Tb[k], data[k][2])
        assert stb[k].status == abi.MAX_ITERATIONS and rot < 0.1 and tra < 0.006, (k, rot, tra)


def _config4_shards(world=8, per_rank=8):
    """the shards `bench.py --gpus 8` forms: LPT over the tabulated a-priori costs, 8 pairs per rank (bench.py main(), --shard lpt)"""
    import json, os
    from mandala_mapping_amd import sharding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    costs, _ = sharding.table_costs(json.load(open(os.path.join(root, "mandala_mapping_amd", "config4_costs.json"))), world * per_rank)   # (the measured costs where the table has them)
    return sharding.lpt_assign(costs, world, capacity=per_rank)


def test_config4_all_shards(reg):
    """BASELINE config 4 as a whole: all 64 loop-closure pairs, in the eight LPT shards `bench.py --gpus 8` would hand to its ranks, one shard
    at a time on this GPU (8 pairs x 100 000 rays per batch, the bench's parameters). Every pose lands inside BASELINE.md §3's bar, every
    registration runs its 20 iterations, and for one pair of every shard — its most crowded one, the pair that sets the shard's time — the
    batch result equals the single registration bit for bit."""
    import json, os
    p = _params(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    R = reg.Registrar(p)
    shards = _config4_shards()
    assert sorted(k for s in shards for k in s) == list(range(64)) and all(len(s) == 8 for s in shards)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    costs = json.load(open(os.path.join(root, "mandala_mapping_amd", "config4_costs.json")))["costs"]
    worst = (0.0, 0.0)
    for shard in shards:
        data = [synth.config4_pair(k) for k in shard]
        cl = R.clouds([c for src, tgt, _ in data for c in (src, tgt)], wait=False, source_only=[True, False] * 8)
        Tb, stb = R.align_batch([(cl[2 * j], cl[2 * j + 1], None) for j in range(8)])
        for j, k in enumerate(shard):
            rot, tra = synth.pose_error(Tb[j], data[j][2])
            assert stb[j].status == abi.MAX_ITERATIONS and stb[j].iterations == 20 and rot < 0.1 and tra < 0.006, (k, rot, tra, stb[j].as_dict())
            worst = (max(worst[0], rot), max(worst[1], tra))
        j = max(range(8), key=lambda j_: costs[shard[j_]])
        cs, ct = R.clouds([data[j][0], data[j][1]], source_only=[True, False])
        T1, st1 = R.align(cs, ct)
        assert np.array_equal(T1, Tb[j]) and st1.n_corr == stb[j].n_corr, shard[j]
        for c in list(cl) + [cs, ct]:
            c.free()
    print(f"config 4, 64 pairs: worst rotation error {worst[0]:.4f} deg, worst translation error {1e3 * worst[1]:.2f} mm")


@pytest.mark.parametrize("which", ["config3", "config4_pair2", "config4_pair31"])
def test_full_size_pairs_equal_the_oracle(reg, orc, which):
    """BASELINE configs 3 and 4 at FULL size (100 000 rays per sweep) against the oracle, bit for bit: the single config-3 pair, an ordinary
    config-4 pair (2: the lowest a-priori cost of the 64) and the most crowded one (31: 1.3 m from an obstacle, the highest cost) — bucketing of the
    target, every per-iteration pose, the final pose and the statistics. The OpenMP build of the oracle takes about a second per pair."""
    src, tgt, Tgt = synth.config3() if which == "config3" else synth.config4_pair(int(which[len("config4_pair"):]))
    p = _params(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    R = reg.Registrar(p)
    cs, ct = R.clouds([src, tgt], source_only=[True, False])
    os_, ot = orc.Cloud(p, src, omp=True, source_only=True), orc.Cloud(p, tgt, omp=True)
    _check_bucketing(ct, ot, 1)
    T1, st1 = R.align(cs, ct)
    T2, st2, tr2 = orc.align(p, os_, ot, trace_cap=32)
    assert np.array_equal(R.trace(), tr2) and np.array_equal(T1, T2)
    _same_stats(st1, st2)
    rot, tra = synth.pose_error(T1, Tgt)
    assert rot < 0.1 and tra < 0.006, (rot, tra)


@pytest.mark.parametrize("metric", [abi.POINT_TO_POINT, abi.POINT_TO_PLANE])
def test_pose_ring_wraps_within_one_level(reg, orc, monkeypatch, metric):
    """The 8-byte per-query record keeps the iteration of a query's last real search modulo 32 and recomputes where the query was from the pair's pose
    ring (icp.hip: m3d_cert_state). 75 fixed iterations on ONE level: queries that stay certified for 32 iterations meet their own slot again — the record is
    then stale by construction and the query must search — twice over. Every per-iteration pose equals the oracle's, with the fused late launches
    and without them, through the tiles and through the global walk."""
    src, tgt, _ = synth.hdl32_pair(500, 71, 72, dx=0.2, dy=0.05, dyaw_deg=1.5)
    p = _params(leaf=0.15, iterations=75, max_corr_dist=0.5, metric=metric, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    T2, st2, tr2 = orc.align(p, orc.Cloud(p, src, omp=True, source_only=True), orc.Cloud(p, tgt, omp=True), trace_cap=128)
    assert st2.iterations == 75
    for env in ({}, {"M3DREG_FUSE_FROM": "0"}, {"M3DREG_TILES": "0", "M3DREG_LEAN": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        R = reg.Registrar(p)
        cs, ct = R.clouds([src, tgt], source_only=[True, False])
        T1, st1 = R.align(cs, ct)
        assert np.array_equal(R.trace(), tr2) and np.array_equal(T1, T2), env
        _same_stats(st1, st2)
        for k in env:
            monkeypatch.delenv(k)


def test_device_density_is_the_lpt_cost(reg):
    """m3dreg_cloud_density (the bucketing pipeline's sum of squared voxel populations / finite points) equals synth.crowdedness — the numpy estimate the
    LPT sharding of config 4 was tabulated with (mandala_mapping_amd/config4_costs.json) — for full targets and source-only clouds alike, and the table's
    entries are what the device says."""
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    costs = json.load(open(os.path.join(root, "mandala_mapping_amd", "config4_costs.json")))["costs"]
    p = _params(leaf=0.1, iterations=2, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    R = reg.Registrar(p)
    for k in (4, 31, 50):
        src, tgt, _ = synth.config4_pair(k)
        src = src.copy(); src[::997] = np.nan
        cs, ct = R.clouds([src, tgt], source_only=[True, False])
        ds, dt = cs.density(), ct.density()
        assert abs(ds - synth.crowdedness(src)) < 1e-3 * ds and abs(dt - synth.crowdedness(tgt)) < 1e-3 * dt, (k, ds, dt)
        assert abs(ds + dt - costs[k]) < 0.02 * costs[k], (k, ds + dt, costs[k])   # (the table was made from clouds without the NaNs)
    two = _params(leaf=(0.4, 0.1), iterations=(2, 2), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    c2 = reg.Registrar(two).cloud(tgt)
    assert abs(c2.density(0) - synth.crowdedness(tgt, 0.4)) < 1e-3 * c2.density(0) and abs(c2.density() - synth.crowdedness(tgt)) < 1e-3 * c2.density()


def test_source_only_clouds_skip_the_normal_grid(reg, orc):
    """m3dreg_cloud_desc.source_only: a cloud that will only ever be a source is sorted but gets no normals (its normal-estimation grid
    is not built). Registering it gives exactly the oracle's poses; as a point-to-plane TARGET it is refused."""
    p = _params(leaf=(0.4, 0.2), iterations=(6, 8), max_corr_dist=(1.0, 0.5), metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    R = reg.Registrar(p)
    src, tgt, Tgt = synth.hdl32_pair(700, 61, 62, dx=0.25, dy=0.1, dyaw_deg=2.5)
    src = src.copy(); src[::37] = np.nan
    T0 = synth.perturb(Tgt, np.random.default_rng(2), 0.8, 0.06)
    To, sto, tro = orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt), T0, trace_cap=32)
    for wait in (True, False):
        cs, ct = R.clouds([src, tgt], wait=wait, source_only=[True, False])
        T, st = R.align(cs, ct, T0)
        assert np.array_equal(R.trace(), tro) and np.array_equal(T, To)
        _same_stats(st, sto)
        assert cs.grid_info(1).has_normals == 0 and ct.grid_info(1).has_normals == 1
        e, eo = cs.export(1), orc.Cloud(p, src).export(1)
        assert np.array_equal(e["perm"], eo["perm"]) and np.array_equal(e["sorted_xyz"], eo["sorted_xyz"], equal_nan=True)
        with pytest.raises(abi.M3dregError) as ei:
            R.align(ct, cs, T0)
        assert ei.value.code == abi.ERR_LEVEL_MISMATCH
        with pytest.raises(abi.M3dregError):
            cs.nn(src[:10], 0.5, level=1)                   # no bucket table to search
