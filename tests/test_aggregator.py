"""SURVEY.md §8 row f1 — the aggregation step (m3d_aggregator.cpp:53-88, 231-288).

CPU suite: the oracle restatement against an independent numpy restatement of the same reference lines.
GPU suite: the HIP aggregator against the oracle — bit-exact for the PointCloud2 path (double-precision rigid
transform, box filter, stable order, angular distance, progress, ready); the LaserScan path uses cosf/sinf,
whose device implementation may differ from glibc's in the last ulp, so its coordinates are compared to 2 ulp
(documented tolerance; everything downstream of the trig is exact)."""
import numpy as np
import pytest

from mandala_mapping_amd import abi, pointcloud2 as pc2, synth

BBOX = (1.0, -1.0, 1.0, -1.0, 1.0, -1.0)   # node defaults, m3d_aggregator.cpp:164-171


def quat_z(a):
    return [0.0, 0.0, np.sin(a / 2), np.cos(a / 2)]


def head_tf(k, n_msgs):
    """rotating head: yaw advances by 1.2*pi over the sweep, small fixed lever arm"""
    a = 1.2 * np.pi * k / (n_msgs - 1)
    return [0.05, -0.02, 0.3] + quat_z(a)


def numpy_transform_filter(xyz, tf7):
    x, y, z, w = tf7[3:]
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    M = np.array([[1 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1 - (xx + yy)]])
    p = xyz.astype(np.float64)
    p1 = np.stack([M[r, 0] * p[:, 0] + M[r, 1] * p[:, 1] + M[r, 2] * p[:, 2] + tf7[r] for r in range(3)], 1)
    pp = p1.astype(np.float32)
    keep = (pp[:, 0] > 1) | (pp[:, 0] < -1) | (pp[:, 1] > 1) | (pp[:, 1] < -1) | (pp[:, 2] > 1) | (pp[:, 2] < -1)
    return pp[keep]


def messages(n_msgs=12, pts=1500, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_msgs):
        xyz = rng.uniform(-3, 3, size=(pts + 17 * k, 3)).astype(np.float32)   # many points fall inside the +-1 m box
        out.append((pc2.encode_xyz(xyz), xyz, head_tf(k, n_msgs)))
    return out


def test_oracle_follows_reference_lines(orc):
    agg = orc.Aggregator(BBOX)
    expect = []
    for msg, xyz, tf7 in messages():
        agg.add_cloud(msg, tf7)
        expect.append(numpy_transform_filter(xyz, tf7))
    expect = np.concatenate(expect)
    got = agg.points()
    assert got.shape[0] == expect.shape[0] and 0 < got.shape[0] < sum(m[0].n for m in messages())
    assert np.array_equal(got[:, :3].view(np.uint32), expect.view(np.uint32)) and (got[:, 3] == 0).all()
    st = agg.status()
    assert abs(st["angle"] - 1.2 * np.pi) < 1e-3 and st["ready"]            # 1.2*pi > 1.1*pi (:30,:98)
    assert st["progress"] == 0.1 * np.floor(st["angle"] * 1000.0 / (1.1 * np.pi))
    agg.restart()
    assert agg.status() == {"progress": 0.0, "ready": False, "angle": 0.0, "n": 0}


def test_oracle_progress_not_ready_before_threshold(orc):
    agg = orc.Aggregator(BBOX)
    msgs = messages()
    for msg, _, tf7 in msgs[:6]:
        agg.add_cloud(msg, tf7)
    st = agg.status()
    assert not st["ready"] and 0 < st["progress"] < 100


@pytest.mark.gpu
def test_hip_aggregator_pointcloud_path_bit_exact(reg, orc):
    R = reg.Registrar(abi.Params.make(leaf=0.25, iterations=5, metric=abi.POINT_TO_POINT))
    a, o = reg.Aggregator(R, BBOX, capacity=200000), orc.Aggregator(BBOX)
    for msg, _, tf7 in messages(n_msgs=14, pts=5000, seed=3):
        a.add_cloud(msg, tf7)
        o.add_cloud(msg, tf7)
        sa, so = a.status(), o.status()
        assert sa == so, (sa, so)                                            # count, angle (same double), progress, ready
    assert np.array_equal(a.points().view(np.uint32), o.points().view(np.uint32))
    assert a.status()["ready"]
    # 32-byte points with shifted fields
    a.restart(); o.restart()
    xyz = synth.planes_cloud(3000, 5) - np.float32(3.0)
    m = pc2.encode_xyz(xyz, point_step=32, offsets=(4, 12, 20))
    a.add_cloud(m, head_tf(1, 5)); o.add_cloud(m, head_tf(1, 5))
    assert np.array_equal(a.points().view(np.uint32), o.points().view(np.uint32))


def _scan_messages():
    rng = np.random.default_rng(1)
    return [(rng.uniform(0.3, 25.0, size=1081).astype(np.float32), head_tf(k, 20)) for k in range(20)]   # SICK LMS: 270 deg / 0.25 deg


def test_oracle_laserscan_trig_readings_differ_in_the_last_bit(orc):
    """`cos(ang)*dist` (m3d_aggregator.cpp:281-282): the double reading (default) and the float-overload reading are two different
    restatements — about a third of the coordinates differ, by one float ulp of the coordinate."""
    a, b = orc.Aggregator(BBOX), orc.Aggregator(BBOX)
    for ranges, tf7 in _scan_messages()[:4]:
        a.add_scan(ranges, np.float32(-2.3561945), np.float32(0.004363323), tf7)
        b.add_scan(ranges, np.float32(-2.3561945), np.float32(0.004363323), tf7, float_overload=True)
    pa, pb = a.points(), b.points()
    assert len(pa) == len(pb) and 0.05 < (pa != pb).any(axis=1).mean() < 0.95
    assert np.abs(pa - pb).max() <= 4 * 1.2e-7 * 25.0 * 1.5


@pytest.mark.gpu
def test_hip_aggregator_laserscan_path(reg, orc):
    """The LaserScan path under the DEFAULT reading of m3d_aggregator.cpp:281-282 (double cos / sin, the product rounded once into the float
    field: m3d_agg_oracle.c): bit-exact against the oracle — the device's and glibc's double cos / sin may differ in their last double bit, which
    the rounding to float absorbs."""
    R = reg.Registrar(abi.Params.make(leaf=0.25, iterations=5, metric=abi.POINT_TO_POINT))
    a, o = reg.Aggregator(R, BBOX, capacity=100000), orc.Aggregator(BBOX)
    for ranges, tf7 in _scan_messages():
        a.add_scan(ranges, np.float32(-2.3561945), np.float32(0.004363323), tf7)
        o.add_scan(ranges, np.float32(-2.3561945), np.float32(0.004363323), tf7)
    assert a.status() == o.status()
    pa, po = a.points(), o.points()
    assert len(pa) > 15000 and np.array_equal(pa.view(np.uint32), po.view(np.uint32))


def test_specified_float_trig_is_within_one_ulp_of_the_c_library_and_of_the_truth(orc):
    """Spec §Trig (m3d_sincosf_spec / orc_sincosf_spec): the float sine / cosine the float-overload reading uses — against float64 numpy (the truth, rounded to float:
    at most 1 ulp away, and equal for all but a handful in 10^5) and against glibc's sinf / cosf through numpy's float32 ufuncs (at most 1 ulp). Quadrant logic: a sweep
    through +-1000 rad, exact multiples of pi/2 as floats, zero, denormals, huge and non-finite arguments."""
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-7, 7, 60000), rng.uniform(-1000, 1000, 40000), np.arange(-64, 65) * (np.pi / 2), [0.0, -0.0, 1e-40, -1e-40, 1e-20, 3e5, -3e5]]).astype(np.float32)
    s, c = orc.sincosf_spec(x)
    ts, tc = np.sin(x.astype(np.float64)), np.cos(x.astype(np.float64))
    for got, true, lib_ in ((s, ts, np.sin(x)), (c, tc, np.cos(x))):
        ulp = np.maximum(np.abs(np.spacing(true.astype(np.float32))), np.float32(1e-45)).astype(np.float64)
        assert (np.abs(got.astype(np.float64) - true) <= 1.0 * ulp).all()
        assert (got != true.astype(np.float32)).mean() < 1e-3                       # correctly rounded nearly always
        assert (np.abs(got.astype(np.float64) - lib_.astype(np.float64)) <= ulp).all()
    for bad in (np.inf, -np.inf, np.nan):
        sb, cb = orc.sincosf_spec(np.float32(bad))
        assert np.isnan(sb[0]) and np.isnan(cb[0])


@pytest.mark.gpu
def test_hip_aggregator_laserscan_path_float_overload(reg, orc):
    """The other reading (m3dagg_set_scan_trig(1): cos / sin in float — GCC >= 6 with the C++ <math.h> wrapper in sight), BIT-EXACT since round 6: the float sine /
    cosine of that reading are specified (Spec §Trig: double reduction + kernels, one rounding), the device and the oracle evaluate the same arithmetic. Side check: the
    C library's own cosf / sinf (oracle mode 2) give the same sweep to 1 ulp of the unit vector."""
    R = reg.Registrar(abi.Params.make(leaf=0.25, iterations=5, metric=abi.POINT_TO_POINT))
    a, o, o2 = reg.Aggregator(R, BBOX, capacity=100000), orc.Aggregator(BBOX), orc.Aggregator(BBOX)
    a.set_scan_trig(True)
    for ranges, tf7 in _scan_messages():
        a.add_scan(ranges, np.float32(-2.3561945), np.float32(0.004363323), tf7)
        o.add_scan(ranges, np.float32(-2.3561945), np.float32(0.004363323), tf7, float_overload=1)
        o2.add_scan(ranges, np.float32(-2.3561945), np.float32(0.004363323), tf7, float_overload=2)
    assert a.status() == o.status()
    pa, po, p2 = a.points(), o.points(), o2.points()
    assert len(pa) > 15000 and np.array_equal(pa.view(np.uint32), po.view(np.uint32))
    if len(p2) == len(po):     # (a point within 1 ulp of a box face may flip between the specified functions and glibc's)
        assert np.abs(p2 - po).max() <= 2 * 1.2e-7 * 25.0 * 1.5


@pytest.mark.gpu
def test_aggregated_sweep_registers_without_leaving_the_device(reg, orc):
    """aggregate two sweeps from packet-sized messages, take them as clouds, register: same pose as registering
    the same points handed over as PointCloud2."""
    src, tgt, Tgt = synth.hdl32_pair(600, 7, 8, dx=0.2, dy=0.05, dyaw_deg=1.5)
    p = abi.Params.make(leaf=0.2, iterations=15, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    ident = [0, 0, 0, 0, 0, 0, 1.0]
    clouds = []
    for cloud in (src, tgt):
        a = reg.Aggregator(R, (0.3, -0.3, 0.3, -0.3, 0.3, -0.3), capacity=len(cloud) + 16)
        for chunk in np.array_split(cloud, 9):
            a.add_cloud(pc2.encode_xyz(chunk), ident)
        kept = a.points()[:, :3].copy()
        clouds.append((a.take_cloud(), kept))
        assert a.status()["n"] == 0                                         # cleared after publish (:212)
    T1, st1 = R.align(clouds[0][0], clouds[1][0])
    T2, st2 = R.align(R.cloud(clouds[0][1]), R.cloud(clouds[1][1]))
    assert np.array_equal(T1, T2) and st1.n_corr == st2.n_corr
    rot, tra = synth.pose_error(T1, Tgt)
    assert rot < 0.1 and tra < 0.03
