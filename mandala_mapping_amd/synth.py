"""Seeded synthetic workloads for BASELINE.json `configs`.

No dataset ships with the reference and the GPU box has no network, so every cloud is generated
from a seed by the functions below; the same numpy version runs here and on the GPU box, so both
see identical bytes. All clouds are returned as float32 [n,3] in the SENSOR frame of their scan,
the way a lidar driver (and m3d_aggregator, whose output frame is the unit's m3d_link) hands them on.

Pose convention: T_gt maps SOURCE-frame points into the TARGET frame (what m3dreg_align returns).
"""
import numpy as np


# ------------------------------------------------------------------------------------------------
# SE(3) helpers (float64)
# ------------------------------------------------------------------------------------------------
def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def make_T(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def inv_T(T):
    R, t = T[:3, :3], T[:3, 3]
    return make_T(R.T, -R.T @ t)


def apply_T(T, xyz):
    return (xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3])


def pose_error(T_est, T_gt):
    """(rotation error [deg], translation error [m]) of T_est against T_gt."""
    D = inv_T(np.asarray(T_gt, dtype=np.float64)) @ np.asarray(T_est, dtype=np.float64)
    c = np.clip((np.trace(D[:3, :3]) - 1.0) * 0.5, -1.0, 1.0)
    return float(np.degrees(np.arccos(c))), float(np.linalg.norm(D[:3, 3]))


def random_T(rng, max_deg, max_trans):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.radians(rng.uniform(0.3 * max_deg, max_deg))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = rng.normal(size=3)
    t *= rng.uniform(0.3 * max_trans, max_trans) / np.linalg.norm(t)
    return make_T(R, t)


# ------------------------------------------------------------------------------------------------
# config 1: three mutually orthogonal 10 m x 10 m planes, 10 k points, sigma = 1 cm
# ------------------------------------------------------------------------------------------------
def planes_cloud(n, seed, sigma=0.01, size=10.0):
    rng = np.random.default_rng(seed)
    counts = [n - 2 * (n // 3), n // 3, n // 3]  # 3334/3333/3333 at n = 10 000
    parts = []
    for axis, m in enumerate(counts):
        uv = rng.uniform(0.0, size, size=(m, 2))
        w = rng.normal(0.0, sigma, size=m)
        p = np.empty((m, 3))
        others = [a for a in range(3) if a != (2, 0, 1)[axis]]
        p[:, (2, 0, 1)[axis]] = w  # planes z=0, x=0, y=0
        p[:, others[0]] = uv[:, 0]
        p[:, others[1]] = uv[:, 1]
        parts.append(p)
    return np.concatenate(parts).astype(np.float32)


def config1_T_gt():
    R = rot_z(np.radians(3.0)) @ rot_y(np.radians(-1.5)) @ rot_x(np.radians(2.0))
    return make_T(R, np.array([0.10, -0.05, 0.08]))


def config1(n=10000):
    """(source, target, T_gt): target = planes(seed 42); source = independent resample (seed 43)
    expressed in a frame displaced by T_gt, i.e. source = T_gt^-1 * resample."""
    tgt = planes_cloud(n, 42)
    res = planes_cloud(n, 43)
    T = config1_T_gt()
    src = apply_T(inv_T(T), res).astype(np.float32)
    return src, tgt, T


# ------------------------------------------------------------------------------------------------
# configs 2-5: synthetic Velodyne HDL-32E in a box room with box obstacles (analytic ray casting)
# ------------------------------------------------------------------------------------------------
ROOM = (np.array([-20.0, -15.0, 0.0]), np.array([20.0, 15.0, 6.0]))  # 40 x 30 x 6 m
OBSTACLES = [  # (min, max) axis-aligned boxes standing on the floor
    (np.array([4.0, 3.0, 0.0]), np.array([6.5, 5.0, 2.2])),
    (np.array([-9.0, -6.0, 0.0]), np.array([-6.0, -4.5, 1.6])),
    (np.array([10.0, -10.0, 0.0]), np.array([13.0, -7.0, 3.0])),
    (np.array([-14.0, 6.0, 0.0]), np.array([-11.5, 10.0, 2.6])),
    (np.array([1.0, -9.0, 0.0]), np.array([2.5, -7.5, 1.2])),
    (np.array([-3.0, 8.0, 0.0]), np.array([0.5, 9.5, 4.0])),
]
HDL32_ELEV_DEG = 10.67 - (4.0 / 3.0) * np.arange(32)  # +10.67 ... -30.67 deg
SENSOR_HEIGHT = 1.8


def _ray_box_exit(o, d, bmin, bmax):
    """distance along rays (origin inside the box) to the box surface."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (bmin - o) / d
        t2 = (bmax - o) / d
    tfar = np.nanmin(np.maximum(t1, t2), axis=1)
    return tfar


def _ray_box_enter(o, d, bmin, bmax):
    """distance to the first hit of a box seen from outside (inf when missed)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (bmin - o) / d
        t2 = (bmax - o) / d
    tn = np.nanmax(np.minimum(t1, t2), axis=1)
    tf = np.nanmin(np.maximum(t1, t2), axis=1)
    hit = (tn <= tf) & (tn > 0.0)
    return np.where(hit, tn, np.inf)


def hdl32_scan(pose, n_azimuth, seed, sigma=0.02, rmin=0.4, rmax=100.0, self_filter=None):
    """One sweep of a 32-beam lidar at world pose `pose` (4x4, sensor -> world); returns points in the
    SENSOR frame, firing order (azimuth-major, 32 beams per azimuth step).
    self_filter (metres or None): the aggregator's self-filter box — m3d_aggregator keeps a point only when at least one
    of its coordinates lies outside [-box, +box] (m3d_aggregator.cpp:65-73, node defaults +-1 m at :164-171), so the clouds it
    publishes never hold the returns of whatever stands within a metre of the unit."""
    rng = np.random.default_rng(seed)
    az = (2.0 * np.pi / n_azimuth) * np.arange(n_azimuth)
    el = np.radians(HDL32_ELEV_DEG)
    ca, sa = np.cos(az)[:, None], np.sin(az)[:, None]
    ce, se = np.cos(el)[None, :], np.sin(el)[None, :]
    d_s = np.stack([ca * ce, sa * ce, np.broadcast_to(se, (n_azimuth, 32))], axis=-1).reshape(-1, 3)
    R, o = pose[:3, :3], pose[:3, 3]
    d_w = d_s @ R.T
    o_w = np.broadcast_to(o, d_w.shape)
    t = _ray_box_exit(o_w, d_w, *ROOM)
    for bmin, bmax in OBSTACLES:
        t = np.minimum(t, _ray_box_enter(o_w, d_w, bmin, bmax))
    r = t + rng.normal(0.0, sigma, size=t.shape)
    keep = (r >= rmin) & (r <= rmax) & np.isfinite(r)
    pts = (d_s[keep] * r[keep, None]).astype(np.float32)
    if self_filter is not None:
        pts = pts[(np.abs(pts) > np.float32(self_filter)).any(axis=1)]
    return pts


def sensor_pose(x, y, yaw_deg, z=SENSOR_HEIGHT):
    return make_T(rot_z(np.radians(yaw_deg)), np.array([x, y, z]))


def hdl32_pair(n_azimuth, seed_tgt, seed_src, dx=0.5, dy=0.1, dyaw_deg=3.0, base=(0.0, 0.0, 0.0), self_filter=None):
    """(source, target, T_gt) for two sweeps taken at poses P1 (target) and P2 (source)."""
    P1 = sensor_pose(base[0], base[1], base[2])
    P2 = P1 @ make_T(rot_z(np.radians(dyaw_deg)), np.array([dx, dy, 0.0]))
    tgt = hdl32_scan(P1, n_azimuth, seed_tgt, self_filter=self_filter)
    src = hdl32_scan(P2, n_azimuth, seed_src, self_filter=self_filter)
    return src, tgt, inv_T(P1) @ P2


def config2():
    """32 x 2188 = 70 016 rays per sweep."""
    return hdl32_pair(2188, 100, 101)


def config3():
    """32 x 3125 = 100 000 rays per sweep."""
    return hdl32_pair(3125, 100, 101)


def config4_pair(k, n_azimuth=3125, self_filter=1.0):
    """k-th loop-closure candidate pair of config 4 (seeds 1000+k): random base pose in the room,
    random relative motion <= 5 deg / <= 0.5 m (yaw + planar translation dominate, as for a ground robot).
    The clouds are what m3d_aggregator would publish from these sweeps: its +-1 m self-filter box is applied (hdl32_scan).
    The poses are random, a few stand half a metre from a wall: unfiltered, such a sweep puts a thousand points into one
    10 cm voxel and a third of the cloud into voxels of more than 32 (self_filter=None reproduces that; tests and
    scripts/crowded_bench.py keep exercising it)."""
    rng = np.random.default_rng(1000 + k)
    base = (rng.uniform(-6.0, 6.0), rng.uniform(-4.0, 4.0), rng.uniform(-180.0, 180.0))
    dyaw = rng.uniform(-5.0, 5.0)
    dx, dy = rng.uniform(-0.35, 0.35, size=2)
    return hdl32_pair(n_azimuth, 2000 + 2 * k, 2001 + 2 * k, dx=dx, dy=dy, dyaw_deg=dyaw, base=base, self_filter=self_filter)


def crowdedness(xyz, leaf=0.1):
    """Mean population of the voxel a point of the cloud lies in (sum of squared voxel counts / points): what a query meets in its
    home voxel. Single registrations of the config-4 pairs take 0.72 ... 1.40 ms on an MI355X and this number explains it
    (correlation 0.9 over 19 measured pairs): the cost estimate the bench shards the pairs by."""
    c = xyz[np.isfinite(xyz).all(axis=1)]
    v = np.floor((c - c.min(0)) / np.float32(leaf)).astype(np.int64)
    _, cnt = np.unique(v[:, 0] + 8192 * (v[:, 1] + 8192 * v[:, 2]), return_counts=True)
    return float((cnt.astype(np.float64) ** 2).sum() / max(1, len(c)))


def config5(n_scans=22, n_azimuth=3125, dedup=0.01):
    """(live scan, map, T_gt, T_init): map = n_scans sweeps along a 10 m straight trajectory merged in the
    frame of the first sweep and de-duplicated on a `dedup` grid — 2 066 481 points at the defaults (BASELINE config 5 says
    2 M: rounds 1-3 merged 20 sweeps on a 2 cm grid, which left 1.51 M; 20 sweeps hold 2.0 M returns before any
    de-duplication, so the map takes 22 sweeps on a 1 cm grid); live scan =
    one more sweep 0.3 m / 0.2 m / 2 deg off the trajectory's midpoint; T_init = the midpoint pose itself (the
    odometry prior a map-based localiser starts from)."""
    P0 = sensor_pose(-5.0, 0.0, 0.0)
    parts = []
    for i in range(n_scans):
        Pi = sensor_pose(-5.0 + 10.0 * i / max(1, n_scans - 1), 0.0, 0.0)
        s = hdl32_scan(Pi, n_azimuth, 500 + i)
        parts.append(apply_T(inv_T(P0) @ Pi, s))
    m = np.concatenate(parts)
    if dedup > 0:
        q = np.floor(m / dedup).astype(np.int64)
        key = (q[:, 0] + 4096) + ((q[:, 1] + 4096) << 14) + ((q[:, 2] + 4096) << 28)
        _, first = np.unique(key, return_index=True)
        m = m[np.sort(first)]
    Pl = sensor_pose(0.3, 0.2, 2.0)
    live = hdl32_scan(Pl, n_azimuth, 777)
    return live, m.astype(np.float32), inv_T(P0) @ Pl, inv_T(P0) @ sensor_pose(0.0, 0.0, 0.0)


def perturb(T, rng, deg, trans):
    """A noisy initial guess around T (odometry prior)."""
    return T @ random_T(rng, deg, trans)


# ------------------------------------------------------------------------------------------------
# a calibration sweep of the rotating 2-D laser unit (m3d_calibration nodes)
# ------------------------------------------------------------------------------------------------
def offset_matrix(params):
    """laserOffsetMatrix of m3d_calibration_twiddle.cpp:202-220 in float64: rotate(Rx(yaw) Ry(pitch) Rz(roll)) then
    translate((x, y, z)) in the rotated frame. params = (x, y, z, yaw, pitch, roll)."""
    x, y, z, yaw, pitch, roll = [float(v) for v in params]
    R = rot_x(yaw) @ rot_y(pitch) @ rot_z(roll)
    return make_T(R, R @ np.array([x, y, z]))


CAL_ROOM = (np.array([-3.2, -2.6, 0.0]), np.array([3.0, 2.8, 3.0]))   # a lab-sized room: ranges of 1.3 - 4 m
CAL_OBSTACLES = [
    (np.array([1.6, 1.2, 0.0]), np.array([2.4, 2.0, 1.1])),
    (np.array([-2.6, -2.0, 0.0]), np.array([-1.9, -0.8, 1.7])),
]


def calibration_sweep(n_seg=480, n_rays=480, true_params=(0.0, 0.03, -0.02, 0.03, 0.0, 0.02), seed=7, sigma=0.003, turns=1.0):
    """Segments of one calibration sweep: a planar laser (rays in its own XY plane) on a head that turns about the
    head's X axis — the axis lies in the scan plane, so over a full turn every surface is seen twice, once by the
    rays with y > 0 and once by those with y <= 0 (laserUpAxis = 1). The laser sits on the head with the mounting
    offset `true_params`; the tf of every message (`original_Transform`) only knows the head angle. The room is small
    enough for neighbouring scan planes to land closer than the 0.05 m radius of the cost function.
    Returns [(points float32 [k,3] in the laser frame, original_T float32 4x4)]."""
    rng = np.random.default_rng(seed)
    E = offset_matrix(true_params)
    base = make_T(np.eye(3), np.array([0.1, -0.15, 1.45]))
    a = (2.0 * np.pi / n_rays) * (np.arange(n_rays) + 0.5)
    d_l = np.stack([np.cos(a), np.sin(a), np.zeros_like(a)], axis=-1)
    segs = []
    for i in range(n_seg):
        th = 2.0 * np.pi * turns * i / n_seg
        head = base @ make_T(rot_x(th), np.zeros(3))
        W = head @ E
        d_w = d_l @ W[:3, :3].T
        o_w = np.broadcast_to(W[:3, 3], d_w.shape)
        t = _ray_box_exit(o_w, d_w, *CAL_ROOM)
        for bmin, bmax in CAL_OBSTACLES:
            t = np.minimum(t, _ray_box_enter(o_w, d_w, bmin, bmax))
        r = t + rng.normal(0.0, sigma, size=t.shape)
        keep = (r > 1.0) & np.isfinite(r)            # the calibration node drops r <= 1 (m3d_calibration_twiddle.cpp:479)
        segs.append(((d_l[keep] * r[keep, None]).astype(np.float32), head.astype(np.float32)))
    return segs


def loop_poses(n_keyframes=50, per_lap=40, seed=9000, step_noise_deg=0.15, step_noise_m=0.01):
    """[(T_true, T_odom, scan seed)] of loop_trajectory (the poses alone: the sweeps can then be ray-cast in parallel)."""
    rng = np.random.default_rng(seed)
    out, T_odo, T_prev = [], None, None
    for k in range(n_keyframes):
        a = 2.0 * np.pi * k / per_lap
        x, y = 12.0 * np.cos(a), 8.0 * np.sin(a)
        yaw = np.degrees(np.arctan2(8.0 * np.cos(a), -12.0 * np.sin(a)))   # tangent of the ellipse
        T = sensor_pose(x, y, yaw)
        if T_odo is None:
            T_odo = T.copy()
        else:
            T_odo = T_odo @ (inv_T(T_prev) @ T) @ random_T(rng, step_noise_deg, step_noise_m)
        T_prev = T
        out.append((T, T_odo.copy(), seed + 1 + k))
    return out


def loop_trajectory(n_keyframes=50, per_lap=40, n_azimuth=400, seed=9000, step_noise_deg=0.15, step_noise_m=0.01, self_filter=1.0):
    """A closed trajectory for the loop-closure candidate generation: the unit drives an ellipse (12 m x 8 m
    semi-axes) through the room, `per_lap` keyframes per lap, n_keyframes in all — with the defaults a lap and a quarter, so keyframes
    40 .. 49 stand where 0 .. 9 stood. Returns [(cloud in the sensor frame, T_true, T_odom)]: T_true = sensor -> world, T_odom = the pose a
    drifting odometry would report (the true steps chained with a small random error each: what the signatures are built from)."""
    return [(hdl32_scan(T, n_azimuth, sd, self_filter=self_filter), T, T_odo)
            for T, T_odo, sd in loop_poses(n_keyframes, per_lap, seed, step_noise_deg, step_noise_m)]


def rotating_laser_sweep(pose, n_msgs=260, n_rays=541, seed=0, turn=1.15 * np.pi, sigma=0.01, fov_deg=270.0):
    """The INPUT of m3d_aggregator for one sweep: a 2-D laser scanner on a head that turns about the unit's x axis —
    `n_msgs` sensor_msgs/LaserScan messages over `turn` radians of head rotation (the aggregator publishes after 1.1 pi, m3d_aggregator.cpp:30),
    each with the tf of its callback's lookup (laser frame -> unit frame: the head angle). Returns [(ranges float32[n_rays], angle_min,
    angle_increment, tf7)], ranges ray-cast into the box room from the unit's world pose `pose`."""
    rng = np.random.default_rng(seed)
    a_min = np.float32(-np.radians(fov_deg) / 2.0)
    a_inc = np.float32(np.radians(fov_deg) / (n_rays - 1))
    ang = (a_min + np.arange(n_rays, dtype=np.float32) * a_inc).astype(np.float64)
    d_l = np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], axis=-1)
    out = []
    for m in range(n_msgs):
        th = turn * m / (n_msgs - 1)
        Rh = rot_x(th)
        d_w = d_l @ (pose[:3, :3] @ Rh).T
        o_w = np.broadcast_to(pose[:3, 3], d_w.shape)
        t = _ray_box_exit(o_w, d_w, *ROOM)
        for bmin, bmax in OBSTACLES:
            t = np.minimum(t, _ray_box_enter(o_w, d_w, bmin, bmax))
        r = (t + rng.normal(0.0, sigma, size=t.shape)).astype(np.float32)
        tf7 = np.array([0.0, 0.0, 0.0, np.sin(th / 2.0), 0.0, 0.0, np.cos(th / 2.0)])
        out.append((r, float(a_min), float(a_inc), tf7))
    return out
