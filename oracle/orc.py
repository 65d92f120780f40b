"""ctypes wrapper of oracle/libm3d_oracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
(mandala_mapping_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from mandala_mapping_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms). Called by __graft_entry__.build()."""
    targets = ["libm3d_oracle.so", "libm3d_oracle_omp.so"]
    if force or not all(os.path.exists(os.path.join(_HERE, t)) for t in targets):
        subprocess.run(["make", "-C", _HERE, "-s"] + targets, check=True)


def _lib(omp=False):
    name = "libm3d_oracle_omp.so" if omp else "libm3d_oracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, i32p, u32p, f32p, f64p, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int64)
        L.orc_default_params.argtypes = [C.POINTER(abi.Params)]
        L.orc_cloud_create.argtypes = [C.POINTER(abi.Params), vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]
        L.orc_cloud_create_source.argtypes = [C.POINTER(abi.Params), vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]
        L.orc_cloud_destroy.argtypes = [vp]
        L.orc_cloud_destroy.restype = None
        L.orc_cloud_grid_info.argtypes = [vp, C.c_int, C.POINTER(abi.GridInfo)]
        L.orc_cloud_export.argtypes = [vp, C.c_int, u32p, u32p, i32p, f32p, f32p, u32p, i32p]
        L.orc_debug_nn.argtypes = [vp, C.c_int, f32p, C.c_size_t, C.c_float, i32p, f32p]
        L.orc_debug_accumulate.argtypes = [C.POINTER(abi.Params), vp, vp, C.c_int, f32p, i64p, i32p, i32p, f32p]
        L.orc_align_clouds.argtypes = [C.POINTER(abi.Params), vp, vp, f32p, f32p, C.POINTER(abi.Stats), f64p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_agg_create.argtypes = [f64p]
        L.orc_agg_create.restype = vp
        L.orc_agg_destroy.argtypes = [vp]
        L.orc_agg_destroy.restype = None
        L.orc_agg_restart.argtypes = [vp]
        L.orc_agg_restart.restype = None
        L.orc_agg_add_cloud.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, f64p]
        L.orc_agg_add_cloud.restype = None
        L.orc_agg_add_scan.argtypes = [vp, f32p, C.c_size_t, C.c_float, C.c_float, f64p]
        L.orc_agg_add_scan.restype = None
        L.orc_agg_add_scan2.argtypes = [vp, f32p, C.c_size_t, C.c_float, C.c_float, f64p, C.c_int]
        L.orc_agg_add_scan2.restype = None
        L.orc_agg_count.argtypes = [vp]
        L.orc_agg_count.restype = C.c_size_t
        L.orc_agg_points.argtypes = [vp, f32p]
        L.orc_agg_points.restype = None
        for fn in ("orc_agg_angle", "orc_agg_progress"):
            getattr(L, fn).argtypes = [vp]
            getattr(L, fn).restype = C.c_double
        L.orc_agg_ready.argtypes = [vp]
        L.orc_cal_test_data.argtypes = [f32p, i32p, f32p, C.c_int32, C.c_int32, f32p, i64p]
        L.orc_cal_test_data.restype = C.c_int64
        L.orc_cal_test_data_bruteforce.argtypes = [f32p, i32p, f32p, C.c_int32, C.c_int32, f32p]
        L.orc_cal_test_data_bruteforce.restype = C.c_int64
        L.orc_cal_offset_matrix.argtypes = [f32p, f32p, f32p]
        L.orc_cal_offset_matrix.restype = None
        L.orc_map_create.argtypes = [C.c_float, C.c_size_t]
        L.orc_map_create.restype = vp
        L.orc_map_destroy.argtypes = [vp]
        L.orc_map_destroy.restype = None
        L.orc_map_size.argtypes = [vp]
        L.orc_map_size.restype = C.c_size_t
        L.orc_map_points.argtypes = [vp, f32p]
        L.orc_map_points.restype = None
        L.orc_map_insert.argtypes = [vp, f32p, C.c_size_t, f32p]
        L.orc_map_insert.restype = C.c_size_t
        L.orc_sincosf.argtypes = [C.c_float, f32p, f32p]
        L.orc_sincosf.restype = None
        L.orc_loop_create.argtypes = [C.POINTER(abi.LoopParams)]
        L.orc_loop_create.restype = vp
        L.orc_loop_destroy.argtypes = [vp]
        L.orc_loop_destroy.restype = None
        L.orc_loop_size.argtypes = [vp]
        L.orc_loop_add.argtypes = [vp, f32p, C.c_size_t, f32p]
        L.orc_loop_update.argtypes = [vp, C.c_int, f32p, C.c_size_t, f32p]
        L.orc_loop_update.restype = None
        L.orc_loop_signature.argtypes = [vp, C.c_int, u32p, u32p]
        L.orc_loop_signature.restype = None
        L.orc_loop_candidates.argtypes = [vp, C.c_int, C.c_int, C.POINTER(abi.LoopCandidate), C.c_size_t]
        L.orc_loop_candidates.restype = C.c_size_t
        _LIBS[name] = L
    return _LIBS[name]


def set_threads(n, omp=True):
    """OpenMP thread count of the oracle (the env var is read too early when torch loaded libgomp first)."""
    return _lib(omp).orc_set_threads(int(n))


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def default_params():
    p = abi.Params()
    _lib().orc_default_params(C.byref(p))
    return p


def _check(rc, where):
    if rc != 0:
        raise abi.M3dregError(rc, "oracle." + where)


class Cloud:
    """A bucketed cloud held by the oracle."""

    def __init__(self, params, data, n=None, point_step=16, offsets=(0, 4, 8), omp=False, source_only=False):
        self._L = _lib(omp)
        if isinstance(data, np.ndarray) and data.dtype == np.float32 and data.ndim == 2 and data.shape[1] == 3:
            from mandala_mapping_amd.pointcloud2 import encode_xyz
            msg = encode_xyz(data, point_step, offsets)
            data, n = msg.data, msg.n
        self.n = int(n)
        self.params = params
        self._h = C.c_void_p()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        fn = self._L.orc_cloud_create_source if source_only else self._L.orc_cloud_create
        _check(fn(C.byref(params), buf, self.n, point_step, offsets[0], offsets[1], offsets[2], C.byref(self._h)), "cloud_create")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_cloud_destroy(self._h)
            self._h = None

    def grid_info(self, level=0):
        g = abi.GridInfo()
        _check(self._L.orc_cloud_grid_info(self._h, level, C.byref(g)), "grid_info")
        return g

    def export(self, level=0):
        g = self.grid_info(level)
        n = self.n
        out = {
            "keys": np.empty(n, np.uint32), "sorted_keys": np.empty(n, np.uint32), "perm": np.empty(n, np.int32),
            "sorted_xyz": np.empty((n, 3), np.float32),
            "normals": np.empty((n, 3), np.float32) if g.has_normals else None,
            "cell_key": np.empty(g.n_cells, np.uint32), "cell_start": np.empty(g.n_cells + 1, np.int32),
        }
        _check(self._L.orc_cloud_export(self._h, level, _ptr(out["keys"], C.c_uint32), _ptr(out["sorted_keys"], C.c_uint32),
                                        _ptr(out["perm"], C.c_int32), _ptr(out["sorted_xyz"], C.c_float),
                                        _ptr(out["normals"], C.c_float), _ptr(out["cell_key"], C.c_uint32),
                                        _ptr(out["cell_start"], C.c_int32)), "cloud_export")
        return out

    def nn(self, queries, max_corr_dist, level=0):
        q = np.ascontiguousarray(queries, np.float32)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float32)
        _check(self._L.orc_debug_nn(self._h, level, _ptr(q, C.c_float), len(q), max_corr_dist, _ptr(idx, C.c_int32), _ptr(d2, C.c_float)), "debug_nn")
        return idx, d2


def _T16(T):
    """4x4 (row-major numpy) -> column-major float32[16]."""
    return np.ascontiguousarray(np.asarray(T, np.float64).T.reshape(16), np.float32)


def _from16(t, dtype=np.float64):
    return np.asarray(t, dtype).reshape(4, 4).T.copy()


def accumulate(params, src, tgt, T, level=0, want_nn=False):
    sums = np.zeros(abi.NSUMS, np.int64)
    exps = np.zeros(6, np.int32)
    nn = np.empty(src.n, np.int32) if want_nn else None
    d2 = np.empty(src.n, np.float32) if want_nn else None
    t = _T16(T)
    _check(tgt._L.orc_debug_accumulate(C.byref(params), src._h, tgt._h, level, _ptr(t, C.c_float), _ptr(sums, C.c_int64),
                                       _ptr(exps, C.c_int32), _ptr(nn, C.c_int32), _ptr(d2, C.c_float)), "debug_accumulate")
    return (sums, exps, nn, d2) if want_nn else (sums, exps)


def align(params, src, tgt, init_T=None, trace_cap=0):
    """Returns (T 4x4 float32-valued float64 array, Stats, trace [k,4,4] float64)."""
    t0 = _T16(np.eye(4) if init_T is None else init_T)
    out = np.zeros(16, np.float32)
    st = abi.Stats()
    trace = np.zeros((max(trace_cap, 1), 16), np.float64)
    tn = C.c_size_t(0)
    _check(tgt._L.orc_align_clouds(C.byref(params), src._h, tgt._h, _ptr(t0, C.c_float), _ptr(out, C.c_float), C.byref(st),
                                   _ptr(trace, C.c_double) if trace_cap else None, trace_cap, C.byref(tn)), "align_clouds")
    k = min(tn.value, trace_cap)
    return _from16(out), st, np.stack([_from16(trace[i]) for i in range(k)]) if k else np.zeros((0, 4, 4))


class Aggregator:
    """oracle/m3d_agg_oracle.c: the reference's pointCloudAggregator restated (m3d_aggregator.cpp:22-143)."""

    def __init__(self, bbox=(1.0, -1.0, 1.0, -1.0, 1.0, -1.0)):
        self._L = _lib()
        bb = np.asarray(bbox, np.float64)
        self._a = C.c_void_p(self._L.orc_agg_create(_ptr(bb, C.c_double)))

    def __del__(self):
        if getattr(self, "_a", None):
            self._L.orc_agg_destroy(self._a)
            self._a = None

    def add_cloud(self, msg, tf7):
        ox, oy, oz = msg.xyz_offsets()
        buf = (C.c_char * len(msg.data)).from_buffer_copy(msg.data)
        t = np.asarray(tf7, np.float64)
        self._L.orc_agg_add_cloud(self._a, buf, msg.n, msg.point_step, ox, oy, oz, _ptr(t, C.c_double))

    def add_scan(self, ranges, angle_min, angle_increment, tf7, float_overload=False):
        """float_overload: which cos / sin `m3d_aggregator.cpp:281-282` resolves to (m3d_agg_oracle.c: orc_agg_add_scan2)"""
        r = np.ascontiguousarray(ranges, np.float32)
        t = np.asarray(tf7, np.float64)
        self._L.orc_agg_add_scan2(self._a, _ptr(r, C.c_float), len(r), angle_min, angle_increment, _ptr(t, C.c_double), int(float_overload))   # (1 / True: Spec §Trig; 2: the C library's cosf / sinf)

    def status(self):
        return {"progress": self._L.orc_agg_progress(self._a), "ready": bool(self._L.orc_agg_ready(self._a)),
                "angle": self._L.orc_agg_angle(self._a), "n": self._L.orc_agg_count(self._a)}

    def points(self):
        n = self._L.orc_agg_count(self._a)
        out = np.zeros((max(n, 1), 4), np.float32)
        self._L.orc_agg_points(self._a, _ptr(out, C.c_float))
        return out[:n]

    def restart(self):
        self._L.orc_agg_restart(self._a)


class Calibration:
    """oracle/m3d_cal_oracle.c: the reference's `testData` restated (m3d_calibration_twiddle.cpp:199-308)."""

    def __init__(self, laser_up_axis=1):
        self._L = _lib()
        self.axis = laser_up_axis
        self._xyz, self._n, self._T = [], [], []

    def add_segment(self, xyz, original_T):
        a = np.ascontiguousarray(xyz, np.float32)
        T = np.asarray(original_T, np.float32)
        self._xyz.append(a)
        self._n.append(len(a))
        self._T.append(np.concatenate([T[:3, :3].reshape(9), T[:3, 3]]).astype(np.float32))

    def _pack(self):
        xyz = np.ascontiguousarray(np.concatenate(self._xyz), np.float32) if self._xyz else np.zeros((0, 3), np.float32)
        return xyz, np.asarray(self._n, np.int32), np.ascontiguousarray(np.stack(self._T), np.float32)

    def test_data(self, params, brute=False):
        """params: (x, y, z, yaw, pitch, roll) -> (cost, sizes[4]) or cost (brute force variant)"""
        xyz, n, T = self._pack()
        p = np.asarray(params, np.float32)
        if brute:
            return int(self._L.orc_cal_test_data_bruteforce(_ptr(xyz, C.c_float), _ptr(n, C.c_int32), _ptr(T, C.c_float), len(n), self.axis, _ptr(p, C.c_float)))
        sizes = np.zeros(4, np.int64)
        c = int(self._L.orc_cal_test_data(_ptr(xyz, C.c_float), _ptr(n, C.c_int32), _ptr(T, C.c_float), len(n), self.axis, _ptr(p, C.c_float), _ptr(sizes, C.c_int64)))
        return c, sizes

    def offset_matrix(self, params):
        p = np.asarray(params, np.float32)
        l, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
        self._L.orc_cal_offset_matrix(_ptr(p, C.c_float), _ptr(l, C.c_float), _ptr(t, C.c_float))
        M = np.eye(4, dtype=np.float32)
        M[:3, :3] = l.reshape(3, 3); M[:3, 3] = t
        return M


def decode_pc2(msg):
    """sensor_msgs/PointCloud2 -> float32 [n,3], field by field and point by point (SURVEY §8 row f3 oracle): x / y / z
    found by name as pcl::fromPCLPointCloud2 does (m3d_aggregator.cpp:243-246), FLOAT32 or FLOAT64 (rounded to nearest
    float), either byte order, point i at (i // width) * row_step + (i % width) * point_step."""
    raw = memoryview(msg.data)
    off, kind = {}, {}
    for f in msg.fields:
        if f.name in ("x", "y", "z"):
            if f.datatype not in (7, 8):
                raise ValueError("x/y/z must be FLOAT32 or FLOAT64")
            off[f.name], kind[f.name] = f.offset, f.datatype
    if len(off) != 3:
        raise ValueError("PointCloud2 lacks x/y/z")
    row_step = msg.row_step or msg.width * msg.point_step
    n = msg.width * msg.height
    i = np.arange(n)
    base = (i // msg.width) * row_step + (i % msg.width) * msg.point_step
    data = np.frombuffer(raw, dtype=np.uint8)
    out = np.empty((n, 3), np.float32)
    for a, name in enumerate(("x", "y", "z")):
        sz = 8 if kind[name] == 8 else 4
        idx = base[:, None] + off[name] + np.arange(sz)[None, :]
        b = np.ascontiguousarray(data[idx])
        dt = (">" if msg.is_bigendian else "<") + ("f8" if sz == 8 else "f4")
        out[:, a] = b.view(dt).reshape(n).astype(np.float32)
    return out


class Map:
    """oracle/m3d_map_oracle.c: the persistent voxel-deduplicated map of SURVEY §8 row f4."""

    def __init__(self, leaf, capacity):
        self._L = _lib()
        self._m = C.c_void_p(self._L.orc_map_create(leaf, capacity))

    def __del__(self):
        try:
            self._L.orc_map_destroy(self._m)
        except Exception:
            pass

    def insert(self, xyz, T):
        a = np.ascontiguousarray(xyz, np.float32)
        t = np.ascontiguousarray(np.asarray(T, np.float64).T.reshape(16).astype(np.float32))   # column-major float[16]
        return int(self._L.orc_map_insert(self._m, _ptr(a, C.c_float), len(a), _ptr(t, C.c_float)))

    def points(self):
        n = int(self._L.orc_map_size(self._m))
        out = np.zeros((max(n, 1), 3), np.float32)
        self._L.orc_map_points(self._m, _ptr(out, C.c_float))
        return out[:n]


def sincosf_spec(x):
    """Spec §Trig (oracle/m3d_agg_oracle.c: orc_sincosf_spec): (sin, cos) of float32 values, as float32 arrays"""
    a = np.ascontiguousarray(np.atleast_1d(x), np.float32)
    s_, c_ = np.empty_like(a), np.empty_like(a)
    L = _lib()
    sv, cv = C.c_float(), C.c_float()
    for i, v in enumerate(a):
        L.orc_sincosf(float(v), C.byref(sv), C.byref(cv))
        s_[i], c_[i] = sv.value, cv.value
    return s_, c_


class Loop:
    """oracle/m3d_loop_oracle.c: loop-closure candidate generation of SURVEY §8 row f4 (keyframes = pose + points; candidates by signature overlap)."""

    def __init__(self, params):
        self._L = _lib()
        self.params = params
        self._l = C.c_void_p(self._L.orc_loop_create(C.byref(params)))

    def __del__(self):
        try:
            self._L.orc_loop_destroy(self._l)
        except Exception:
            pass

    def __len__(self):
        return int(self._L.orc_loop_size(self._l))

    def add_keyframe(self, xyz, T):
        a = np.ascontiguousarray(xyz, np.float32)
        return int(self._L.orc_loop_add(self._l, _ptr(a, C.c_float), len(a), _ptr(_T16(T), C.c_float)))

    def update_pose(self, k, xyz, T):
        a = np.ascontiguousarray(xyz, np.float32)
        self._L.orc_loop_update(self._l, k, _ptr(a, C.c_float), len(a), _ptr(_T16(T), C.c_float))

    def signature(self, k):
        w = np.zeros(1 << (self.params.sig_log2_bits - 5), np.uint32)
        pop = C.c_uint32()
        self._L.orc_loop_signature(self._l, k, _ptr(w, C.c_uint32), C.byref(pop))
        return w, pop.value

    def candidates(self, first=0, count=-1):
        cap = max(1, (len(self) if count < 0 else count) * self.params.top_k)
        arr = (abi.LoopCandidate * cap)()
        n = int(self._L.orc_loop_candidates(self._l, first, count, arr, cap))
        return [arr[i] for i in range(n)]


# ---- bench.py's cpu_baseline legs: -O3 -march=native builds, compiled on the box that runs them -------------------------------
_NATIVE = {}


def build_native():
    """gcc -O3 -march=native of the oracle (OpenMP) and of the k-d tree ICP into a per-host directory under /tmp."""
    import hashlib
    import platform
    tag = hashlib.sha1((platform.node() + open("/proc/cpuinfo").read().split("flags")[1][:2000]).encode()).hexdigest()[:12]
    d = os.path.join("/tmp", "m3d_native_" + tag)
    if not (os.path.exists(os.path.join(d, "libm3d_oracle_native.so")) and os.path.exists(os.path.join(d, "libm3d_kdicp_native.so"))):
        subprocess.run(["make", "-C", _HERE, "-s", "native", "NATIVE_DIR=" + d], check=True)
    return d


def native_libs():
    if not _NATIVE:
        d = build_native()
        _LIBS.pop("native", None)
        L = C.CDLL(os.path.join(d, "libm3d_kdicp_native.so"))
        L.kdicp_align.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.kdicp_align.restype = C.c_longlong
        _NATIVE["kd"] = L
        _NATIVE["dir"] = d
    return _NATIVE


def kdtree_icp(src, tgt, metric, max_corr_dist, iterations, threads=1, T0=None, normal_k=10):
    """The from-scratch k-d tree ICP (oracle/m3d_kdtree_icp.c): returns (T [4,4], correspondences, {build, normals, iterations} ms)."""
    L = native_libs()["kd"]
    s = np.ascontiguousarray(src, np.float32)
    t = np.ascontiguousarray(tgt, np.float32)
    T = np.ascontiguousarray((np.eye(4) if T0 is None else np.asarray(T0, np.float64)).T.reshape(16))   # column-major
    ms = np.zeros(3, np.float64)
    n = L.kdicp_align(_ptr(s, C.c_float), len(s), _ptr(t, C.c_float), len(t), int(metric), float(max_corr_dist), int(iterations), int(normal_k),
                      int(threads), _ptr(T, C.c_double), _ptr(ms, C.c_double))
    if n < 0:
        raise RuntimeError(f"kdicp_align failed: {n}")
    return T.reshape(4, 4).T.copy(), int(n), {"build": ms[0], "normals": ms[1], "iterations": ms[2]}


def native_oracle():
    """The oracle built -O3 -march=native -fopenmp (same bits as the -O2 build: -ffp-contract=off is kept)."""
    if "oracle" not in _NATIVE:
        d = native_libs()["dir"]
        saved = _LIBS.get("libm3d_oracle_omp.so")
        _LIBS.pop("libm3d_oracle_omp.so", None)
        # _lib() declares the prototypes on whatever it loads under that key: load the native object through it
        global _HERE
        here, _HERE = _HERE, d
        try:
            os.symlink(os.path.join(d, "libm3d_oracle_native.so"), os.path.join(d, "libm3d_oracle_omp.so")) if not os.path.exists(os.path.join(d, "libm3d_oracle_omp.so")) else None
            _NATIVE["oracle"] = _lib(omp=True)
        finally:
            _HERE = here
            _LIBS.pop("libm3d_oracle_omp.so", None)
            if saved is not None:
                _LIBS["libm3d_oracle_omp.so"] = saved
    return _NATIVE["oracle"]
