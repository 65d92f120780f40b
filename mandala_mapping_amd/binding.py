"""ctypes binding of libm3dreg.so (include/m3dreg.h) and the host-side mirror of the
`gpu_6dslam_node` call surface.

The reference couples the aggregator to the registration node through a ROS topic
(/root/reference/m3d/m3d_husky_launch/launch/m3d_husky_bringup.launch:13 starts `gpu_6dslam_node`;
/root/reference/m3d/m3d_aggregator/src/m3d_aggregator.cpp:174,209 publishes the PointCloud2 it
consumes). `Gpu6dSlamNode` below is that consumer: `on_cloud(msg)` is the topic callback.

There is no CPU fallback anywhere in this module: if the HIP library is missing or no MI355X is
visible, loading/creating fails loudly.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import abi
from .pointcloud2 import PointCloud2, encode_xyz, to_little_endian

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libm3dreg.so")
_lib = None

EXPORTS = [
    "m3dreg_default_params", "m3dreg_create", "m3dreg_destroy", "m3dreg_backend_name", "m3dreg_last_error",
    "m3dreg_abi_version", "m3dreg_set_target_xyz", "m3dreg_align", "m3dreg_cloud_create", "m3dreg_cloud_destroy",
    "m3dreg_align_clouds", "m3dreg_align_batch", "m3dreg_set_latency_mode", "m3dreg_align_batch_async", "m3dreg_batch_wait", "m3dreg_synchronize",
    "m3dreg_get_stream", "m3dreg_cloud_levels", "m3dreg_cloud_grid_info", "m3dreg_cloud_export", "m3dreg_debug_nn",
    "m3dreg_debug_accumulate", "m3dreg_debug_trace", "m3dreg_profile_enable", "m3dreg_profile_batches", "m3dreg_profile_read", "m3dreg_debug_counters", "m3dreg_cloud_create_batch",
    "m3dreg_cloud_create_batch_async", "m3dreg_cloud_status",
    "m3dreg_cloud_create_pc2", "m3dreg_cloud_density",
    "m3dagg_create", "m3dagg_destroy", "m3dagg_add_cloud", "m3dagg_add_scan", "m3dagg_set_scan_trig", "m3dagg_set_rearm", "m3dagg_status", "m3dagg_take_cloud", "m3dagg_restart",
    "m3dagg_download",
    "m3dmap_create", "m3dmap_destroy", "m3dmap_insert", "m3dmap_size", "m3dmap_as_cloud", "m3dmap_download", "m3dmap_clear",
    "m3dcal_create", "m3dcal_destroy", "m3dcal_add_segment", "m3dcal_evaluate", "m3dcal_twiddle", "m3dcal_anneal",
    "m3dreg_multi_create", "m3dreg_multi_destroy", "m3dreg_multi_align", "m3dreg_multi_last_error", "m3dreg_debug_multi_clouds",
    "m3dreg_debug_fail_alloc", "m3dreg_debug_throw", "m3dreg_debug_checks", "m3dreg_debug_cloud_raw",
    "m3dloop_default_params", "m3dloop_create", "m3dloop_destroy", "m3dloop_clear", "m3dloop_add_keyframe", "m3dloop_update_pose", "m3dloop_size", "m3dloop_candidates",
    "m3dloop_make_pairs", "m3dloop_make_pair_descs", "m3dloop_gate", "m3dloop_signature", "m3dloop_last_profile",
    "m3dreg_debug_candidates", "m3dreg_host_alloc", "m3dreg_host_free", "m3dreg_host_register", "m3dreg_host_unregister",
]


def lib():
    """Load libm3dreg.so once. torch (when installed) is imported first so that this process ends up
    with a single HIP runtime: torch bundles libamdhip64.so.7 under the same soname the library needs."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    LIB_PATH = os.environ.get("M3DREG_LIB") or LIB_PATH   # A/B of two builds on one box (scripts/ab.sh)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the registration path)")
    try:
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.m3dreg_abi_version.restype = C.c_int
    if L.m3dreg_abi_version() != abi.ABI_VERSION:
        # checked for EVERY library, M3DREG_LIB included (a stale build is most likely to arrive that way: struct strides differ between versions, every
        # pair after the first would be read at the wrong offset). An A/B against an older build must say so: M3DREG_ALLOW_ABI_MISMATCH=1 (scripts/ab2.sh).
        msg = f"libm3dreg.so ABI version mismatch: library {L.m3dreg_abi_version()}, binding {abi.ABI_VERSION} ({L._name})"
        if os.environ.get("M3DREG_ALLOW_ABI_MISMATCH") != "1":
            raise RuntimeError(msg)
        print("[m3dreg] " + msg + " - allowed by M3DREG_ALLOW_ABI_MISMATCH=1, at the caller's risk", file=sys.stderr)
    vp, sz = C.c_void_p, C.c_size_t
    f32p, f64p, i32p, u32p, i64p = (C.POINTER(t) for t in (C.c_float, C.c_double, C.c_int32, C.c_uint32, C.c_int64))
    L.m3dreg_default_params.argtypes = [C.POINTER(abi.Params)]
    L.m3dreg_create.argtypes = [C.POINTER(abi.Params), C.c_int, vp, C.POINTER(vp)]
    L.m3dreg_destroy.argtypes = [vp]
    L.m3dreg_backend_name.restype = C.c_char_p
    L.m3dreg_last_error.argtypes = [vp]
    L.m3dreg_last_error.restype = C.c_char_p
    L.m3dreg_set_target_xyz.argtypes = [vp, vp, sz, sz, sz, sz, sz]
    L.m3dreg_align.argtypes = [vp, vp, sz, sz, sz, sz, sz, f32p, f32p, C.POINTER(abi.Stats)]
    L.m3dreg_cloud_create.argtypes = [vp, vp, sz, sz, sz, sz, sz, C.c_int, C.POINTER(vp)]
    L.m3dreg_cloud_create_batch.argtypes = [vp, C.POINTER(abi.CloudDesc), sz, C.POINTER(vp)]
    L.m3dreg_cloud_create_batch_async.argtypes = [vp, C.POINTER(abi.CloudDesc), sz, C.POINTER(vp)]
    L.m3dreg_cloud_status.argtypes = [vp, vp]
    L.m3dreg_cloud_create_pc2.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(abi.PointField), sz, C.c_int, C.c_int, C.POINTER(vp)]
    L.m3dreg_cloud_destroy.argtypes = [vp, vp]
    L.m3dreg_align_clouds.argtypes = [vp, vp, vp, f32p, f32p, C.POINTER(abi.Stats)]
    L.m3dreg_align_batch.argtypes = [vp, C.POINTER(abi.Pair), sz, f32p, C.POINTER(abi.Stats)]
    L.m3dreg_align_batch_async.argtypes = [vp, C.POINTER(abi.Pair), sz]
    L.m3dreg_batch_wait.argtypes = [vp, f32p, C.POINTER(abi.Stats)]
    L.m3dreg_synchronize.argtypes = [vp]
    L.m3dreg_get_stream.argtypes = [vp]
    L.m3dreg_host_alloc.argtypes = [sz, C.POINTER(vp)]
    L.m3dreg_debug_candidates.argtypes = [vp, vp, C.c_int, f32p, sz, C.POINTER(C.c_int32)]
    L.m3dreg_host_free.argtypes = [vp]
    L.m3dreg_host_register.argtypes = [vp, sz]
    L.m3dreg_host_unregister.argtypes = [vp]
    L.m3dreg_get_stream.restype = vp
    L.m3dreg_cloud_levels.argtypes = [vp]
    L.m3dreg_cloud_density.argtypes = [vp, vp, C.c_int, f64p]
    L.m3dreg_cloud_grid_info.argtypes = [vp, vp, C.c_int, C.POINTER(abi.GridInfo)]
    L.m3dreg_cloud_export.argtypes = [vp, vp, C.c_int, u32p, u32p, i32p, f32p, f32p]
    L.m3dreg_debug_nn.argtypes = [vp, vp, C.c_int, f32p, sz, C.c_float, i32p, f32p]
    L.m3dreg_debug_accumulate.argtypes = [vp, vp, vp, C.c_int, f32p, i64p, i32p]
    L.m3dreg_debug_trace.argtypes = [vp, f64p, sz, C.POINTER(sz)]
    L.m3dagg_create.argtypes = [vp, f64p, sz, C.POINTER(vp)]
    L.m3dagg_destroy.argtypes = [vp]
    L.m3dagg_add_cloud.argtypes = [vp, vp, sz, sz, sz, sz, sz, f64p]
    L.m3dagg_add_scan.argtypes = [vp, f32p, sz, C.c_float, C.c_float, f64p]
    L.m3dagg_set_scan_trig.argtypes = [vp, C.c_int]
    L.m3dagg_set_rearm.argtypes = [vp, C.c_int]
    L.m3dagg_status.argtypes = [vp, f64p, C.POINTER(C.c_int), f64p, C.POINTER(sz)]
    L.m3dagg_take_cloud.argtypes = [vp, C.POINTER(vp)]
    L.m3dagg_restart.argtypes = [vp]
    L.m3dagg_download.argtypes = [vp, f32p, sz, C.POINTER(sz)]
    L.m3dmap_create.argtypes = [vp, C.c_float, sz, C.POINTER(vp)]
    L.m3dmap_destroy.argtypes = [vp]
    L.m3dmap_insert.argtypes = [vp, vp, f32p, C.POINTER(sz)]
    L.m3dmap_size.argtypes = [vp, C.POINTER(sz)]
    L.m3dmap_as_cloud.argtypes = [vp, C.POINTER(vp)]
    L.m3dmap_download.argtypes = [vp, f32p, sz, C.POINTER(sz)]
    L.m3dmap_clear.argtypes = [vp]
    L.m3dcal_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.m3dcal_destroy.argtypes = [vp]
    L.m3dcal_add_segment.argtypes = [vp, vp, sz, sz, sz, sz, sz, f32p]
    L.m3dcal_evaluate.argtypes = [vp, f32p, sz, i64p, i64p]
    L.m3dcal_twiddle.argtypes = [vp, C.c_int, f32p, f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.m3dcal_anneal.argtypes = [vp, C.c_uint, f32p, f32p, C.POINTER(C.c_int)]
    L.m3dreg_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.m3dreg_set_latency_mode.argtypes = [vp, C.c_int]
    L.m3dreg_profile_enable.argtypes = [vp, C.c_int]
    L.m3dreg_profile_batches.argtypes = [vp, C.c_int]
    L.m3dreg_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64), f64p, C.c_int]
    L.m3dreg_multi_create.argtypes = [C.POINTER(abi.Params), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.m3dreg_multi_destroy.argtypes = [vp]
    L.m3dreg_multi_align.argtypes = [vp, C.POINTER(abi.PairDesc), sz, f32p, C.POINTER(abi.Stats), i32p]
    L.m3dreg_multi_last_error.argtypes = [vp]
    L.m3dreg_debug_multi_clouds.argtypes = [vp]
    L.m3dreg_multi_last_error.restype = C.c_char_p
    L.m3dloop_default_params.argtypes = [C.POINTER(abi.LoopParams)]
    L.m3dloop_create.argtypes = [vp, C.POINTER(abi.LoopParams), C.POINTER(vp)]
    L.m3dloop_destroy.argtypes = [vp]
    L.m3dloop_clear.argtypes = [vp]
    L.m3dloop_add_keyframe.argtypes = [vp, vp, f32p, C.POINTER(abi.CloudDesc), i32p]
    L.m3dloop_update_pose.argtypes = [vp, C.c_int32, f32p]
    L.m3dloop_size.argtypes = [vp, C.POINTER(sz)]
    L.m3dloop_candidates.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(abi.LoopCandidate), sz, C.POINTER(sz)]
    L.m3dloop_make_pairs.argtypes = [vp, C.POINTER(abi.LoopCandidate), sz, C.POINTER(abi.Pair)]
    L.m3dloop_make_pair_descs.argtypes = [vp, C.POINTER(abi.LoopCandidate), sz, C.POINTER(abi.PairDesc)]
    L.m3dloop_gate.argtypes = [C.POINTER(abi.LoopCandidate), C.POINTER(abi.Stats), sz, C.c_int64, C.c_double, C.POINTER(C.c_uint8)]
    L.m3dloop_signature.argtypes = [vp, C.c_int32, u32p, u32p]
    L.m3dloop_last_profile.argtypes = [vp, f64p, C.POINTER(C.c_uint64)]
    L.m3dreg_debug_checks.argtypes = [vp, u32p, C.c_int]
    L.m3dreg_debug_cloud_raw.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.m3dreg_debug_fail_alloc.argtypes = [C.c_int]
    L.m3dreg_debug_throw.argtypes = [C.c_int]
    _lib = L
    return L


def default_params():
    p = abi.Params()
    lib().m3dreg_default_params(C.byref(p))
    return p


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def T_to_colmajor16(T):
    return np.ascontiguousarray(np.asarray(T, np.float64).T.reshape(16), np.float32)


def colmajor16_to_T(t, dtype=np.float64):
    return np.asarray(t, dtype).reshape(4, 4).T.copy()


class Cloud:
    """A bucketed cloud resident in HBM (m3dreg_cloud)."""

    def __init__(self, reg, ptr, n):
        self._reg, self._p, self.n = reg, ptr, n

    def free(self):
        if self._p:
            lib().m3dreg_cloud_destroy(self._reg._h, self._p)
            self._p = None

    def __del__(self):
        try:
            if self._p and self._reg._h:
                self.free()
        except Exception:
            pass

    def status(self):
        """0, or the (negative) m3dreg_error the device found while bucketing this cloud; waits for the bucketing."""
        return lib().m3dreg_cloud_status(self._reg._h, self._p)

    def density(self, level=None):
        """mean population of the voxel a point lies in (m3dreg_cloud_density; default: the finest level) — the LPT cost estimate"""
        d = C.c_double()
        lv = lib().m3dreg_cloud_levels(self._p) - 1 if level is None else level
        self._reg._check(lib().m3dreg_cloud_density(self._reg._h, self._p, lv, C.byref(d)), "cloud_density")
        return d.value

    def grid_info(self, level=0):
        g = abi.GridInfo()
        self._reg._check(lib().m3dreg_cloud_grid_info(self._reg._h, self._p, level, C.byref(g)), "cloud_grid_info")
        return g

    def export(self, level=0):
        g = self.grid_info(level)
        n = self.n
        out = {"keys": np.empty(n, np.uint32), "sorted_keys": np.empty(n, np.uint32), "perm": np.empty(n, np.int32),
               "sorted_xyz": np.empty((n, 3), np.float32), "normals": np.empty((n, 3), np.float32) if g.has_normals else None}
        self._reg._check(lib().m3dreg_cloud_export(self._reg._h, self._p, level, _ptr(out["keys"], C.c_uint32),
                                                   _ptr(out["sorted_keys"], C.c_uint32), _ptr(out["perm"], C.c_int32),
                                                   _ptr(out["sorted_xyz"], C.c_float), _ptr(out["normals"], C.c_float)), "cloud_export")
        return out

    RAW = {"htab": 0, "thdr": 1, "timg": 2, "timeta": 3, "occ": 4, "meta": 5, "order": 6}

    def raw(self, what, level=0):
        """diagnosis: the bytes of one search structure of this level as they lie in HBM (m3dreg_debug_cloud_raw; layouts: csrc/m3d_device.h), as a uint32 array"""
        n = C.c_size_t()
        self._reg._check(lib().m3dreg_debug_cloud_raw(self._reg._h, self._p, level, self.RAW[what], None, 0, C.byref(n)), "debug_cloud_raw")
        out = np.empty(n.value // 4, np.uint32)
        self._reg._check(lib().m3dreg_debug_cloud_raw(self._reg._h, self._p, level, self.RAW[what], _ptr(out, C.c_uint32), n.value, C.byref(n)), "debug_cloud_raw")
        return out

    def nn(self, queries, max_corr_dist, level=0):
        q = np.ascontiguousarray(queries, np.float32)
        idx, d2 = np.empty(len(q), np.int32), np.empty(len(q), np.float32)
        self._reg._check(lib().m3dreg_debug_nn(self._reg._h, self._p, level, _ptr(q, C.c_float), len(q), max_corr_dist,
                                               _ptr(idx, C.c_int32), _ptr(d2, C.c_float)), "debug_nn")
        return idx, d2


    def candidates(self, queries, level=0):
        """points in the 27 voxels around every query (what the spec's exhaustive search compares)"""
        q = np.ascontiguousarray(queries, np.float32)
        cnt = np.empty(len(q), np.int32)
        self._reg._check(lib().m3dreg_debug_candidates(self._reg._h, self._p, level, _ptr(q, C.c_float), len(q), _ptr(cnt, C.c_int32)), "debug_candidates")
        return cnt


class Registrar:
    """One m3dreg_handle: a device, a stream, a parameter set."""

    def __init__(self, params=None, device=0, stream=None):
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params()
        rc = lib().m3dreg_create(C.byref(self.params), device, stream, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise abi.M3dregError(rc, "m3dreg_create", "no usable MI355X / HIP runtime (no CPU fallback exists)" if rc == abi.ERR_NO_DEVICE else "")
        self.device = device

    def close(self):
        if self._h:
            lib().m3dreg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, where):
        if rc != 0:
            raise abi.M3dregError(rc, where, lib().m3dreg_last_error(self._h).decode())

    @property
    def stream(self):
        return lib().m3dreg_get_stream(self._h)

    def synchronize(self):
        self._check(lib().m3dreg_synchronize(self._h), "synchronize")

    def profile_enable(self, on=True, every=1):
        """every = n: bracket every n-th iteration only (each event record is a barrier packet on the stream)"""
        self._check(lib().m3dreg_profile_enable(self._h, max(1, int(every)) if on else 0), "profile_enable")

    def profile_batches(self, every=1):
        """the per-batch brackets (bucketing, whole chain) on every n-th batch only (m3dreg_profile_batches; default: every batch)"""
        self._check(lib().m3dreg_profile_batches(self._h, max(1, int(every))), "profile_batches")

    def profile_read(self, what=1, reset=True):
        """(launches, total ms) since the last reset, from hipEvents on the stream.
        what = 0: whole iterations (search + reduction + solve); what = 1: the dominant kernel (k_nn_iter) alone."""
        n, ms = C.c_uint64(0), C.c_double(0.0)
        self._check(lib().m3dreg_profile_read(self._h, what, C.byref(n), C.byref(ms), 1 if reset else 0), "profile_read")
        return n.value, ms.value

    # ---- clouds -------------------------------------------------------------------------------
    def cloud(self, data, n=None, point_step=16, offsets=(0, 4, 8)):
        """data: float32 [n,3] array, PointCloud2, or raw payload bytes."""
        if isinstance(data, np.ndarray):
            data = encode_xyz(data)
        if isinstance(data, PointCloud2):
            msg = to_little_endian(data)
            data, n, point_step, offsets = msg.data, msg.n, msg.point_step, msg.xyz_offsets()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        p = C.c_void_p()
        self._check(lib().m3dreg_cloud_create(self._h, buf, n, point_step, offsets[0], offsets[1], offsets[2], 0, C.byref(p)), "cloud_create")
        return Cloud(self, p, n)

    def cloud_pc2(self, msg: PointCloud2, source_only=False):
        """The message as it is — field table, byte order, row padding — decoded on the device (m3dreg_cloud_create_pc2).
        source_only: the sweep will only ever be a source (and a map insert): sorted, no table, no normals."""
        k = len(msg.fields)
        ft = (abi.PointField * k)()
        for i, f in enumerate(msg.fields):
            ft[i].name, ft[i].offset, ft[i].datatype, ft[i].count = f.name.encode(), f.offset, f.datatype, f.count
        buf = (C.c_char * len(msg.data)).from_buffer_copy(msg.data)
        p = C.c_void_p()
        self._check(lib().m3dreg_cloud_create_pc2(self._h, buf, len(msg.data), msg.width, msg.height, msg.point_step, msg.row_step, ft, k,
                                                  1 if msg.is_bigendian else 0, 2 if source_only else 0, C.byref(p)), "cloud_create_pc2")
        return Cloud(self, p, msg.n)

    def cloud_from_device(self, dev_ptr, n, point_step=16, offsets=(0, 4, 8)):
        """dev_ptr: integer device address of a PointCloud2-layout payload already in HBM."""
        p = C.c_void_p()
        self._check(lib().m3dreg_cloud_create(self._h, C.c_void_p(dev_ptr), n, point_step, offsets[0], offsets[1], offsets[2], 1, C.byref(p)), "cloud_create(device)")
        return Cloud(self, p, n)

    def clouds_from_device(self, items, wait=True, source_only=None):
        """items: list of (device address, n[, point_step, (ox, oy, oz)]) -> list of Clouds, bucketed in ONE batch.
        wait=False: m3dreg_cloud_create_batch_async — enqueue only, no host synchronisation; a cloud in error (no finite point,
        grid too large) then ends its registrations with status BAD_CLOUD and Cloud.status() names the error."""
        k = len(items)
        descs = (abi.CloudDesc * k)()
        for i, it in enumerate(items):
            step = it[2] if len(it) > 2 else 16
            off = it[3] if len(it) > 3 else (0, 4, 8)
            descs[i].data, descs[i].n, descs[i].point_step = it[0], it[1], step
            descs[i].off_x, descs[i].off_y, descs[i].off_z, descs[i].data_is_device = off[0], off[1], off[2], 1
            descs[i].source_only = 1 if (source_only is not None and source_only[i]) else 0   # no normals for clouds that are only ever sources
        out = (C.c_void_p * k)()
        fn = lib().m3dreg_cloud_create_batch if wait else lib().m3dreg_cloud_create_batch_async
        self._check(fn(self._h, descs, k, out), "cloud_create_batch")
        return [Cloud(self, C.c_void_p(out[i]), items[i][1]) for i in range(k)]

    def clouds(self, arrays, wait=True, source_only=None):
        """arrays: list of float32 [n,3] arrays / PointCloud2 messages -> list of Clouds, bucketed in ONE batch (wait: see
        clouds_from_device; the host buffers must stay alive until the copies have run when wait=False)."""
        msgs = [to_little_endian(encode_xyz(a) if isinstance(a, np.ndarray) else a) for a in arrays]
        k = len(msgs)
        bufs = [np.frombuffer(m.data, np.uint8) for m in msgs]   # views of the messages' own buffers: the payload is not copied on the host
        descs = (abi.CloudDesc * k)()
        for i, m in enumerate(msgs):
            ox, oy, oz = m.xyz_offsets()
            descs[i].data, descs[i].n, descs[i].point_step = bufs[i].ctypes.data, m.n, m.point_step
            descs[i].off_x, descs[i].off_y, descs[i].off_z, descs[i].data_is_device = ox, oy, oz, 0
            descs[i].source_only = 1 if (source_only is not None and source_only[i]) else 0
        out = (C.c_void_p * k)()
        fn = lib().m3dreg_cloud_create_batch if wait else lib().m3dreg_cloud_create_batch_async
        self._check(fn(self._h, descs, k, out), "cloud_create_batch")
        cl = [Cloud(self, C.c_void_p(out[i]), msgs[i].n) for i in range(k)]
        if not wait:
            for c, b in zip(cl, bufs):
                c._keep = b
        return cl

    # ---- registration -------------------------------------------------------------------------
    def align(self, source: Cloud, target: Cloud, init_T=None):
        t0 = T_to_colmajor16(np.eye(4) if init_T is None else init_T)
        out = np.zeros(16, np.float32)
        st = abi.Stats()
        self._check(lib().m3dreg_align_clouds(self._h, source._p, target._p, _ptr(t0, C.c_float), _ptr(out, C.c_float), C.byref(st)), "align_clouds")
        return colmajor16_to_T(out), st

    def _pairs(self, pairs):
        arr = (abi.Pair * len(pairs))()
        for i, pr in enumerate(pairs):
            s, t = pr[0], pr[1]
            T0 = pr[2] if len(pr) > 2 and pr[2] is not None else np.eye(4)
            arr[i].source, arr[i].target = s._p.value, t._p.value
            arr[i].init_T[:] = T_to_colmajor16(T0).tolist()
        return arr

    def align_batch(self, pairs):
        """pairs: list of (source Cloud, target Cloud[, init_T]). Returns ([k,4,4] poses, [Stats])."""
        arr = self._pairs(pairs)
        k = len(pairs)
        out = np.zeros((k, 16), np.float32)
        st = (abi.Stats * k)()
        self._check(lib().m3dreg_align_batch(self._h, arr, k, _ptr(out, C.c_float), st), "align_batch")
        return np.stack([colmajor16_to_T(out[i]) for i in range(k)]), list(st)

    def align_batch_arr(self, pairs_arr, k):
        """the synchronous call on a prepared m3dreg_pair array (self._pairs): what a serial caller — the ROS node — makes"""
        out = np.zeros((k, 16), np.float32)
        st = (abi.Stats * k)()
        self._check(lib().m3dreg_align_batch(self._h, pairs_arr, k, _ptr(out, C.c_float), st), "align_batch")
        return np.stack([colmajor16_to_T(out[i]) for i in range(k)]), list(st)

    def set_latency_mode(self, on=True):
        """ABI 7: this handle's batches have the GPU to themselves (a serial caller — the ROS node): launch grids sized for latency; same results"""
        self._check(lib().m3dreg_set_latency_mode(self._h, int(bool(on))), "set_latency_mode")

    def align_batch_async(self, pairs_arr, k):
        self._check(lib().m3dreg_align_batch_async(self._h, pairs_arr, k), "align_batch_async")

    def batch_wait(self, k):
        out = np.zeros((k, 16), np.float32)
        st = (abi.Stats * k)()
        self._check(lib().m3dreg_batch_wait(self._h, _ptr(out, C.c_float), st), "batch_wait")
        return np.stack([colmajor16_to_T(out[i]) for i in range(k)]), list(st)

    # ---- the gpu_6dslam_node surface ------------------------------------------------------------
    def set_target(self, msg: PointCloud2):
        msg = to_little_endian(msg)
        ox, oy, oz = msg.xyz_offsets()
        buf = (C.c_char * len(msg.data)).from_buffer_copy(msg.data)
        self._check(lib().m3dreg_set_target_xyz(self._h, buf, msg.n, msg.point_step, ox, oy, oz), "set_target_xyz")

    def align_msg(self, msg: PointCloud2, init_T=None):
        msg = to_little_endian(msg)
        ox, oy, oz = msg.xyz_offsets()
        buf = (C.c_char * len(msg.data)).from_buffer_copy(msg.data)
        t0 = T_to_colmajor16(np.eye(4) if init_T is None else init_T)
        out = np.zeros(16, np.float32)
        st = abi.Stats()
        self._check(lib().m3dreg_align(self._h, buf, msg.n, msg.point_step, ox, oy, oz, _ptr(t0, C.c_float), _ptr(out, C.c_float), C.byref(st)), "align")
        return colmajor16_to_T(out), st

    # ---- introspection --------------------------------------------------------------------------
    def accumulate(self, source: Cloud, target: Cloud, T, level=0):
        sums, exps = np.zeros(abi.NSUMS, np.int64), np.zeros(6, np.int32)
        t = T_to_colmajor16(T)
        self._check(lib().m3dreg_debug_accumulate(self._h, source._p, target._p, level, _ptr(t, C.c_float), _ptr(sums, C.c_int64), _ptr(exps, C.c_int32)), "debug_accumulate")
        return sums, exps

    def counters(self):
        out = (C.c_uint64 * 2)()
        self._check(lib().m3dreg_debug_counters(self._h, out), "debug_counters")
        return int(out[0]), int(out[1])

    def checks(self, reset=False):
        """the diagnosis build's report (libm3dreg_checked.so): {"icp": (offences, site, index, bound), "bucket": (...)}; raises on the shipped library"""
        out = (C.c_uint32 * 8)()
        self._check(lib().m3dreg_debug_checks(self._h, out, int(reset)), "debug_checks")
        return {"icp": tuple(out[0:4]), "bucket": tuple(out[4:8])}

    def trace(self, cap=256):
        buf = np.zeros((cap, 16), np.float64)
        n = C.c_size_t(0)
        self._check(lib().m3dreg_debug_trace(self._h, _ptr(buf, C.c_double), cap, C.byref(n)), "debug_trace")
        k = min(n.value, cap)
        return np.stack([colmajor16_to_T(buf[i]) for i in range(k)]) if k else np.zeros((0, 4, 4))


class PinnedBuffer:
    """bytes of pinned host memory from m3dreg_host_alloc (freed with the object)"""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        rc = lib().m3dreg_host_alloc(C.c_size_t(nbytes), C.byref(self.ptr))
        if rc != 0:
            self.ptr = None
            raise abi.M3dregError(rc, "m3dreg_host_alloc")

    def __del__(self):
        try:
            if self.ptr:
                lib().m3dreg_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class MultiRegistrar:
    """m3dreg_multi: ONE process, several devices (or several streams of one). Pairs are given as raw PointCloud2 payloads
    (numpy (n, 3) float32 arrays are encoded the aggregator's way); the library shards, uploads, registers and gathers."""

    def __init__(self, params=None, devices=(0,)):
        self._m = C.c_void_p()
        self.params = params if params is not None else default_params()
        dev = (C.c_int * len(devices))(*devices)
        rc = lib().m3dreg_multi_create(C.byref(self.params), dev, len(devices), C.byref(self._m))
        if rc != 0:
            self._m = None
            raise abi.M3dregError(rc, "m3dreg_multi_create")
        self.devices = tuple(devices)

    def close(self):
        if self._m:
            lib().m3dreg_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self, pairs, source_only=False, pinned=False, groups=None):
        """pairs: [(src_xyz, tgt_xyz, T0 or None)] -> (descriptor array, buffers to keep alive): host PointCloud2 payloads, encoded the
        aggregator's way; source_only: the sources are only sorted (m3dreg_cloud_desc.source_only); pinned: the payloads live in
        m3dreg_host_alloc memory (asynchronous DMA, no staging copy) instead of pageable memory; groups[i] > 0: the pairs with this id share ONE
        target payload (m3dreg_pair_desc.target_group: the first pair's target array is encoded once and named by all of them)."""
        from .pointcloud2 import encode_xyz
        n = len(pairs)
        descs = (abi.PairDesc * n)()
        keep = []
        shared = {}   # group id -> the descriptor fields of its (one) target payload
        for i, (src, tgt, T0) in enumerate(pairs):
            gid = int(groups[i]) if groups is not None else 0
            descs[i].target_group = gid
            for d, xyz, so, is_tgt in ((descs[i].source, src, source_only, False), (descs[i].target, tgt, False, True)):
                if is_tgt and gid > 0 and gid in shared:
                    d.data, d.n, d.point_step = shared[gid]
                    d.off_x, d.off_y, d.off_z = 0, 4, 8
                    d.data_is_device = 0; d.source_only = 0
                    continue
                msg = encode_xyz(np.ascontiguousarray(xyz, np.float32))
                if pinned:
                    buf = PinnedBuffer(len(msg.data))
                    C.memmove(buf.ptr, bytes(msg.data), len(msg.data))
                    keep.append(buf)
                    d.data = buf.ptr
                else:
                    buf = (C.c_char * len(msg.data)).from_buffer_copy(msg.data)
                    keep.append(buf)
                    d.data = C.cast(buf, C.c_void_p)
                d.n = msg.n; d.point_step = msg.point_step
                d.off_x, d.off_y, d.off_z = 0, 4, 8
                d.data_is_device = 0; d.source_only = 1 if so else 0
                if is_tgt and gid > 0:
                    shared[gid] = (d.data, d.n, d.point_step)
            t0 = T_to_colmajor16(np.eye(4) if T0 is None else T0)
            for k in range(16):
                descs[i].init_T[k] = float(t0[k])
        return descs, keep

    def align_described(self, descs):
        """One m3dreg_multi_align call on descriptors built by describe(): (poses [n, 4, 4], stats list, device of every pair)."""
        n = len(descs)
        out = np.zeros(16 * n, np.float32)
        st = (abi.Stats * n)()
        dev = np.zeros(n, np.int32)
        rc = lib().m3dreg_multi_align(self._m, descs, n, _ptr(out, C.c_float), st, _ptr(dev, C.c_int32))
        if rc != 0:
            raise abi.M3dregError(rc, "m3dreg_multi_align", lib().m3dreg_multi_last_error(self._m).decode())
        return np.stack([colmajor16_to_T(out[16 * i:16 * i + 16]) for i in range(n)]), list(st), dev

    def clouds_bucketed(self):
        """clouds uploaded + bucketed by the last align (a target group's target counts once)"""
        return lib().m3dreg_debug_multi_clouds(self._m)

    def align(self, pairs, source_only=False, groups=None):
        """pairs: [(src_xyz, tgt_xyz, T0 or None)] -> (poses [n, 4, 4], stats list, device of every pair)"""
        descs, keep = self.describe(pairs, source_only, groups=groups)
        return self.align_described(descs)


class Aggregator:
    """Device-side mirror of m3d_aggregator's pointCloudAggregator (m3d_aggregator.cpp:22-143): messages in,
    one bucketed cloud per 1.1*pi of head rotation out, the sweep itself never leaves HBM."""

    def __init__(self, reg: Registrar, bbox=(1.0, -1.0, 1.0, -1.0, 1.0, -1.0), capacity=1 << 20):
        self._reg = reg
        self._a = C.c_void_p()
        bb = np.asarray(bbox, np.float64)
        reg._check(lib().m3dagg_create(reg._h, _ptr(bb, C.c_double), capacity, C.byref(self._a)), "m3dagg_create")

    def close(self):
        if self._a:
            lib().m3dagg_destroy(self._a)
            self._a = None

    def __del__(self):
        try:
            if self._reg._h:
                self.close()
        except Exception:
            pass

    def add_cloud(self, msg: PointCloud2, tf7):
        msg = to_little_endian(msg)
        ox, oy, oz = msg.xyz_offsets()
        buf = (C.c_char * len(msg.data)).from_buffer_copy(msg.data)
        t = np.asarray(tf7, np.float64)
        self._reg._check(lib().m3dagg_add_cloud(self._a, buf, msg.n, msg.point_step, ox, oy, oz, _ptr(t, C.c_double)), "m3dagg_add_cloud")

    def set_scan_trig(self, float_overload: bool):
        """which cos / sin m3d_aggregator.cpp:281-282 resolves to: False (default) = double cos(double), True = the float overload"""
        self._reg._check(lib().m3dagg_set_scan_trig(self._a, int(bool(float_overload))), "m3dagg_set_scan_trig")

    def set_rearm(self, automatic: bool):
        """False: take_cloud leaves the aggregator idle until restart(), like the reference's node between a published cloud and the next ~request"""
        self._reg._check(lib().m3dagg_set_rearm(self._a, int(bool(automatic))), "m3dagg_set_rearm")

    def add_scan(self, ranges, angle_min, angle_increment, tf7):
        r = np.ascontiguousarray(ranges, np.float32)
        t = np.asarray(tf7, np.float64)
        self._reg._check(lib().m3dagg_add_scan(self._a, _ptr(r, C.c_float), len(r), angle_min, angle_increment, _ptr(t, C.c_double)), "m3dagg_add_scan")

    def status(self):
        pr, rd, an, n = C.c_double(), C.c_int(), C.c_double(), C.c_size_t()
        self._reg._check(lib().m3dagg_status(self._a, C.byref(pr), C.byref(rd), C.byref(an), C.byref(n)), "m3dagg_status")
        return {"progress": pr.value, "ready": bool(rd.value), "angle": an.value, "n": n.value}

    def points(self):
        n = self.status()["n"]
        out = np.zeros((max(n, 1), 4), np.float32)
        k = C.c_size_t()
        self._reg._check(lib().m3dagg_download(self._a, _ptr(out, C.c_float), n, C.byref(k)), "m3dagg_download")
        return out[:n]

    def take_cloud(self):
        n = self.status()["n"]
        p = C.c_void_p()
        self._reg._check(lib().m3dagg_take_cloud(self._a, C.byref(p)), "m3dagg_take_cloud")
        return Cloud(self._reg, p, n)

    def restart(self):
        self._reg._check(lib().m3dagg_restart(self._a), "m3dagg_restart")


class Map:
    """The persistent voxel-deduplicated map in HBM: registered scans in, a bucketed target out."""

    def __init__(self, reg: Registrar, dedup_leaf=0.02, capacity=1 << 22):
        self._reg = reg
        self._m = C.c_void_p()
        reg._check(lib().m3dmap_create(reg._h, dedup_leaf, capacity, C.byref(self._m)), "m3dmap_create")

    def close(self):
        if self._m:
            lib().m3dmap_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            if self._reg._h:
                self.close()
        except Exception:
            pass

    def insert(self, scan: Cloud, T):
        t = T_to_colmajor16(T)
        k = C.c_size_t()
        self._reg._check(lib().m3dmap_insert(self._m, scan._p, _ptr(t, C.c_float), C.byref(k)), "m3dmap_insert")
        return k.value

    def __len__(self):
        n = C.c_size_t()
        self._reg._check(lib().m3dmap_size(self._m, C.byref(n)), "m3dmap_size")
        return n.value

    def as_cloud(self):
        p = C.c_void_p()
        self._reg._check(lib().m3dmap_as_cloud(self._m, C.byref(p)), "m3dmap_as_cloud")
        return Cloud(self._reg, p, len(self))

    def points(self):
        n = len(self)
        out = np.zeros((max(n, 1), 4), np.float32)
        k = C.c_size_t()
        self._reg._check(lib().m3dmap_download(self._m, _ptr(out, C.c_float), n, C.byref(k)), "m3dmap_download")
        return out[:n, :3]

    def clear(self):
        self._reg._check(lib().m3dmap_clear(self._m), "m3dmap_clear")


class LoopCloser:
    """Loop-closure candidate generation (m3dloop_*): keyframes = (pose, resident cloud); candidates (i, j) are scored on the
    device by the overlap of their coarse-voxel signatures and come out as m3dreg_pair[] / m3dreg_pair_desc[] for the batch path."""

    def __init__(self, reg: Registrar, params=None):
        self._reg = reg
        self._l = C.c_void_p()
        if params is None:
            params = abi.LoopParams()
            lib().m3dloop_default_params(C.byref(params))
        self.params = params
        self._clouds = []   # keeps the keyframes' Cloud objects alive (the library holds their pointers)
        reg._check(lib().m3dloop_create(reg._h, C.byref(params), C.byref(self._l)), "m3dloop_create")

    def close(self):
        if self._l:
            lib().m3dloop_destroy(self._l)
            self._l = None
        self._clouds = []

    def __del__(self):
        try:
            if self._reg._h:
                self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_size_t()
        self._reg._check(lib().m3dloop_size(self._l, C.byref(n)), "m3dloop_size")
        return n.value

    def add_keyframe(self, cloud: Cloud, T, payload=None):
        """T: 4x4 pose of the sweep in the map frame; payload: an abi.CloudDesc of its raw PointCloud2 payload (for pair_descs) or None"""
        t = T_to_colmajor16(T)
        k = C.c_int32(-1)
        self._reg._check(lib().m3dloop_add_keyframe(self._l, cloud._p, _ptr(t, C.c_float), C.byref(payload) if payload is not None else None, C.byref(k)), "m3dloop_add_keyframe")
        self._clouds.append((cloud, payload))
        return k.value

    def update_pose(self, index, T):
        t = T_to_colmajor16(T)
        self._reg._check(lib().m3dloop_update_pose(self._l, index, _ptr(t, C.c_float)), "m3dloop_update_pose")

    def clear(self):
        self._reg._check(lib().m3dloop_clear(self._l), "m3dloop_clear")
        self._clouds = []

    def signature(self, index):
        w = np.zeros(1 << (self.params.sig_log2_bits - 5), np.uint32)
        pop = C.c_uint32()
        self._reg._check(lib().m3dloop_signature(self._l, index, _ptr(w, C.c_uint32), C.byref(pop)), "m3dloop_signature")
        return w, pop.value

    def candidates(self, first=0, count=-1):
        """-> ctypes array of abi.LoopCandidate (rows first .. first + count - 1; count < 0: to the newest keyframe)"""
        n = C.c_size_t()
        cap = max(1, (len(self) if count < 0 else count) * self.params.top_k)
        arr = (abi.LoopCandidate * cap)()
        self._reg._check(lib().m3dloop_candidates(self._l, first, count, arr, cap, C.byref(n)), "m3dloop_candidates")
        return (abi.LoopCandidate * n.value).from_buffer_copy(bytes(arr)[: n.value * C.sizeof(abi.LoopCandidate)]) if n.value else (abi.LoopCandidate * 0)()

    def last_profile(self):
        ms, b = C.c_double(), C.c_uint64()
        self._reg._check(lib().m3dloop_last_profile(self._l, C.byref(ms), C.byref(b)), "m3dloop_last_profile")
        return ms.value, b.value

    def pairs(self, cands):
        arr = (abi.Pair * max(1, len(cands)))()
        self._reg._check(lib().m3dloop_make_pairs(self._l, cands, len(cands), arr), "m3dloop_make_pairs")
        return arr

    def pair_descs(self, cands):
        arr = (abi.PairDesc * max(1, len(cands)))()
        self._reg._check(lib().m3dloop_make_pair_descs(self._l, cands, len(cands), arr), "m3dloop_make_pair_descs")
        return arr

    @staticmethod
    def gate(cands, stats, min_corr, max_rms):
        n = len(cands)
        st = (abi.Stats * max(1, n))(*stats)
        acc = (C.c_uint8 * max(1, n))()
        rc = lib().m3dloop_gate(cands, st, n, int(min_corr), float(max_rms), acc)
        if rc != 0:
            raise abi.M3dregError(rc, "m3dloop_gate")
        return [bool(acc[i]) for i in range(n)]


class Calibrator:
    """Device-side mirror of the cost function and the two optimiser loops of the reference's calibration nodes
    (m3d_calibration_twiddle.cpp:199-396, m3d_calibration_sa.cpp:199-356): scan segments in, outlier counts /
    the calibrated mounting offset out. Many parameter candidates are evaluated per launch."""

    def __init__(self, reg: Registrar, laser_up_axis=1):
        self._reg = reg
        self._c = C.c_void_p()
        reg._check(lib().m3dcal_create(reg._h, laser_up_axis, C.byref(self._c)), "m3dcal_create")

    def close(self):
        if self._c:
            lib().m3dcal_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            if self._reg._h:
                self.close()
        except Exception:
            pass

    def add_segment(self, xyz, original_T):
        """xyz: float32 [n,3] points of one scan in the laser frame; original_T: 4x4 transform of that scan (:56-69)"""
        a = np.ascontiguousarray(xyz, np.float32)
        t = np.ascontiguousarray(np.asarray(original_T, np.float32).T.reshape(16))   # column-major, as Eigen::Affine3f::data()
        self._reg._check(lib().m3dcal_add_segment(self._c, a.ctypes.data_as(C.c_void_p), len(a), 12, 0, 4, 8, _ptr(t, C.c_float)), "m3dcal_add_segment")

    def evaluate(self, params, with_voxels=False):
        """params: [k,6] (x, y, z, yaw, pitch, roll) -> int64 [k] costs (and [k,2] voxel counts of the two halves)"""
        p = np.ascontiguousarray(np.atleast_2d(params), np.float32)
        k = len(p)
        out = np.zeros(k, np.int64)
        vox = np.zeros((k, 2), np.int64)
        self._reg._check(lib().m3dcal_evaluate(self._c, _ptr(p, C.c_float), k, _ptr(out, C.c_int64), _ptr(vox, C.c_int64)), "m3dcal_evaluate")
        return (out, vox) if with_voxels else out

    def twiddle(self, max_sweeps=0):
        p = np.zeros(5, np.float32)
        err, sw, ev = C.c_float(), C.c_int(), C.c_int()
        self._reg._check(lib().m3dcal_twiddle(self._c, max_sweeps, _ptr(p, C.c_float), C.byref(err), C.byref(sw), C.byref(ev)), "m3dcal_twiddle")
        return p, err.value, sw.value, ev.value

    def anneal(self, seed, p0=(0.0, 0.12, 0.0, 0.0, 0.0)):
        p = np.asarray(p0, np.float32).copy()
        err, ev = C.c_float(), C.c_int()
        self._reg._check(lib().m3dcal_anneal(self._c, seed, _ptr(p, C.c_float), C.byref(err), C.byref(ev)), "m3dcal_anneal")
        return p, err.value, ev.value


def _mul4_f32(A, B):
    """A * B in float, the sums in the order ros/gpu_6dslam_node.cpp forms them (s = 0; s += A[r][k] * B[k][c], k = 0 .. 3)"""
    A32, B32 = np.asarray(A, np.float64).astype(np.float32), np.asarray(B, np.float64).astype(np.float32)
    out = np.zeros((4, 4), np.float32)
    for c in range(4):
        for r in range(4):
            acc = np.float32(0.0)
            for k in range(4):
                acc = np.float32(acc + np.float32(A32[r, k] * B32[k, c]))
            out[r, c] = acc
    return out.astype(np.float64)


class Gpu6dSlamNode:
    """Host-side mirror of the node the reference launches as `gpu_6dslam_node`
    (m3d_husky_bringup.launch:13): it receives the aggregator's clouds one at a time (queue depth 1,
    m3d_aggregator.cpp:174) and registers every cloud against the previous one, chaining the poses
    into an odometry estimate. The ROS wiring itself (subscriber, tf broadcaster) is the source-only
    shim in ros/gpu_6dslam_node.cpp; this class is what tests and bench drive."""

    def __init__(self, params=None, device=0, mode="scan_to_scan", map_leaf=0.05, map_capacity=1 << 22, loop_params=None, loop_min_corr=2000,
                 loop_max_rms=0.05, aggregate_on_device=False, bbox=(1.0, -1.0, 1.0, -1.0, 1.0, -1.0), aggregate_capacity=1 << 21, scan_trig_float=False):
        """mode: "scan_to_scan" (every sweep against the previous one, poses chained), "scan_to_map" (every sweep against the
        voxel-deduplicated map of all earlier sweeps, kept in HBM: SURVEY §8 row f4) or "slam" (scan-to-scan odometry, every registered sweep a
        keyframe, loop-closure candidates of each new keyframe registered in one batch: self.closures).
        aggregate_on_device: the node takes the aggregator's INPUTS (on_scan / on_laser_cloud with the tf of each message) and aggregates on the
        device: sweeps are born in HBM."""
        self.reg = Registrar(params, device)
        self.reg.set_latency_mode(True)
        self.mode = mode
        self.pose = np.eye(4)          # pose of the latest cloud in the frame of the first
        self.last_delta = np.eye(4)    # constant-velocity prior for the next registration
        self._prev = None              # scan_to_scan: the previous sweep, already bucketed: target of the next registration
        self._map = Map(self.reg, map_leaf, map_capacity) if mode == "scan_to_map" else None
        self._loop = LoopCloser(self.reg, loop_params) if mode == "slam" else None
        self._gate = (loop_min_corr, loop_max_rms)
        self.keyframes = []            # slam: (Cloud, pose) of every keyframe
        self.closures = []             # slam: (source, target, T source -> target, Stats) of every accepted loop closure
        self._agg = Aggregator(self.reg, bbox, aggregate_capacity) if aggregate_on_device else None
        if self._agg is not None:
            self._agg.set_rearm(False)      # like the reference's node: idle after a sweep until ~request
            if scan_trig_float:
                self._agg.set_scan_trig(True)
        self.history = []

    def on_cloud(self, msg: PointCloud2):
        """Topic callback for `/m3d_test/aggregator/cloud`. Returns (pose 4x4, Stats or None). Like the shim, the message
        crosses the ABI with its own field table and every sweep is bucketed once."""
        return self.on_sweep(self.reg.cloud_pc2(msg, source_only=self._map is not None))   # scan-to-map: a sweep is only ever a source and a map insert

    # ---- aggregate_on_device: the aggregator's callbacks (m3d_aggregator.cpp:231-288), the sweep never leaves HBM ------------------------------
    def on_scan(self, ranges, angle_min, angle_increment, tf7):
        self._agg.add_scan(ranges, angle_min, angle_increment, tf7)
        return self._after_message()

    def on_laser_cloud(self, msg: PointCloud2, tf7):
        self._agg.add_cloud(msg, tf7)
        return self._after_message()

    def on_request(self):
        self._agg.restart()

    def _after_message(self):
        if not self._agg.status()["ready"]:
            return None
        return self.on_sweep(self._agg.take_cloud())

    # ---- one sweep, bucketed and resident ------------------------------------------------------------------------------------------------------------
    def on_sweep(self, cur: Cloud):
        if self._map is not None:
            if len(self._map) == 0:
                self._map.insert(cur, self.pose)
                self.history.append((self.pose.copy(), None))
                return self.pose.copy(), None
            tgt = self._map.as_cloud()                       # the map so far, bucketed in place
            T, st = self.reg.align(cur, tgt, self.pose @ self.last_delta)   # prior: constant velocity in the map frame
            if st.status in (abi.CONVERGED, abi.MAX_ITERATIONS):
                self.last_delta = np.linalg.inv(self.pose) @ T
                self.pose = T
                self._map.insert(cur, T)                     # only points in still-empty voxels are kept
            self.history.append((self.pose.copy(), st))
            return self.pose.copy(), st
        if self._prev is None:
            self._prev = cur
            if self._loop is not None:
                self._add_keyframe(cur)
            self.history.append((self.pose.copy(), None))
            return self.pose.copy(), None
        T, st = self.reg.align(cur, self._prev, self.last_delta)
        if st.status in (abi.CONVERGED, abi.MAX_ITERATIONS):
            self.last_delta = T
            self.pose = _mul4_f32(self.pose, T)   # (float, operation for operation like the shim's pose_: the keyframes' signatures are built from these bits)
            if self._loop is not None:
                self._add_keyframe(cur)
        elif self._loop is not None:                         # slam: a sweep the odometry could not place is no keyframe
            self.history.append((self.pose.copy(), st))
            return self.pose.copy(), st
        self._prev = cur
        self.history.append((self.pose.copy(), st))
        return self.pose.copy(), st

    def _add_keyframe(self, cur: Cloud):
        k = self._loop.add_keyframe(cur, self.pose)
        self.keyframes.append((cur, self.pose.copy()))
        cands = self._loop.candidates(k, 1)
        n = len(cands)
        if n == 0:
            return
        out = np.zeros((n, 16), np.float32)
        st = (abi.Stats * n)()
        self.reg._check(lib().m3dreg_align_batch(self.reg._h, self._loop.pairs(cands), n, _ptr(out, C.c_float), st), "align_batch")
        acc = LoopCloser.gate(cands, list(st), *self._gate)
        for i in range(n):
            if acc[i]:
                self.closures.append((cands[i].source, cands[i].target, colmajor16_to_T(out[i]), st[i]))
