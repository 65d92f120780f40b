"""ctypes mirror of include/m3dreg.h (structures, enums, error text).

The same structure layouts are used by oracle/orc.py, so one Params object can be handed to the
HIP library and to the CPU oracle in the parity tests.
"""
import ctypes as C

ABI_VERSION = 8
MAX_LEVELS = 4
NSUMS = 29

# m3dreg_error
OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_GRID_TOO_LARGE = -4
ERR_EMPTY_CLOUD = -5
ERR_NO_TARGET = -6
ERR_LEVEL_MISMATCH = -7
ERR_OUT_OF_MEMORY = -8
ERROR_NAMES = {
    OK: "M3DREG_OK", ERR_INVALID_ARG: "M3DREG_ERR_INVALID_ARG", ERR_NO_DEVICE: "M3DREG_ERR_NO_DEVICE",
    ERR_HIP: "M3DREG_ERR_HIP", ERR_GRID_TOO_LARGE: "M3DREG_ERR_GRID_TOO_LARGE",
    ERR_EMPTY_CLOUD: "M3DREG_ERR_EMPTY_CLOUD", ERR_NO_TARGET: "M3DREG_ERR_NO_TARGET",
    ERR_LEVEL_MISMATCH: "M3DREG_ERR_LEVEL_MISMATCH", ERR_OUT_OF_MEMORY: "M3DREG_ERR_OUT_OF_MEMORY",
}

# m3dreg_metric
POINT_TO_POINT = 0
POINT_TO_PLANE = 1

# m3dreg_status
CONVERGED = 0
MAX_ITERATIONS = 1
TOO_FEW_CORR = 2
RANK_DEFICIENT = 3
DIVERGED = 4
BAD_CLOUD = 5   # a cloud of the pair came out of the asynchronous bucketing in error (Cloud.status() says which)
STATUS_NAMES = {0: "converged", 1: "max_iterations", 2: "too_few_corr", 3: "rank_deficient", 4: "diverged"}


class Params(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32),
        ("leaf", C.c_float * MAX_LEVELS),
        ("iterations", C.c_int32 * MAX_LEVELS),
        ("max_corr_dist", C.c_float * MAX_LEVELS),
        ("metric", C.c_int32),
        ("min_correspondences", C.c_int32),
        ("eps_rot", C.c_double),
        ("eps_trans", C.c_double),
        ("pivot_rel_tol", C.c_double),
        ("plane_ratio", C.c_float),
        ("normal_min_pts", C.c_int32),
        ("normal_leaf", C.c_float),
        ("normal_min_spread", C.c_float),
    ]

    @classmethod
    def make(cls, leaf=0.1, iterations=30, max_corr_dist=0.5, metric=POINT_TO_PLANE, min_correspondences=10,
             eps_rot=1e-5, eps_trans=1e-5, pivot_rel_tol=1e-9, plane_ratio=0.25, normal_min_pts=5,
             normal_leaf=0.4, normal_min_spread=0.25):
        """leaf / iterations / max_corr_dist may be scalars (one level) or equal-length sequences
        (coarse -> fine)."""
        def seq(v):
            return list(v) if isinstance(v, (list, tuple)) else [v]
        leaf, iterations, max_corr_dist = seq(leaf), seq(iterations), seq(max_corr_dist)
        n = len(leaf)
        if len(iterations) == 1:
            iterations = iterations * n
        if len(max_corr_dist) == 1:
            max_corr_dist = max_corr_dist * n
        if not (1 <= n <= MAX_LEVELS and len(iterations) == n and len(max_corr_dist) == n):
            raise ValueError("bad level specification")
        p = cls()
        p.n_levels = n
        for i in range(n):
            p.leaf[i] = leaf[i]
            p.iterations[i] = iterations[i]
            p.max_corr_dist[i] = max_corr_dist[i]
        p.metric = metric
        p.min_correspondences = min_correspondences
        p.eps_rot, p.eps_trans, p.pivot_rel_tol = eps_rot, eps_trans, pivot_rel_tol
        p.plane_ratio, p.normal_min_pts = plane_ratio, normal_min_pts
        p.normal_leaf, p.normal_min_spread = normal_leaf, normal_min_spread
        return p


class Stats(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("iterations", C.c_int32),
        ("n_corr", C.c_int64),
        ("rms", C.c_double),
        ("last_rot", C.c_double),
        ("last_trans", C.c_double),
    ]

    def as_dict(self):
        return {"status": STATUS_NAMES.get(self.status, self.status), "iterations": self.iterations,
                "n_corr": self.n_corr, "rms": self.rms, "last_rot": self.last_rot, "last_trans": self.last_trans}


class GridInfo(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("n_valid", C.c_int32), ("n_cells", C.c_int32),
        ("dims", C.c_int32 * 3), ("bits", C.c_int32 * 3),
        ("mn", C.c_float * 3), ("mx", C.c_float * 3), ("center", C.c_float * 3),
        ("leaf", C.c_float), ("inv_leaf", C.c_float), ("lbound", C.c_float),
        ("has_normals", C.c_int32),
    ]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if hasattr(getattr(self, k), "__len__") else getattr(self, k))
                for k, _ in self._fields_}


class CloudDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_size_t), ("point_step", C.c_size_t), ("off_x", C.c_size_t), ("off_y", C.c_size_t),
                ("off_z", C.c_size_t), ("data_is_device", C.c_int32), ("source_only", C.c_int32)]


class PointField(C.Structure):   # m3dreg_point_field
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint32), ("datatype", C.c_uint8), ("count", C.c_uint32)]


class Pair(C.Structure):
    _fields_ = [("source", C.c_void_p), ("target", C.c_void_p), ("init_T", C.c_float * 16)]


class PairDesc(C.Structure):   # m3dreg_pair_desc (m3dreg_multi_align)
    _fields_ = [("source", CloudDesc), ("target", CloudDesc), ("init_T", C.c_float * 16), ("target_group", C.c_int32), ("reserved", C.c_int32)]


class LoopParams(C.Structure):   # m3dloop_params
    _fields_ = [("sig_leaf", C.c_float), ("sig_log2_bits", C.c_int32), ("radius", C.c_float), ("min_gap", C.c_int32), ("top_k", C.c_int32),
                ("min_overlap", C.c_float), ("max_keyframes", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def make(cls, sig_leaf=2.0, sig_log2_bits=16, radius=10.0, min_gap=10, top_k=2, min_overlap=0.5, max_keyframes=4096):
        p = cls()
        p.sig_leaf, p.sig_log2_bits, p.radius, p.min_gap, p.top_k, p.min_overlap, p.max_keyframes, p.reserved = sig_leaf, sig_log2_bits, radius, min_gap, top_k, min_overlap, max_keyframes, 0
        return p


class LoopCandidate(C.Structure):   # m3dloop_candidate
    _fields_ = [("source", C.c_int32), ("target", C.c_int32), ("overlap", C.c_uint32), ("pop_source", C.c_uint32), ("pop_target", C.c_uint32),
                ("dist2", C.c_float), ("init_T", C.c_float * 16)]

    def as_tuple(self):
        return (self.source, self.target, self.overlap, self.pop_source, self.pop_target, bytes(C.c_float(self.dist2)), bytes(self.init_T))


class M3dregError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {ERROR_NAMES.get(code, code)}{(' — ' + detail) if detail else ''}")
