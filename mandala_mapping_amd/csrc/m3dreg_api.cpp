// m3dreg_api.cpp — the C ABI of include/m3dreg.h on top of the HIP kernels (bucket.hip, icp.hip).
//
// Host-side role: own the device memory of bucketed clouds, derive every grid / fixed-point
// parameter from the exact AABB with the operation order DESIGN.md §Spec fixes (this file is built
// with -ffp-contract=off like the kernels), enqueue the launches of a registration without any
// host synchronisation between Gauss-Newton iterations, and translate failures into int codes.
// There is deliberately no CPU implementation behind this ABI: if HIP is unusable every entry
// point fails (M3DREG_ERR_NO_DEVICE / M3DREG_ERR_HIP).
#include "../../include/m3dreg.h"
#include "m3d_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct DevLevel {
    M3dGrid grid{};
    int32_t bits[3]{};
    float mx[3]{};
    float lbound = 0.f;
    float4* pts = nullptr;
    M3dBucket* htab = nullptr;
    uint32_t* bigcum = nullptr;
    uint32_t bigcap = 0;
    uint32_t hcap = 0;             // allocated entries (worst case); the used size lives in n_cells[1..2]
    uint32_t n_cells_host = 0;
    uint32_t* keys = nullptr;
    uint32_t* skey = nullptr;
    uint32_t* perm = nullptr;
    uint32_t* n_cells = nullptr;   // device: {occupied voxels, hmask, hshift, occupied buckets, big buckets, ...}
};

}  // namespace

struct m3dreg_cloud {
    int32_t n = 0;
    int32_t n_valid = 0;
    int32_t n_levels = 0;
    float leaf[M3DREG_MAX_LEVELS]{};
    bool has_normals = false;
    float *x = nullptr, *y = nullptr, *z = nullptr;
    float4* nrm_in = nullptr;      // normals by input index (shared by all levels)
    float mn[3]{}, mx[3]{};
    DevLevel lv[M3DREG_MAX_LEVELS];
    std::vector<void*> allocs;
};

struct m3dreg_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    m3dreg_params params{};
    std::string err;
    // sort workspace (grown on demand)
    size_t ws_n = 0;
    uint32_t *ka = nullptr, *va = nullptr, *kb = nullptr, *vb = nullptr, *hist = nullptr, *aabb = nullptr;
    long long* mom = nullptr;          // [10 * ws_n] per-voxel moments of the normal grid
    // batch state
    size_t cap_pairs = 0;
    M3dJob* d_jobs = nullptr;          // [levels][cap_pairs]
    M3dPairState* d_states = nullptr;  // [cap_pairs]
    double* d_trace = nullptr;         // [M3D_MAX_TRACE][16], first pair only
    M3dJob* h_jobs = nullptr;          // pinned
    M3dPairState* h_states = nullptr;  // pinned
    double* h_trace = nullptr;         // pinned
    size_t pending_pairs = 0;
    size_t last_trace_n = 0;
    // gpu_6dslam_node surface
    m3dreg_cloud* target = nullptr;
    int icp_variant = 2;               // 2 = split search/reduce kernels (default), 1 = fused LDS-staged, 0 = fused per-thread (M3DREG_ICP_VARIANT)
    int* d_match = nullptr;            // [match_pairs * match_stride] NN result per query (variant 2)
    size_t match_cap = 0;
    int match_stride = 0;
    // measurement: event pairs around the dominant kernel
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    uint64_t prof_launches = 0;
    double prof_ms = 0.0;
};

namespace {

int fail(m3dreg_handle* h, int code, const char* where, hipError_t e = hipSuccess) {
    if (h) {
        char buf[256];
        if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
        else snprintf(buf, sizeof(buf), "%s", where);
        h->err = buf;
    }
    return code;
}

#define HIPCHK(h, call)                                                   \
    do {                                                                  \
        hipError_t _e = (call);                                           \
        if (_e != hipSuccess) return fail((h), M3DREG_ERR_HIP, #call, _e); \
    } while (0)

int bits_for(int32_t d) {
    int b = 1;
    while ((int64_t(1) << b) < int64_t(d)) b++;
    return b;
}

int ceil_log2_d(double x) {
    int ex;
    double m = std::frexp(x, &ex);
    return (m == 0.5) ? ex - 1 : ex;
}

float cell_f(float v, float mn, float inv_leaf) {
    float d = v - mn;
    float s = d * inv_leaf;
    return std::floor(s);
}

// Spec §Grid: everything the kernels need, from the exact AABB. Returns an m3dreg_error.
int make_grid(const float mn[3], const float mx[3], float leaf, int32_t n, int32_t n_valid, DevLevel& L) {
    M3dGrid& g = L.grid;
    g.leaf = leaf;
    g.inv_leaf = 1.0f / leaf;
    g.n_valid = n_valid;
    float half_max = 0.0f, amax = 0.0f, ext_max = 0.0f;
    int total_bits = 0;
    for (int a = 0; a < 3; a++) {
        g.mn[a] = mn[a];
        L.mx[a] = mx[a];
        float fc = cell_f(mx[a], mn[a], g.inv_leaf);
        if (!(fc < 1073741824.0f)) return M3DREG_ERR_GRID_TOO_LARGE;
        g.dims[a] = int32_t(fc) + 1;
        L.bits[a] = bits_for((g.dims[a] + 1) >> 1);   // bit width of the BUCKET coordinate
        if (L.bits[a] > 11) return M3DREG_ERR_GRID_TOO_LARGE;
        g.cb[a] = L.bits[a];
        total_bits += L.bits[a];
        float ext = mx[a] - mn[a];
        float half = ext * 0.5f;
        g.center[a] = mn[a] + half;
        if (half > half_max) half_max = half;
        amax = std::fmax(amax, std::fmax(std::fabs(mn[a]), std::fabs(mx[a])));
        ext_max = std::fmax(ext_max, ext);
    }
    if (total_bits + 3 > 31) return M3DREG_ERR_GRID_TOO_LARGE;
    L.lbound = half_max + 3.0f * leaf;
    // hash table: worst-case allocation is a power of two >= 2n; the used size (power of two >= 2 * occupied
    // voxels) is derived on the device after the sort and read back once at the end of cloud_create
    uint32_t hs = 16;
    while (hs < 2u * uint32_t(n)) hs <<= 1;
    L.hcap = hs;
    g.hmask = 0;
    g.hshift = 0;
    // pruning slack (not part of the results: only makes the box test conservative)
    g.prune_slack = 1.0e-6f * (amax + ext_max) + 1.0e-3f * leaf;
    return M3DREG_OK;
}

void fixed_exps(float lbound, float max_corr_dist, int32_t e[6]) {
    const double lb = double(lbound), D = double(max_corr_dist) * 1.001;
    e[0] = 30 - ceil_log2_d(3.0 * lb * lb);
    e[1] = 30 - ceil_log2_d(1.7320508075688772 * lb);
    e[2] = 30;
    e[3] = 30 - ceil_log2_d(1.7320508075688772 * lb * D);
    e[4] = 30 - ceil_log2_d(D);
    e[5] = 30 - ceil_log2_d(D * D);
}

template <typename T>
int dmalloc(m3dreg_handle* h, m3dreg_cloud* c, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) return fail(h, M3DREG_ERR_HIP, "hipMalloc", e);
    if (c) c->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return M3DREG_OK;
}

int ensure_workspace(m3dreg_handle* h, size_t n) {
    if (!h->aabb) { int rc = dmalloc(h, nullptr, &h->aabb, 8); if (rc) return rc; }
    if (n <= h->ws_n) return M3DREG_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (void* p : { (void*)h->ka, (void*)h->va, (void*)h->kb, (void*)h->vb, (void*)h->hist, (void*)h->mom }) if (p) hipFree(p);
    h->ka = h->va = h->kb = h->vb = h->hist = nullptr; h->mom = nullptr;
    h->ws_n = 0;
    size_t cap = n + n / 4 + 1024;
    int rc;
    if ((rc = dmalloc(h, nullptr, &h->ka, cap)) || (rc = dmalloc(h, nullptr, &h->va, cap)) || (rc = dmalloc(h, nullptr, &h->kb, cap)) ||
        (rc = dmalloc(h, nullptr, &h->vb, cap)) || (rc = dmalloc(h, nullptr, &h->hist, 256 * size_t(m3d_sort_tiles(int(cap)) + 1))) ||
        (rc = dmalloc(h, nullptr, &h->mom, 10 * cap)))
        return rc;
    h->ws_n = cap;
    return M3DREG_OK;
}

void free_cloud(m3dreg_cloud* c) {
    if (!c) return;
    for (void* p : c->allocs) hipFree(p);
    delete c;
}

int sort_passes_for(const DevLevel& L, bool has_invalid) {
    if (has_invalid) return 4;  // the 0xFFFFFFFF keys of non-finite points must end up last
    int bits = L.bits[0] + L.bits[1] + L.bits[2] + 3;
    return (bits + 7) / 8;
}

// bucket one level of `c` (geometry from the cloud's AABB)
int bucket_level(m3dreg_handle* h, m3dreg_cloud* c, DevLevel& L, float leaf) {
    int rc = make_grid(c->mn, c->mx, leaf, c->n, c->n_valid, L);
    if (rc) return fail(h, rc, "voxel grid needs more than 31 key bits (coarsen leaf or crop the cloud)");
    const size_t n = size_t(c->n);
    if ((rc = dmalloc(h, c, &L.pts, n)) || (rc = dmalloc(h, c, &L.htab, size_t(L.hcap))) || (rc = dmalloc(h, c, &L.keys, n)) ||
        (rc = dmalloc(h, c, &L.skey, n)) || (rc = dmalloc(h, c, &L.perm, n)) || (rc = dmalloc(h, c, &L.n_cells, 8)))
        return rc;
    L.bigcap = uint32_t(n / 65536 + 1);
    if ((rc = dmalloc(h, c, &L.bigcum, size_t(L.bigcap) * 8))) return rc;
    HIPCHK(h, hipMemsetAsync(L.bigcum, 0, sizeof(uint32_t) * 8 * size_t(L.bigcap), h->stream));
    M3dBucketArgs a{};
    a.n = c->n; a.x = c->x; a.y = c->y; a.z = c->z; a.grid = L.grid;
    a.sort_passes = sort_passes_for(L, c->n_valid != c->n);
    a.keys = L.keys; a.ka = h->ka; a.va = h->va; a.kb = h->kb; a.vb = h->vb; a.hist = h->hist;
    a.skey_out = L.skey; a.perm_out = L.perm; a.pts = L.pts; a.htab = L.htab; a.hcap = L.hcap; a.bigcum = L.bigcum; a.bigcap = L.bigcap;
    a.n_cells = L.n_cells;
    HIPCHK(h, m3d_launch_bucket_level(h->stream, a));
    return M3DREG_OK;
}

M3dLevelDev level_dev(const DevLevel& L, const float4* nrm_in) {
    M3dLevelDev d{};
    d.pts = L.pts; d.nrm = nrm_in; d.htab = L.htab; d.bigcum = L.bigcum; d.g = L.grid;
    return d;
}

int check_levels(m3dreg_handle* h, const m3dreg_cloud* c) {
    if (c->n_levels != h->params.n_levels) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "cloud bucketed with a different number of levels");
    for (int l = 0; l < c->n_levels; l++)
        if (c->leaf[l] != h->params.leaf[l]) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "cloud bucketed with different leaf sizes");
    return M3DREG_OK;
}

int ensure_batch(m3dreg_handle* h, size_t n_pairs) {
    if (n_pairs <= h->cap_pairs) return M3DREG_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_jobs) hipFree(h->d_jobs);
    if (h->d_states) hipFree(h->d_states);
    if (h->h_jobs) hipHostFree(h->h_jobs);
    if (h->h_states) hipHostFree(h->h_states);
    h->d_jobs = nullptr; h->d_states = nullptr; h->h_jobs = nullptr; h->h_states = nullptr; h->cap_pairs = 0;
    size_t cap = n_pairs < 8 ? 8 : n_pairs;
    HIPCHK(h, hipMalloc((void**)&h->d_jobs, sizeof(M3dJob) * cap * M3DREG_MAX_LEVELS));
    HIPCHK(h, hipMalloc((void**)&h->d_states, sizeof(M3dPairState) * cap));
    HIPCHK(h, hipHostMalloc((void**)&h->h_jobs, sizeof(M3dJob) * cap * M3DREG_MAX_LEVELS, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void**)&h->h_states, sizeof(M3dPairState) * cap, hipHostMallocDefault));
    if (!h->d_trace) {
        HIPCHK(h, hipMalloc((void**)&h->d_trace, sizeof(double) * 16 * M3D_MAX_TRACE));
        HIPCHK(h, hipHostMalloc((void**)&h->h_trace, sizeof(double) * 16 * M3D_MAX_TRACE, hipHostMallocDefault));
    }
    h->cap_pairs = cap;
    return M3DREG_OK;
}

int ensure_match(m3dreg_handle* h, size_t n_pairs, int max_n_src) {
    const size_t stride = (size_t(max_n_src) + 63) & ~size_t(63);
    if (n_pairs * stride > h->match_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_match) hipFree(h->d_match);
        h->d_match = nullptr; h->match_cap = 0;
        const size_t cap = n_pairs * stride + n_pairs * stride / 4;
        HIPCHK(h, hipMalloc((void**)&h->d_match, sizeof(int) * cap));
        h->match_cap = cap;
    }
    h->match_stride = int(stride);
    return M3DREG_OK;
}

int validate_params(const m3dreg_params* p) {
    if (!p || p->n_levels < 1 || p->n_levels > M3DREG_MAX_LEVELS) return M3DREG_ERR_INVALID_ARG;
    for (int l = 0; l < p->n_levels; l++)
        if (!(p->leaf[l] > 0.f) || !(p->max_corr_dist[l] > 0.f) || p->iterations[l] < 0 || !std::isfinite(p->leaf[l]) ||
            !std::isfinite(p->max_corr_dist[l]))
            return M3DREG_ERR_INVALID_ARG;
    if (p->metric != M3DREG_POINT_TO_POINT && p->metric != M3DREG_POINT_TO_PLANE) return M3DREG_ERR_INVALID_ARG;
    if (p->metric == M3DREG_POINT_TO_PLANE && (!(p->normal_leaf > 0.f) || !(p->plane_ratio > 0.f))) return M3DREG_ERR_INVALID_ARG;
    return M3DREG_OK;
}

// fill one job per (level, pair) + initial state
int build_jobs(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs, int& max_n_src) {
    const m3dreg_params& P = h->params;
    max_n_src = 0;
    for (size_t i = 0; i < n_pairs; i++) {
        const m3dreg_cloud* s = pairs[i].source;
        const m3dreg_cloud* t = pairs[i].target;
        if (!s || !t) return fail(h, M3DREG_ERR_INVALID_ARG, "null cloud in pair");
        int rc = check_levels(h, t);
        if (rc) return rc;
        if (P.metric == M3DREG_POINT_TO_PLANE && !t->has_normals) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "target cloud has no normals");
        if (s->n_valid > max_n_src) max_n_src = s->n_valid;
        for (int l = 0; l < P.n_levels; l++) {
            M3dJob& J = h->h_jobs[size_t(l) * h->cap_pairs + i];
            memset(&J, 0, sizeof(J));
            J.src = s->lv[s->n_levels - 1].pts; J.n_src = s->n_valid; J.metric = P.metric;
            J.tgt = level_dev(t->lv[l], t->nrm_in);
            J.dmax2 = P.max_corr_dist[l] * P.max_corr_dist[l];
            fixed_exps(t->lv[l].lbound, P.max_corr_dist[l], J.exps);
            for (int k = 0; k < 6; k++) J.S[k] = std::ldexp(1.0f, J.exps[k]);
            J.min_corr = P.min_correspondences;
            J.last_level = (l == P.n_levels - 1) ? 1 : 0;
            J.eps_rot2 = P.eps_rot * P.eps_rot;
            J.eps_trans2 = P.eps_trans * P.eps_trans;
            J.pivot_rel_tol = P.pivot_rel_tol;
            J.st = h->d_states + i;
            J.trace = (i == 0) ? h->d_trace : nullptr;
        }
        M3dPairState& S = h->h_states[i];
        memset(&S, 0, sizeof(S));
        for (int k = 0; k < 16; k++) S.T[k] = double(pairs[i].init_T[k]);
        S.status = M3DREG_MAX_ITERATIONS;
    }
    return M3DREG_OK;
}

hipEvent_t next_event(m3dreg_handle* h) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}

// fold the recorded event pairs into the running totals (requires the stream to be idle)
void drain_events(m3dreg_handle* h) {
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i], h->ev_pool[i + 1]) == hipSuccess) { h->prof_ms += double(ms); h->prof_launches++; }
    }
    h->ev_used = 0;
}

void stats_from_state(const M3dPairState& S, m3dreg_stats* st) {
    st->status = S.status;
    st->iterations = S.iters;
    st->n_corr = S.n_corr;
    st->rms = S.n_corr > 0 ? std::sqrt(std::ldexp(double(S.ssr), -S.ssr_exp) / double(S.n_corr)) : 0.0;
    st->last_rot = std::sqrt(S.th2);
    st->last_trans = std::sqrt(S.tr2);
}

}  // namespace

extern "C" {

int m3dreg_abi_version(void) { return M3DREG_ABI_VERSION; }
const char* m3dreg_backend_name(void) { return "hip-gfx950"; }
const char* m3dreg_last_error(const m3dreg_handle* h) { return h ? h->err.c_str() : "null handle"; }

int m3dreg_default_params(m3dreg_params* p) {
    if (!p) return M3DREG_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->n_levels = 1; p->leaf[0] = 0.1f; p->iterations[0] = 30; p->max_corr_dist[0] = 0.5f;
    p->metric = M3DREG_POINT_TO_PLANE; p->min_correspondences = 10;
    p->eps_rot = 1e-5; p->eps_trans = 1e-5; p->pivot_rel_tol = 1e-9;
    p->plane_ratio = 0.25f; p->normal_min_pts = 5; p->normal_leaf = 0.4f; p->normal_min_spread = 0.25f;
    return M3DREG_OK;
}

int m3dreg_create(const m3dreg_params* params, int device, void* stream, m3dreg_handle** out) {
    if (!out) return M3DREG_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = validate_params(params);
    if (rc) return rc;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return M3DREG_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return M3DREG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return M3DREG_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return M3DREG_ERR_NO_DEVICE;  // the code object is gfx950-only
    m3dreg_handle* h = new m3dreg_handle();
    h->device = device;
    h->params = *params;
    if (const char* v = getenv("M3DREG_ICP_VARIANT")) { int q = atoi(v); h->icp_variant = (q >= 0 && q <= 2) ? q : 2; }
    if (stream) { h->stream = static_cast<hipStream_t>(stream); h->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return M3DREG_ERR_HIP; }
        h->own_stream = true;
    }
    *out = h;
    return M3DREG_OK;
}

int m3dreg_destroy(m3dreg_handle* h) {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    free_cloud(h->target);
    for (void* p : { (void*)h->ka, (void*)h->va, (void*)h->kb, (void*)h->vb, (void*)h->hist, (void*)h->mom, (void*)h->aabb, (void*)h->d_jobs, (void*)h->d_states,
                     (void*)h->d_trace, (void*)h->d_match })
        if (p) hipFree(p);
    for (void* p : { (void*)h->h_jobs, (void*)h->h_states, (void*)h->h_trace }) if (p) hipHostFree(p);
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
    return M3DREG_OK;
}

void* m3dreg_get_stream(m3dreg_handle* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

int m3dreg_synchronize(m3dreg_handle* h) {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return M3DREG_OK;
}

int m3dreg_cloud_create(m3dreg_handle* h, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                        int data_is_device, m3dreg_cloud** out) {
    if (!h || !data || !out || n == 0 || n >= 0x7FFFFFFFull) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create: bad argument");
    if (off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step || point_step > 0x7FFFFFFFull)
        return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create: field offsets outside point_step");
    *out = nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    const m3dreg_params& P = h->params;
    m3dreg_cloud* c = new m3dreg_cloud();
    c->n = int32_t(n);
    c->n_levels = P.n_levels;
    for (int l = 0; l < P.n_levels; l++) c->leaf[l] = P.leaf[l];
    int rc;
#define CLOUD_TRY(expr) do { rc = (expr); if (rc) { hipStreamSynchronize(h->stream); free_cloud(c); return rc; } } while (0)
#define CLOUD_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { hipStreamSynchronize(h->stream); free_cloud(c); return fail(h, M3DREG_ERR_HIP, #expr, _e); } } while (0)
    CLOUD_TRY(ensure_workspace(h, n));
    // stage the payload on the device (a2). Device code reads 4-byte aligned floats; anything else is
    // repacked on the host first (never seen from m3d_aggregator, whose layout is 16/0/4/8).
    const uint8_t* raw_d = nullptr;
    uint8_t* staged = nullptr;
    std::vector<float> repack;
    const bool aligned = (point_step % 4 == 0) && (off_x % 4 == 0) && (off_y % 4 == 0) && (off_z % 4 == 0) && (reinterpret_cast<uintptr_t>(data) % 4 == 0);
    if (data_is_device) {
        if (!aligned) { free_cloud(c); return fail(h, M3DREG_ERR_INVALID_ARG, "device payloads must be 4-byte aligned"); }
        raw_d = static_cast<const uint8_t*>(data);
    } else {
        const void* src = data;
        size_t bytes = n * point_step;
        if (!aligned) {
            repack.resize(3 * n);
            const uint8_t* b = static_cast<const uint8_t*>(data);
            for (size_t i = 0; i < n; i++) {
                memcpy(&repack[3 * i], b + i * point_step + off_x, 4);
                memcpy(&repack[3 * i + 1], b + i * point_step + off_y, 4);
                memcpy(&repack[3 * i + 2], b + i * point_step + off_z, 4);
            }
            src = repack.data(); bytes = 12 * n; point_step = 12; off_x = 0; off_y = 4; off_z = 8;
        }
        CLOUD_HIP(hipMalloc((void**)&staged, bytes));
        hipError_t e = hipMemcpyAsync(staged, src, bytes, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) { hipFree(staged); CLOUD_HIP(e); }
        raw_d = staged;
    }
    auto drop_staged = [&]() { if (staged) { hipStreamSynchronize(h->stream); hipFree(staged); staged = nullptr; } };
#undef CLOUD_TRY
#undef CLOUD_HIP
#define CLOUD_TRY(expr) do { rc = (expr); if (rc) { drop_staged(); hipStreamSynchronize(h->stream); free_cloud(c); return rc; } } while (0)
#define CLOUD_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { drop_staged(); hipStreamSynchronize(h->stream); free_cloud(c); return fail(h, M3DREG_ERR_HIP, #expr, _e); } } while (0)
    CLOUD_TRY(dmalloc(h, c, &c->x, n));
    CLOUD_TRY(dmalloc(h, c, &c->y, n));
    CLOUD_TRY(dmalloc(h, c, &c->z, n));
    CLOUD_HIP(m3d_launch_decode_aabb(h->stream, raw_d, int(n), int(point_step), int(off_x), int(off_y), int(off_z), c->x, c->y, c->z, h->aabb));
    uint32_t ab[8];
    CLOUD_HIP(hipMemcpyAsync(ab, h->aabb, sizeof(ab), hipMemcpyDeviceToHost, h->stream));
    CLOUD_HIP(hipStreamSynchronize(h->stream));   // the only host sync of the bucketing: grid geometry is host-derived
    drop_staged();
    c->n_valid = int32_t(ab[6]);
    if (c->n_valid == 0) { free_cloud(c); return fail(h, M3DREG_ERR_EMPTY_CLOUD, "cloud has no finite point"); }
    for (int a = 0; a < 3; a++) { c->mn[a] = m3d_unord_f32(ab[a]); c->mx[a] = m3d_unord_f32(ab[3 + a]); }
    // a9: normals on the dedicated normal grid (point-to-plane only), kept in input order
    if (P.metric == M3DREG_POINT_TO_PLANE) {
        DevLevel NG;
        CLOUD_TRY(bucket_level(h, c, NG, P.normal_leaf));
        CLOUD_TRY(dmalloc(h, c, &c->nrm_in, n));
        float4* nrm_in = c->nrm_in;
        CLOUD_HIP(m3d_launch_normals(h->stream, level_dev(NG, nullptr), NG.n_cells, NG.skey, h->mom, P.plane_ratio, P.normal_min_pts, P.normal_min_spread, nrm_in, int(n)));
        c->has_normals = true;
    }
    for (int l = 0; l < P.n_levels; l++) CLOUD_TRY(bucket_level(h, c, c->lv[l], P.leaf[l]));
    // read back the device-derived table geometry of every level (the kernels so far took it from device memory)
    uint32_t dyn[M3DREG_MAX_LEVELS][8];
    for (int l = 0; l < P.n_levels; l++) CLOUD_HIP(hipMemcpyAsync(dyn[l], c->lv[l].n_cells, sizeof(dyn[l]), hipMemcpyDeviceToHost, h->stream));
    CLOUD_HIP(hipStreamSynchronize(h->stream));
    for (int l = 0; l < P.n_levels; l++) {
        c->lv[l].n_cells_host = dyn[l][0];
        c->lv[l].grid.hmask = dyn[l][1];
        c->lv[l].grid.hshift = int32_t(dyn[l][2]);
    }
#undef CLOUD_TRY
#undef CLOUD_HIP
    *out = c;
    return M3DREG_OK;
}

int m3dreg_cloud_destroy(m3dreg_handle* h, m3dreg_cloud* c) {
    if (!h || !c) return M3DREG_ERR_INVALID_ARG;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    free_cloud(c);
    return M3DREG_OK;
}

int m3dreg_align_batch_async(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs) {
    if (!h || !pairs || n_pairs == 0 || n_pairs > 65535) return fail(h, M3DREG_ERR_INVALID_ARG, "align_batch: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_batch(h, n_pairs);
    if (rc) return rc;
    int max_n_src = 0;
    if ((rc = build_jobs(h, pairs, n_pairs, max_n_src))) return rc;
    const m3dreg_params& P = h->params;
    if ((rc = ensure_match(h, n_pairs, max_n_src))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_jobs, h->h_jobs, sizeof(M3dJob) * h->cap_pairs * size_t(P.n_levels), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_states, h->h_states, sizeof(M3dPairState) * n_pairs, hipMemcpyHostToDevice, h->stream));
    for (int l = 0; l < P.n_levels; l++) {
        const M3dJob* dj = h->d_jobs + size_t(l) * h->cap_pairs;
        for (int it = 0; it < P.iterations[l]; it++) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (h->profiling) { e0 = next_event(h); e1 = next_event(h); }
            HIPCHK(h, m3d_launch_icp_iteration(h->stream, dj, int(n_pairs), max_n_src, P.metric, it == 0 ? 1 : 0, h->icp_variant, h->d_match, h->match_stride, e0, e1));
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->h_states, h->d_states, sizeof(M3dPairState) * n_pairs, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_trace, h->d_trace, sizeof(double) * 16 * M3D_MAX_TRACE, hipMemcpyDeviceToHost, h->stream));
    h->pending_pairs = n_pairs;
    return M3DREG_OK;
}

int m3dreg_batch_wait(m3dreg_handle* h, float* out_T, m3dreg_stats* stats) {
    if (!h || h->pending_pairs == 0) return fail(h, M3DREG_ERR_INVALID_ARG, "batch_wait: nothing pending");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->ev_used) drain_events(h);
    for (size_t i = 0; i < h->pending_pairs; i++) {
        const M3dPairState& S = h->h_states[i];
        if (out_T) for (int k = 0; k < 16; k++) out_T[16 * i + k] = float(S.T[k]);
        if (stats) stats_from_state(S, &stats[i]);
    }
    int it0 = h->h_states[0].iters;
    h->last_trace_n = size_t(it0 < M3D_MAX_TRACE ? it0 : M3D_MAX_TRACE);
    h->pending_pairs = 0;
    return M3DREG_OK;
}

int m3dreg_align_batch(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs, float* out_T, m3dreg_stats* stats) {
    int rc = m3dreg_align_batch_async(h, pairs, n_pairs);
    if (rc) return rc;
    return m3dreg_batch_wait(h, out_T, stats);
}

int m3dreg_align_clouds(m3dreg_handle* h, const m3dreg_cloud* source, const m3dreg_cloud* target, const float init_T[16], float out_T[16],
                        m3dreg_stats* stats) {
    if (!h || !source || !target || !init_T || !out_T) return fail(h, M3DREG_ERR_INVALID_ARG, "align_clouds: bad argument");
    m3dreg_pair p;
    p.source = source; p.target = target;
    memcpy(p.init_T, init_T, sizeof(float) * 16);
    return m3dreg_align_batch(h, &p, 1, out_T, stats);
}

int m3dreg_set_target_xyz(m3dreg_handle* h, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z) {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    m3dreg_cloud* c = nullptr;
    int rc = m3dreg_cloud_create(h, data, n, point_step, off_x, off_y, off_z, 0, &c);
    if (rc) return rc;
    if (h->target) m3dreg_cloud_destroy(h, h->target);
    h->target = c;
    return M3DREG_OK;
}

int m3dreg_align(m3dreg_handle* h, const void* src, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                 const float init_T[16], float out_T[16], m3dreg_stats* stats) {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    if (!h->target) return fail(h, M3DREG_ERR_NO_TARGET, "m3dreg_align before m3dreg_set_target_xyz");
    m3dreg_cloud* s = nullptr;
    int rc = m3dreg_cloud_create(h, src, n, point_step, off_x, off_y, off_z, 0, &s);
    if (rc) return rc;
    rc = m3dreg_align_clouds(h, s, h->target, init_T, out_T, stats);
    m3dreg_cloud_destroy(h, s);
    return rc;
}

// ---- measurement ----------------------------------------------------------------------------------------
int m3dreg_profile_enable(m3dreg_handle* h, int on) {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    h->profiling = on != 0;
    return M3DREG_OK;
}

int m3dreg_profile_read(m3dreg_handle* h, uint64_t* n_launches, double* total_ms, int reset) {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drain_events(h);
    if (n_launches) *n_launches = h->prof_launches;
    if (total_ms) *total_ms = h->prof_ms;
    if (reset) { h->prof_launches = 0; h->prof_ms = 0.0; }
    return M3DREG_OK;
}

// ---- introspection ------------------------------------------------------------------------------------
int m3dreg_cloud_levels(const m3dreg_cloud* c) { return c ? c->n_levels : M3DREG_ERR_INVALID_ARG; }

int m3dreg_cloud_grid_info(m3dreg_handle* h, const m3dreg_cloud* c, int level, m3dreg_grid_info* out) {
    if (!h || !c || !out || level < 0 || level >= c->n_levels) return fail(h, M3DREG_ERR_INVALID_ARG, "grid_info: bad argument");
    const DevLevel& L = c->lv[level];
    memset(out, 0, sizeof(*out));
    out->n = c->n; out->n_valid = c->n_valid;
    out->n_cells = int32_t(L.n_cells_host);
    for (int a = 0; a < 3; a++) {
        out->dims[a] = L.grid.dims[a]; out->bits[a] = L.bits[a]; out->mn[a] = L.grid.mn[a]; out->mx[a] = L.mx[a]; out->center[a] = L.grid.center[a];
    }
    out->leaf = L.grid.leaf; out->inv_leaf = L.grid.inv_leaf; out->lbound = L.lbound; out->has_normals = c->has_normals ? 1 : 0;
    return M3DREG_OK;
}

int m3dreg_cloud_export(m3dreg_handle* h, const m3dreg_cloud* c, int level, uint32_t* keys, uint32_t* sorted_keys, int32_t* perm,
                        float* sorted_xyz, float* normals) {
    if (!h || !c || level < 0 || level >= c->n_levels) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_export: bad argument");
    if (normals && !c->has_normals) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_export: cloud has no normals");
    const DevLevel& L = c->lv[level];
    const size_t n = size_t(c->n);
    HIPCHK(h, hipSetDevice(h->device));
    if (keys) HIPCHK(h, hipMemcpyAsync(keys, L.keys, 4 * n, hipMemcpyDeviceToHost, h->stream));
    if (sorted_keys) HIPCHK(h, hipMemcpyAsync(sorted_keys, L.skey, 4 * n, hipMemcpyDeviceToHost, h->stream));
    if (perm) HIPCHK(h, hipMemcpyAsync(perm, L.perm, 4 * n, hipMemcpyDeviceToHost, h->stream));
    float *dx = nullptr, *dn = nullptr;
    if (sorted_xyz || normals) {
        HIPCHK(h, hipMalloc((void**)&dx, 12 * n));
        if (normals) { hipError_t e = hipMalloc((void**)&dn, 12 * n); if (e != hipSuccess) { hipFree(dx); return fail(h, M3DREG_ERR_HIP, "hipMalloc", e); } }
        hipError_t e = m3d_launch_export_sorted(h->stream, L.pts, normals ? c->nrm_in : nullptr, int(n), dx, dn);
        if (e == hipSuccess && sorted_xyz) e = hipMemcpyAsync(sorted_xyz, dx, 12 * n, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && normals) e = hipMemcpyAsync(normals, dn, 12 * n, hipMemcpyDeviceToHost, h->stream);
        hipError_t e2 = hipStreamSynchronize(h->stream);
        hipFree(dx); if (dn) hipFree(dn);
        if (e != hipSuccess) return fail(h, M3DREG_ERR_HIP, "cloud_export", e);
        if (e2 != hipSuccess) return fail(h, M3DREG_ERR_HIP, "cloud_export", e2);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return M3DREG_OK;
}

int m3dreg_debug_nn(m3dreg_handle* h, const m3dreg_cloud* target, int level, const float* queries_xyz, size_t nq, float max_corr_dist,
                    int32_t* out_idx, float* out_d2) {
    if (!h || !target || !queries_xyz || !out_idx || !out_d2 || level < 0 || level >= target->n_levels || nq == 0 || nq >= 0x7FFFFFFFull)
        return fail(h, M3DREG_ERR_INVALID_ARG, "debug_nn: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    float* dq = nullptr; int32_t* di = nullptr; float* dd = nullptr;
    HIPCHK(h, hipMalloc((void**)&dq, 12 * nq));
    hipError_t e = hipMalloc((void**)&di, 4 * nq);
    if (e == hipSuccess) e = hipMalloc((void**)&dd, 4 * nq);
    if (e == hipSuccess) e = hipMemcpyAsync(dq, queries_xyz, 12 * nq, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = m3d_launch_debug_nn(h->stream, level_dev(target->lv[level], target->nrm_in), dq, int(nq), max_corr_dist * max_corr_dist, di, dd);
    if (e == hipSuccess) e = hipMemcpyAsync(out_idx, di, 4 * nq, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_d2, dd, 4 * nq, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    hipFree(dq); if (di) hipFree(di); if (dd) hipFree(dd);
    if (e != hipSuccess) return fail(h, M3DREG_ERR_HIP, "debug_nn", e);
    if (e2 != hipSuccess) return fail(h, M3DREG_ERR_HIP, "debug_nn", e2);
    return M3DREG_OK;
}

int m3dreg_debug_accumulate(m3dreg_handle* h, const m3dreg_cloud* source, const m3dreg_cloud* target, int level, const float T[16],
                            int64_t sums[M3DREG_NSUMS], int32_t exps[6]) {
    if (!h || !source || !target || !T || !sums || !exps || level < 0 || level >= h->params.n_levels)
        return fail(h, M3DREG_ERR_INVALID_ARG, "debug_accumulate: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_batch(h, 1);
    if (rc) return rc;
    m3dreg_pair p;
    p.source = source; p.target = target;
    memcpy(p.init_T, T, sizeof(float) * 16);
    int max_n_src = 0;
    if ((rc = build_jobs(h, &p, 1, max_n_src))) return rc;
    if ((rc = ensure_match(h, 1, max_n_src))) return rc;
    const M3dJob* hj = &h->h_jobs[size_t(level) * h->cap_pairs];
    HIPCHK(h, hipMemcpyAsync(h->d_jobs, hj, sizeof(M3dJob), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_states, h->h_states, sizeof(M3dPairState), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, m3d_launch_accumulate_only(h->stream, h->d_jobs, 1, max_n_src, h->params.metric, h->icp_variant, h->d_match, h->match_stride));
    HIPCHK(h, hipMemcpyAsync(h->h_states, h->d_states, sizeof(M3dPairState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const long long* raw = h->h_states[0].sums;
    if (h->params.metric == M3DREG_POINT_TO_PLANE) for (int i = 0; i < M3DREG_NSUMS; i++) sums[i] = raw[i];
    else {
        // same integer identities as expand_pt2pt() in icp.hip
        auto slot = [](int k, int l) { return k * 6 - (k * (k - 1)) / 2 + (l - k); };
        for (int i = 0; i < M3DREG_NSUMS; i++) sums[i] = 0;
        sums[slot(0, 0)] = raw[0]; sums[slot(0, 1)] = raw[1]; sums[slot(0, 2)] = raw[2];
        sums[slot(1, 1)] = raw[3]; sums[slot(1, 2)] = raw[4]; sums[slot(2, 2)] = raw[5];
        sums[slot(0, 4)] = -raw[8]; sums[slot(0, 5)] = raw[7]; sums[slot(1, 3)] = raw[8]; sums[slot(1, 5)] = -raw[6];
        sums[slot(2, 3)] = -raw[7]; sums[slot(2, 4)] = raw[6];
        sums[slot(3, 3)] = sums[slot(4, 4)] = sums[slot(5, 5)] = raw[16] * (1ll << 30);
        for (int k = 0; k < 6; k++) sums[21 + k] = raw[9 + k];
        sums[27] = raw[15]; sums[28] = raw[16];
    }
    for (int i = 0; i < 6; i++) exps[i] = hj->exps[i];
    return M3DREG_OK;
}

int m3dreg_debug_counters(m3dreg_handle* h, uint64_t out[2]) {
    if (!h || !out || !h->h_states) return M3DREG_ERR_INVALID_ARG;
    out[0] = h->h_states[0].ctr[0]; out[1] = h->h_states[0].ctr[1];
    return M3DREG_OK;
}

int m3dreg_debug_trace(m3dreg_handle* h, double* poses, size_t cap, size_t* n_out) {
    if (!h || !n_out) return M3DREG_ERR_INVALID_ARG;
    size_t k = h->last_trace_n < cap ? h->last_trace_n : cap;
    if (poses && k) memcpy(poses, h->h_trace, sizeof(double) * 16 * k);
    *n_out = h->last_trace_n;
    return M3DREG_OK;
}

}  // extern "C"
