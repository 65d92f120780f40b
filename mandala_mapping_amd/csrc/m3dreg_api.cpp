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

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct DevLevel {
    M3dGrid grid{};
    int32_t bits[3]{};
    float mx[3]{};
    float lbound = 0.f;
    float4* pts = nullptr;
    float* src3 = nullptr;         // finest level only: the sorted points packed {x, y, z} (the source stream of a registration)
    M3dBucket* htab = nullptr;
    uint32_t hcap = 0;             // allocated entries (worst case); the used size is grid.hmask + 1
    uint32_t* bigcum = nullptr;
    float4* cbox = nullptr;        // chunk boxes of the sorted points
    uint32_t* order = nullptr;     // the cloud's 256-point blocks, most crowded first
    M3dTileHdr* thdr = nullptr;    // tile headers / images of the LDS-staged search (null for a source-only cloud)
    uint8_t* timg = nullptr;
    M3dTileImgMeta* timeta = nullptr;
    uint32_t* occ = nullptr;       // occupancy bitmap of the bucket positions
    uint32_t bigcap = 0;
    uint32_t* keys = nullptr;
    uint32_t* skey = nullptr;
    uint32_t* perm = nullptr;
    float4* nrm = nullptr;         // normals in this level's sorted order (point-to-plane)
    uint32_t n_cells_host = 0;     // valid once the cloud's meta data was fetched (fetch_meta)
    float sumsq_host = 0.f;        // (same) sum of squared voxel populations
    uint32_t* dyn = nullptr;       // the level's M3dLevelMeta (144 B) in the cloud's block: grid geometry, table geometry, counts, error state —
                                   // all derived on the device; grid / bits / mx / lbound above are host COPIES, valid after fetch_meta
};

struct Block { void* p = nullptr; size_t bytes = 0; };

// bump allocator over one device block (256-byte aligned pieces); with base == nullptr it only sizes
struct Carver {
    uint8_t* base; size_t off = 0;
    explicit Carver(void* b) : base(static_cast<uint8_t*>(b)) {}
    template <typename T> T* take(size_t count) {
        off = (off + 255) & ~size_t(255);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += sizeof(T) * (count ? count : 1);
        return r;
    }
};

}  // namespace

struct BatchReady;
struct m3dreg_cloud {
    int32_t n = 0;
    int32_t n_valid = 0;
    int32_t n_levels = 0;
    float leaf[M3DREG_MAX_LEVELS]{};
    bool has_normals = false;
    bool source_only = false;      // m3dreg_cloud_desc.source_only: sorted, but no hash table / chunk boxes / normals — never a target
    bool has_tiles = false;        // the levels' tile images were built (k_tiles_normals): the LDS-staged search can use this cloud as a target
    float4* xyz = nullptr;         // coordinates in input order
    float mn[3]{}, mx[3]{};
    DevLevel lv[M3DREG_MAX_LEVELS];
    Block block;                   // ONE device allocation holds every array of the cloud
    m3dreg_handle* owner = nullptr;    // the handle whose stream bucketed it
    struct BatchReady* ready = nullptr;   // event recorded behind the bucketing of the batch this cloud came from (shared, ref-counted)
    hipEvent_t last_use = nullptr;     // recorded on ANOTHER handle's stream behind its last use of this cloud: the owner's stream waits for it before the block is re-used
    bool meta_ready = false;           // geometry / counts / error state read back to the host (lazily: grid_info, export, debug_nn, the synchronous entry points)
    int err = 0;                       // valid once meta_ready: 0 or the m3dreg_error the device found (no finite point, grid too large)
};
struct BatchReady { hipEvent_t ev = nullptr; int refs = 0; };

struct m3dreg_handle {
    int live_clouds = 0;               // clouds this handle owns: it is only released once they are gone (a cloud's block returns to ITS pool)
    bool closed = false;               // m3dreg_destroy was called
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    m3dreg_params params{};
    std::string err;
    // pooled cloud blocks (hipMalloc/hipFree per cloud would serialise the device) and the batch workspace
    std::vector<Block> pool;
    size_t pool_bytes = 0;
    Block ws;                          // device workspace of the bucketing batch
    void* h_ws = nullptr;              // pinned host staging (descriptors of a bucketing batch)
    std::vector<std::vector<float>> repack;   // host-side repacked payloads of the batch in flight (unaligned layouts): alive until `staged` has fired
    hipEvent_t staged = nullptr;       // recorded behind the copies out of h_ws: the next batch waits for it before it overwrites the staging area
    size_t h_ws_bytes = 0;
    // batch state
    size_t cap_pairs = 0;
    M3dJob* d_jobs = nullptr;          // [levels][cap_pairs]
    M3dPairState* d_states = nullptr;  // [cap_pairs]
    float* d_ring = nullptr;           // [cap_pairs][32][12] pose rings of the pairs (the NN certificates' "where was the query at its last search")
    double* d_trace = nullptr;         // [M3D_MAX_TRACE][16], first pair only
    M3dJob* h_jobs = nullptr;          // pinned
    M3dPairState* h_states = nullptr;  // pinned
    double* h_trace = nullptr;         // pinned
    size_t pending_pairs = 0;
    // The dense-level schedule of the batch being enqueued, per level (build_jobs; ABI 8: a function of the batch alone — rounds 4-5 remembered the handle's LAST batch):
    // 0 = no pair's target level is dense (k_nn_coop is not launched), 2 = every pair's is (k_nn_coop is the only search launch), 1 = some are — or the host
    // cannot tell: a cloud of the batch came out of the enqueue-only bucketing and nobody has read its counts back yet (then both kernels are launched and the
    // device-side flags decide, pair by pair: an empty launch costs ~5 us, never a bit).
    uint8_t dense_level[M3DREG_MAX_LEVELS] = { 1, 1, 1, 1 };
    size_t last_trace_n = 0;
    // gpu_6dslam_node surface
    m3dreg_cloud* target = nullptr;
    // sync-free early termination: k_solve_update publishes {sequence, active pairs} into host-mapped memory
    volatile unsigned long long* h_progress = nullptr;   // pinned + mapped
    unsigned long long* d_progress = nullptr;            // device view of the same word
    unsigned int seq = 0;
    uint64_t launched_iters = 0, skipped_iters = 0;
    int certify = 1;
    int xcd_rot = 0;                   // this handle's rotation of the block -> XCD map (handles created one after the other get 0, 3, 6, 1, ...)
    int lane_min = 32;                 // blocks with >= this many queries to search: one query per lane (throughput) instead of 8 lanes per query (latency) — 32 = one cooperative pass at most (two passes of 15 us each were the tail of iterations 4-9)
    float seed_reach = 0.99f;          // seeds farther than this many voxel edges are not used (any value in (0, 0.99] gives identical results)
    int tiles = 1;                     // 1 = dense search blocks go through the LDS-staged target tiles (k_nn_tiles); 0 = every search walks global memory (M3DREG_TILES, A/B)
    int lean = 1;                      // tile iterations run k_nn_iter<true> (classify + bin only, 41 VGPRs; what it cannot bin is walked by the reduction pass) when every target of the batch has tiles and the registration has one level (M3DREG_LEAN)
    bool batch_all_tiles = false;
    int fuse_from = 8;                 // from this iteration of a level on (and never before tile_iters) search and reduction are ONE launch, k_icp_late (M3DREG_FUSE_FROM, 0 = never)
    int tile_iters = 8;                // ... during the first tile_iters iterations of a level (M3DREG_TILE_ITERS): later the few searches left are walked by k_nn_iter itself
    int* d_match = nullptr;            // [pairs * match_stride] x {match int32 | pad | cache int64 | certificate state float4} (variant 2)
    long long* d_partials = nullptr;   // block partial sums of the reduction pass
    unsigned int* d_tickets = nullptr; // arrival counters of the reduction pass
    size_t tickets_cap = 0;
    size_t partials_cap = 0;
    size_t match_cap = 0;
    size_t match_pairs_cap = 0;
    int match_stride = 0;
    size_t match_pairs = 0;
    float4* d_rec = nullptr;           // query records of the LDS-staged search: float4[rec_cap] then float[rec_cap] (seed distances)
    size_t rec_cap = 0, rec_stride = 0;
    unsigned int* d_tcnt = nullptr;    // records per tile
    size_t tcnt_cap = 0;
    uint2* d_witems = nullptr;         // work items of k_nn_tiles: M3D_TILE_LISTS counters (128 B apart), then the lists (witems_cap items each)
    size_t witems_cap = 0;
    int ntile_max = 0, cnt_stride = 0;
    // measurement: event pairs around the dominant kernel
    bool profiling = false;
    int prof_every = 1;                // every n-th iteration is bracketed by events (m3dreg_profile_enable(h, n))
    int prof_batch_every = 1;          // every n-th bucketing batch / registration batch carries its per-batch brackets (m3dreg_profile_batches)
    uint64_t prof_bucketings = 0, prof_batches = 0;   // batches seen while profiling (the sampling counters)
    bool chain_bracketed = false;      // this batch's chain bracket was opened (batch_begin): batch_end closes it
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_kind;          // per recorded event: 0 = before the dominant kernel (= start of an iteration), 1 = after it, 2 = end of the batch
    size_t ev_used = 0;
    uint64_t prof_launches[5] = { 0, 0, 0, 0, 0 };   // M3DREG_PROFILE_*: iteration, correspondence step, bucketing batch, reduce + solve, all iterations of a batch (launches = iterations enqueued)
    double prof_ms[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
    uint64_t chain_iters = 0;          // iterations enqueued between the open kind-5 event and its kind-6 partner
    // the batch being enqueued (batch_begin / batch_step / batch_end)
    struct Run { size_t n_pairs = 0; int max_n_src = 0; int l = 0, it = 0; bool can_stop_early = false, prev_sampled = false, throttle_off = false; uint64_t iters_before = 0; unsigned int level_first_seq = 0; } run;
    hipEvent_t done_ev = nullptr;      // recorded behind a batch's last operation (m3dreg_batch_wait waits for it, not for the whole stream)
    bool done_recorded = false;
    bool throttle = false;             // inside the synchronous m3dreg_align_batch: the enqueue of a convergence-terminated batch stays a few iterations ahead of the device, not a level
    int alone = 0;                     // m3dreg_set_latency_mode: this handle's batches have the GPU to themselves (grids sized for latency)
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

static_assert(M3D_ERR_GRID_TOO_LARGE == M3DREG_ERR_GRID_TOO_LARGE && M3D_ERR_EMPTY_CLOUD == M3DREG_ERR_EMPTY_CLOUD, "device error codes");

// No exception crosses the C boundary (include/m3dreg.h): every exported function runs its body inside this guard. The library
// itself throws nothing; what can throw underneath it is the host allocator (new, std::vector, std::string).
template <class F> int m3d_guarded(m3dreg_handle* h, const char* where, F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { try { return fail(h, M3DREG_ERR_OUT_OF_MEMORY, where); } catch (...) { return M3DREG_ERR_OUT_OF_MEMORY; } }
    catch (...) { try { return fail(h, M3DREG_ERR_HIP, where); } catch (...) { return M3DREG_ERR_HIP; } }
}
// test hook (m3dreg_debug_fail_alloc): the n-th host allocation from now on throws std::bad_alloc
std::atomic<int> g_fail_alloc{0};
inline void alloc_point() { int v = g_fail_alloc.load(); if (v > 0 && g_fail_alloc.fetch_sub(1) == 1) throw std::bad_alloc(); }


// A handle that was destroyed while clouds of it were still alive lives on until the last of them is released — but a stream the CALLER lent it
// (m3dreg_create's `stream`) may be gone by then: after m3dreg_destroy nothing touches a borrowed stream any more, the device is synchronised instead
// (ADVICE r2).
inline bool stream_usable(const m3dreg_handle* h) { return h->own_stream || !h->closed; }
inline hipError_t sync_handle(m3dreg_handle* h) { return stream_usable(h) ? hipStreamSynchronize(h->stream) : hipDeviceSynchronize(); }

// ---- M3DREG_POISON (diagnosis only): every device allocation of the library — and every pooled block handed out again — is filled with a hostile
// pattern before use: "rand" = pseudo-random words (a hash of the word's index), or a hex word ("ffffffff", "7fc00000", "0"). A kernel that reads memory the
// library never wrote then meets the same garbage in every run: the parity tests fail or the process faults deterministically, instead of one run in hundreds
// depending on what the allocation held before. Unset (the default): no fill, no cost.
struct PoisonMode { bool on = false, rnd = false; uint32_t word = 0; };
const PoisonMode& poison_mode() {
    static const PoisonMode m = [] {
        PoisonMode q;
        const char* v = getenv("M3DREG_POISON");
        if (v && *v) { q.on = true; if (!strcmp(v, "rand")) q.rnd = true; else q.word = uint32_t(strtoul(v, nullptr, 16)); }
        return q;
    }();
    return m;
}
hipError_t poison(void* p, size_t bytes, hipStream_t s) {   // (s == nullptr: synchronous)
    const PoisonMode& m = poison_mode();
    if (!m.on || !p || bytes < 4) return hipSuccess;
    hipError_t e = m.rnd ? m3d_launch_poison(s, p, bytes, 0x6D3Du) : hipMemsetD32Async(static_cast<hipDeviceptr_t>(p), int(m.word), bytes / 4, s);
    if (e == hipSuccess && !s) e = hipDeviceSynchronize();
    return e;
}
hipError_t m3d_malloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) e = poison(*p, bytes, nullptr);
    return e;
}

// one polite busy-wait step of the synchronous call's throttle (x86: pause; aarch64: yield; anything else: the scheduler's yield)
inline void m3d_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

const size_t POOL_CAP_BYTES = size_t(16) << 30;   // cached, unused cloud blocks kept for reuse

int pool_get(m3dreg_handle* h, size_t bytes, Block& out) {
    int best = -1;
    for (size_t i = 0; i < h->pool.size(); i++)
        if (h->pool[i].bytes >= bytes && h->pool[i].bytes <= bytes + bytes / 2 && (best < 0 || h->pool[i].bytes < h->pool[size_t(best)].bytes)) best = int(i);
    if (best >= 0) {
        out = h->pool[size_t(best)];
        h->pool_bytes -= out.bytes;
        h->pool.erase(h->pool.begin() + best);
        if (poison_mode().on) HIPCHK(h, poison(out.p, out.bytes, h->stream));   // (diagnosis: a recycled block holds the previous cloud)
        return M3DREG_OK;
    }
    void* p = nullptr;
    hipError_t e = m3d_malloc(&p, bytes);
    if (e != hipSuccess) {   // make room: drop the cache and retry once
        hipStreamSynchronize(h->stream);
        for (Block& b : h->pool) hipFree(b.p);
        h->pool.clear(); h->pool_bytes = 0;
        e = m3d_malloc(&p, bytes);
        if (e != hipSuccess) return fail(h, M3DREG_ERR_HIP, "m3d_malloc(cloud block)", e);
    }
    out.p = p; out.bytes = bytes;
    return M3DREG_OK;
}

// Blocks go back to the cache without a device sync: every use of a block is ordered on the handle's stream.
void pool_put(m3dreg_handle* h, Block b) {
    if (!b.p) return;
    if (h->pool_bytes + b.bytes > POOL_CAP_BYTES) { sync_handle(h); hipFree(b.p); return; }
    h->pool.push_back(b);
    h->pool_bytes += b.bytes;
}

int ensure_ws(m3dreg_handle* h, size_t dev_bytes, size_t host_bytes) {
    if (dev_bytes > h->ws.bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->ws.p) hipFree(h->ws.p);
        h->ws = Block();
        const size_t cap = dev_bytes + dev_bytes / 4;
        HIPCHK(h, m3d_malloc(&h->ws.p, cap));
        h->ws.bytes = cap;
    }
    if (host_bytes > h->h_ws_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->h_ws) hipHostFree(h->h_ws);
        h->h_ws = nullptr; h->h_ws_bytes = 0;
        const size_t cap = host_bytes + host_bytes / 4;
        HIPCHK(h, hipHostMalloc(&h->h_ws, cap, hipHostMallocDefault));
        h->h_ws_bytes = cap;
    }
    return M3DREG_OK;
}

void release_handle(m3dreg_handle* h);
// The block goes back to the OWNER's pool (whichever handle asked): every later use of it is enqueued on the owner's stream, which
// first waits for the last use another handle's stream made of the cloud.
void free_cloud(m3dreg_handle* h, m3dreg_cloud* c) {
    if (!c) return;
    m3dreg_handle* o = c->owner ? c->owner : h;
    if (c->ready && --c->ready->refs == 0) { hipEventDestroy(c->ready->ev); delete c->ready; }
    if (c->last_use) {
        if (o) { hipSetDevice(o->device); if (stream_usable(o)) hipStreamWaitEvent(o->stream, c->last_use, 0); else hipEventSynchronize(c->last_use); }
        hipEventDestroy(c->last_use);
    }
    if (o) pool_put(o, c->block); else if (c->block.p) hipFree(c->block.p);
    const bool counted = c->owner != nullptr;
    delete c;
    if (counted && --o->live_clouds == 0 && o->closed) release_handle(o);
}

// a handle other than the owner has just enqueued work that reads the cloud: remember where that work ends
int note_foreign_use(m3dreg_handle* h, const m3dreg_cloud* cc) {
    m3dreg_cloud* c = const_cast<m3dreg_cloud*>(cc);
    if (!c->owner || c->owner == h) return M3DREG_OK;
    if (!c->last_use) HIPCHK(h, hipEventCreateWithFlags(&c->last_use, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(c->last_use, h->stream));
    return M3DREG_OK;
}

uint32_t table_cap(size_t n) { uint32_t hs = 16; while (hs < 2u * uint32_t(n)) hs <<= 1; return hs; }

// lay a cloud's arrays out in its block (base == nullptr: size only)
size_t carve_cloud(m3dreg_cloud* c, void* base, const m3dreg_params& P) {
    Carver k(base);
    const size_t n = size_t(c->n);
    c->xyz = k.take<float4>(n);
    for (int l = 0; l < P.n_levels; l++) {
        DevLevel& L = c->lv[l];
        if (c->source_only && l < P.n_levels - 1) {   // the coarser levels of a source-only cloud are never built (create_clouds: B.n = 0): only their meta exists
            L = DevLevel();
            L.dyn = k.take<uint32_t>(sizeof(M3dLevelMeta) / 4);
            continue;
        }
        L.hcap = table_cap(n);
        L.bigcap = uint32_t(n / 65536 + 1);
        L.pts = k.take<float4>(n);
        L.src3 = (l == P.n_levels - 1) ? k.take<float>(3 * n) : nullptr;
        if (!c->source_only) {   // (nobody searches a source-only cloud: no bucket table, no chunk boxes)
            L.htab = k.take<M3dBucket>(L.hcap);
            L.bigcum = k.take<uint32_t>(size_t(L.bigcap) * 8);
            L.cbox = k.take<float4>(2 * ((n + M3D_CHUNK - 1) / M3D_CHUNK));
        } else { L.htab = nullptr; L.bigcum = nullptr; L.cbox = nullptr; L.hcap = 0; L.bigcap = 0; }
        L.order = k.take<uint32_t>((n + 255) / 256);
        // tiles only where they are used: a target's FINEST level. A pyramid's coarser levels hold too many points per bucket for an image (their tiles
        // ended up flagged, their searches in the global walk anyway) — not carving them saves 1.5 images of 56 KB per 512 points and a 1 MiB bitmap per
        // level (a 2 M-point map: 330 MB per level) and two thirds of the tile build's work on a pyramid.
        if (!c->source_only && l == P.n_levels - 1) {
            const size_t nt = size_t(m3d_tiles_of(int(n))), ni = nt + size_t(m3d_tile_pool(int(nt)));
            L.thdr = k.take<M3dTileHdr>(nt);
            L.timg = k.take<uint8_t>(ni * M3D_TILE_IMG_BYTES);
            L.timeta = k.take<M3dTileImgMeta>(ni);
            L.occ = k.take<uint32_t>(size_t(1) << (M3D_OCC_BITS - 5));
        } else { L.thdr = nullptr; L.timg = nullptr; L.timeta = nullptr; L.occ = nullptr; }
        L.keys = k.take<uint32_t>(n); L.skey = k.take<uint32_t>(n); L.perm = k.take<uint32_t>(n);
        L.nrm = (P.metric == M3DREG_POINT_TO_PLANE && !c->source_only) ? k.take<float4>(n) : nullptr;
        L.dyn = k.take<uint32_t>(sizeof(M3dLevelMeta) / 4);
    }
    return (k.off + 255) & ~size_t(255);
}

hipEvent_t next_event(m3dreg_handle* h);
void roctx_push(const char* name);
void roctx_pop();

struct CloudInput { const void* data; size_t n, step, ox, oy, oz; bool is_device; bool aligned; bool src_only = false;
                    bool generic = false; size_t width = 0, row_step = 0, data_bytes = 0; bool f64[3] = { false, false, false }; bool bigendian = false; };

// Bucket a batch of clouds: one decode launch, one bucketing pipeline for every grid of every cloud, NO host synchronisation:
// the grid geometry (from the exact AABBs), the table geometry and the error state of every cloud are derived on the device
// and stay there (M3dLevelMeta in the cloud's block); fetch_meta reads them back when somebody asks.
int create_clouds(m3dreg_handle* h, const CloudInput* in, size_t k, m3dreg_cloud** out) {
    const m3dreg_params& P = h->params;
    const bool want_normals = P.metric == M3DREG_POINT_TO_PLANE;
    const int grids_per_cloud = P.n_levels + (want_normals ? 1 : 0);
    const size_t n_builds = k * size_t(grids_per_cloud);
    std::vector<m3dreg_cloud*> cl(k, nullptr);
    auto cleanup = [&]() { hipStreamSynchronize(h->stream); for (m3dreg_cloud*& c : cl) { free_cloud(h, c); c = nullptr; } };
    // a throw anywhere below (a host allocation: std::vector, new — also AFTER the pipeline was enqueued) gives every block back
    // before it leaves: the clouds are not the caller's yet (they leaked, with their HBM, when alloc_point() / new threw late)
    struct Guard { decltype(cleanup)& f; bool armed = true; ~Guard() { if (armed) f(); } } guard{ cleanup };
    size_t max_n = 0;
    for (size_t i = 0; i < k; i++) {
        alloc_point();
        m3dreg_cloud* c = new m3dreg_cloud();
        cl[i] = c;
        c->n = int32_t(in[i].n);
        c->n_levels = P.n_levels;
        c->source_only = in[i].src_only;
        for (int l = 0; l < P.n_levels; l++) c->leaf[l] = P.leaf[l];
        const size_t bytes = carve_cloud(c, nullptr, P);
        int rc = pool_get(h, bytes, c->block);
        if (rc) { cleanup(); return rc; }
        carve_cloud(c, c->block.p, P);
        if (in[i].n > max_n) max_n = in[i].n;
    }
    // ---- workspace layout (device) + pinned host staging -------------------------------------------------
    std::vector<uint8_t*> staged(k, nullptr);
    std::vector<uint32_t*> aabb(k, nullptr);
    struct BuildWs { uint32_t *ka = nullptr, *va = nullptr, *kb = nullptr, *vb = nullptr, *hist = nullptr, *dyn = nullptr, *blkw = nullptr;
                     // a normal grid (a point-to-plane cloud's first build): its voxel table (round 5: never sorted, see bucket.hip)
                     uint32_t *nkeys = nullptr, *nlist = nullptr, *nvcnt = nullptr; long long* mom = nullptr; float4* nnrm = nullptr; uint32_t ncap = 0; };
    std::vector<BuildWs> bw(n_builds);
    M3dDecode* d_dec = nullptr; M3dBuild* d_builds = nullptr; uint8_t* zero_lo = nullptr; uint8_t* zero_hi = nullptr;
    auto layout = [&](void* base) -> size_t {
        Carver w(base);
        d_dec = w.take<M3dDecode>(k);
        d_builds = w.take<M3dBuild>(n_builds);
        // --- region zeroed with one memset: aabb, dyn (the moments are zeroed where they are used: k_finalize_level, per voxel head) ---
        w.take<uint8_t>(0); zero_lo = base ? static_cast<uint8_t*>(base) + ((w.off + 255) & ~size_t(255)) : nullptr;
        uint32_t* aabb_all = w.take<uint32_t>(8 * k);              // contiguous: read back with one copy
        for (size_t i = 0; i < k; i++) aabb[i] = base ? aabb_all + 8 * i : nullptr;
        const size_t mw = sizeof(M3dLevelMeta) / 4;
        uint32_t* dyn_all = w.take<uint32_t>(mw * n_builds);       // one M3dLevelMeta per build (the normal grids' live here, the levels' in their clouds)
        for (size_t b = 0; b < n_builds; b++) bw[b].dyn = base ? dyn_all + mw * b : nullptr;
        w.take<uint8_t>(0); zero_hi = base ? static_cast<uint8_t*>(base) + ((w.off + 255) & ~size_t(255)) : nullptr;
        // --- the rest ---
        for (size_t i = 0; i < k; i++) {
            const size_t n = in[i].n;
            staged[i] = (!in[i].is_device) ? w.take<uint8_t>(in[i].generic ? in[i].data_bytes : (in[i].aligned ? n * in[i].step : 12 * n)) : nullptr;
            for (int gidx = 0; gidx < grids_per_cloud; gidx++) {
                BuildWs& B = bw[i * size_t(grids_per_cloud) + size_t(gidx)];
                if (want_normals && gidx == 0) {   // the normal grid: no sort workspace — a table of its occupied voxels (keys 4 B + moments 80 B + normal 16 B per slot; only taken
                    // slots are ever touched beyond the keys) and the list of the slots every 256-position block of the finest level takes
                    if (!in[i].src_only) {
                        { uint32_t c = 16; while (size_t(c) < n + n / 4 + 16) c <<= 1; B.ncap = c; }   // a power of two > 1.25 n: at most n voxels, so there is always an empty slot to end a probe chain (an ordinary cloud fills a few per cent)
                        B.nkeys = w.take<uint32_t>(2 * size_t(B.ncap));   // keys, then the "further runs of this voxel" counters
                        B.mom = w.take<long long>(10 * size_t(B.ncap)); B.nnrm = w.take<float4>(B.ncap);
                        B.nlist = w.take<uint32_t>(256 * ((n + 255) / 256)); B.nvcnt = w.take<uint32_t>((((n + 255) / 256) + 15) & ~size_t(15));   // (read sixteen at a time)
                    }
                    continue;
                }
                B.ka = w.take<uint32_t>(n); B.va = w.take<uint32_t>(n); B.kb = w.take<uint32_t>(n); B.vb = w.take<uint32_t>(n);
                B.hist = w.take<uint32_t>(256 * size_t(m3d_sort_tiles(int(n)) + 1));
                B.blkw = w.take<uint32_t>((n + 255) / 256);
            }
        }
        return (w.off + 255) & ~size_t(255);
    };
    const size_t ws_bytes = layout(nullptr);
    const size_t host_bytes = sizeof(M3dDecode) * k + sizeof(M3dBuild) * n_builds + 32 * (k + n_builds) + 1024;
    int rc = ensure_ws(h, ws_bytes, host_bytes);
    if (rc) { cleanup(); return rc; }
    layout(h->ws.p);
    if (poison_mode().on) { hipError_t e = poison(h->ws.p, h->ws.bytes, h->stream); if (e != hipSuccess) { cleanup(); return fail(h, M3DREG_ERR_HIP, "poison(workspace)", e); } }   // (diagnosis: the workspace holds the previous batch)
    if (h->staged) { hipError_t e = hipEventSynchronize(h->staged); if (e != hipSuccess) { cleanup(); return fail(h, M3DREG_ERR_HIP, "hipEventSynchronize(staging)", e); } }
    Carver hw(h->h_ws);   // the same sequence of takes as the device layout above: the staging block mirrors it byte for byte
    M3dDecode* h_dec = hw.take<M3dDecode>(k);
    M3dBuild* h_builds = hw.take<M3dBuild>(n_builds);
    hw.take<uint8_t>(0);
    uint32_t* h_aabb0 = hw.take<uint32_t>(8 * k);   // zeros for the AABB accumulators: they travel with the descriptors (no memset)
    memset(h_aabb0, 0, sizeof(uint32_t) * 8 * k);
#define B_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(h, M3DREG_ERR_HIP, #expr, _e); } } while (0)
    (void)zero_lo; (void)zero_hi;   // (the metas in this region are written in full by k_grid_params / k_table_params before anything reads them)
    // ---- a2: stage + decode ----------------------------------------------------------------------------------
    std::vector<std::vector<float>>& repack = h->repack;   // the previous batch's copies are done (h->staged was waited for above)
    repack.assign(k, std::vector<float>());
    for (size_t i = 0; i < k; i++) {
        M3dDecode& D = h_dec[i];
        size_t step = in[i].step, ox = in[i].ox, oy = in[i].oy, oz = in[i].oz;
        const uint8_t* raw = static_cast<const uint8_t*>(in[i].data);
        if (!in[i].is_device) {
            const void* src = in[i].data;
            size_t bytes = in[i].generic ? in[i].data_bytes : in[i].n * in[i].step;
            if (!in[i].generic && !in[i].aligned) {   // never produced by m3d_aggregator (16/0/4/8); repacked on the host
                repack[i].resize(3 * in[i].n);
                const uint8_t* b = static_cast<const uint8_t*>(in[i].data);
                for (size_t j = 0; j < in[i].n; j++) {
                    memcpy(&repack[i][3 * j], b + j * step + ox, 4);
                    memcpy(&repack[i][3 * j + 1], b + j * step + oy, 4);
                    memcpy(&repack[i][3 * j + 2], b + j * step + oz, 4);
                }
                src = repack[i].data(); bytes = 12 * in[i].n; step = 12; ox = 0; oy = 4; oz = 8;
            }
            B_HIP(hipMemcpyAsync(staged[i], src, bytes, hipMemcpyHostToDevice, h->stream));
            raw = staged[i];
        }
        D.raw = raw; D.n = int(in[i].n); D.step = int(step); D.ox = int(ox); D.oy = int(oy); D.oz = int(oz);
        D.generic = in[i].generic ? 1 : 0; D.width = int(in[i].generic ? in[i].width : in[i].n); D.row_step = int(in[i].row_step);
        for (int a = 0; a < 3; a++) D.f64[a] = in[i].f64[a] ? 1 : 0;
        D.bigendian = in[i].bigendian ? 1 : 0;
        D.xyz = cl[i]->xyz; D.aabb = aabb[i];
    }
    // ---- a3/a4/a9: one build descriptor per grid; only sizes, pointers and the leaf come from the host -------------------------
    int any_tiles = 0;   // 0 none, 1 only clouds' last builds (finest levels: the default), 2 some other build too (M3DREG_TILES_ALL_LEVELS)
    bool any_fine = false;
    for (size_t i = 0; i < k; i++) {
        m3dreg_cloud* c = cl[i];
        const bool no_normals = in[i].src_only;   // a source-only cloud: sorted, no normal grid, no normals
        c->has_normals = want_normals && !no_normals;
        c->source_only = no_normals;
        for (int gidx = 0; gidx < grids_per_cloud; gidx++) {
            const size_t bi = i * size_t(grids_per_cloud) + size_t(gidx);
            BuildWs& W = bw[bi];
            const bool is_ng = want_normals && gidx == 0;
            M3dBuild& B = h_builds[bi];
            memset(&B, 0, sizeof(B));
            B.fine = -1; B.nrm_build = -1;
            B.xyz = c->xyz; B.aabb = aabb[i];
            if (is_ng) {   // geometry + error state from k_grid_params (it is the cloud's first build: its aabb pointer is the one that kernel reads), nothing sorted
                B.n = 0; B.ntiles = 0;
                B.grid.leaf = P.normal_leaf;
                B.dyn = W.dyn;
                if (!no_normals) {
                    B.nkeys = W.nkeys; B.mom = W.mom; B.nnrm = W.nnrm; B.nlist = W.nlist; B.nvcnt = W.nvcnt; B.ncap = W.ncap;
                    int lg = 0; while ((1u << lg) < W.ncap) lg++;
                    B.nshift = 32 - lg;
                }
                continue;
            }
            const int lvl = gidx - (want_normals ? 1 : 0);
            DevLevel& L = c->lv[lvl];
            B.n = c->n; B.sort_passes = 0; B.ntiles = m3d_sort_tiles(c->n);
            // a coarser level of a pyramid is sorted from its cloud's finest level's order (M3dBuild::fine)
            B.fine = (P.n_levels > 1 && lvl < P.n_levels - 1) ? int(i * size_t(grids_per_cloud)) + (want_normals ? 1 : 0) + (P.n_levels - 1) : -1;
            B.grid.leaf = P.leaf[lvl];
            B.keys = L.keys; B.ka = W.ka; B.va = W.va; B.kb = W.kb; B.vb = W.vb; B.hist = W.hist;
            B.skey_out = L.skey; B.perm_out = L.perm; B.pts = L.pts; B.src3 = L.src3; B.htab = L.htab; B.hcap = L.hcap;
            B.cbox = L.cbox;
            B.blkw = W.blkw; B.order = L.order;
            B.bigcum = L.bigcum; B.bigcap = L.bigcap; B.dyn = L.dyn;   // a level's meta lives in its cloud (read by the jobs later)
            B.nrm_sorted = no_normals ? nullptr : L.nrm;
            if (want_normals && !no_normals) { B.nrm_build = int(i * size_t(grids_per_cloud)); B.nrm_feed = (lvl == P.n_levels - 1) ? 1 : 0; }
            if (no_normals && lvl < P.n_levels - 1) { B.n = 0; B.ntiles = 0; B.fine = -1; }   // the coarser levels of a source-only cloud's pyramid are not built: a registration streams a source in its FINEST level's order on every level (build_jobs)
            if (no_normals) { B.htab = nullptr; B.cbox = nullptr; }         // nor its bucket table and chunk boxes: nobody will search it
            any_fine = any_fine || B.fine >= 0;
            if (!no_normals && h->tiles && L.thdr) { B.thdr = L.thdr; B.timg = L.timg; B.timeta = L.timeta; B.occ = L.occ; any_tiles = std::max(any_tiles, gidx == grids_per_cloud - 1 ? 1 : 2); c->has_tiles = true; }
        }
    }
    // decode and build descriptors sit side by side, laid out alike on both sides of the bus: ONE copy (every copy is a blit kernel
    // on the stream's critical path)
    {
        const size_t span = size_t(reinterpret_cast<uint8_t*>(aabb[0] + 8 * k) - reinterpret_cast<uint8_t*>(d_dec));
        if (size_t(reinterpret_cast<uint8_t*>(h_aabb0 + 8 * k) - reinterpret_cast<uint8_t*>(h_dec)) != span) { cleanup(); return fail(h, M3DREG_ERR_HIP, "descriptor staging layout"); }
        B_HIP(hipMemcpyAsync(d_dec, h_dec, span, hipMemcpyHostToDevice, h->stream));
    }
    roctx_push("m3dreg:bucketing");
    hipEvent_t pb0 = nullptr;
    if (h->profiling && (h->prof_bucketings++ % uint64_t(h->prof_batch_every)) == 0) { pb0 = next_event(h); if (pb0) { h->ev_kind.push_back(3); (void)hipEventRecord(pb0, h->stream); } }
    B_HIP(m3d_launch_decode_aabb(h->stream, d_dec, int(k), int(max_n)));
    if (!h->staged) B_HIP(hipEventCreateWithFlags(&h->staged, hipEventDisableTiming));
    B_HIP(hipEventRecord(h->staged, h->stream));
    B_HIP(m3d_launch_bucket_batch(h->stream, d_builds, int(k), grids_per_cloud, int(max_n), want_normals, any_tiles, P.plane_ratio,
                                  P.normal_min_pts, P.normal_min_spread, any_fine));
    if (pb0) { hipEvent_t e = next_event(h); if (e) { h->ev_kind.push_back(4); (void)hipEventRecord(e, h->stream); } }
    roctx_pop();
    // One event behind the pipeline lets OTHER handles order their streams after it.
    alloc_point();
    BatchReady* br = new BatchReady();
    if (hipEventCreateWithFlags(&br->ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(br->ev, h->stream) != hipSuccess) {
        if (br->ev) hipEventDestroy(br->ev);
        delete br; cleanup(); return fail(h, M3DREG_ERR_HIP, "hipEventRecord(batch ready)");
    }
#undef B_HIP
    for (size_t i = 0; i < k; i++) {
        cl[i]->owner = h; h->live_clouds++; cl[i]->ready = br; br->refs++; cl[i]->meta_ready = false; cl[i]->err = 0;
        out[i] = cl[i];
    }
    guard.armed = false;
    return M3DREG_OK;
}

// geometry, counts and error state of a cloud on the host (waits for its bucketing). Returns the cloud's error, if it has one.
int fetch_meta(m3dreg_handle* h, m3dreg_cloud* c) {
    if (!c->meta_ready) {
        m3dreg_handle* o = c->owner ? c->owner : h;
        HIPCHK(h, hipSetDevice(o->device));
        HIPCHK(h, hipStreamSynchronize(o->stream));
        for (int l = 0; l < c->n_levels; l++) {
            M3dLevelMeta M;
            HIPCHK(h, hipMemcpy(&M, c->lv[l].dyn, sizeof(M), hipMemcpyDeviceToHost));
            DevLevel& L = c->lv[l];
            L.n_cells_host = M.dyn[0];
            L.sumsq_host = M.sumsq;
            L.grid = M.g;
            L.grid.hmask = M.dyn[1];
            L.grid.hshift = int32_t(M.dyn[2]);
            L.lbound = M.lbound;
            for (int a = 0; a < 3; a++) { L.mx[a] = M.mx[a]; L.bits[a] = M.bits[a]; c->mn[a] = M.g.mn[a]; c->mx[a] = M.mx[a]; }
            c->n_valid = M.g.n_valid;
            c->err = M.err;
        }
        c->meta_ready = true;
    }
    if (c->err == M3DREG_ERR_EMPTY_CLOUD) return fail(h, c->err, "cloud has no finite point");
    if (c->err) return fail(h, c->err, "voxel grid needs more than 31 key bits (coarsen leaf or crop the cloud)");
    return M3DREG_OK;
}

// the synchronous entry points: wait for the bucketing and report what the device found, like the old host-side checks did
int finish_sync(m3dreg_handle* h, m3dreg_cloud** cl, size_t k) {
    int rc = M3DREG_OK;
    for (size_t i = 0; i < k && rc == M3DREG_OK; i++) rc = fetch_meta(h, cl[i]);
    if (rc != M3DREG_OK) {
        const std::string msg = h->err;
        for (size_t i = 0; i < k; i++) { free_cloud(h, cl[i]); cl[i] = nullptr; }
        h->err = msg;
    }
    return rc;
}

M3dLevelDev level_dev(const DevLevel& L, bool tiles = false) {
    M3dLevelDev d{};
    d.pts = L.pts; d.nrm = L.nrm; d.htab = L.htab; d.bigcum = L.bigcum; d.cbox = L.cbox; d.dyn = L.dyn; d.g = L.grid;
    d.thdr = tiles ? L.thdr : nullptr; d.timg = tiles ? L.timg : nullptr; d.timeta = tiles ? L.timeta : nullptr; d.occ = tiles ? L.occ : nullptr;
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
    if (h->d_jobs) hipFree(h->d_jobs);       // jobs and states share one block on either side (one copy moves both)
    if (h->h_jobs) hipHostFree(h->h_jobs);
    if (h->d_ring) hipFree(h->d_ring);
    h->d_jobs = nullptr; h->d_states = nullptr; h->h_jobs = nullptr; h->h_states = nullptr; h->d_ring = nullptr; h->cap_pairs = 0;
    size_t cap = n_pairs < 8 ? 8 : n_pairs;
    static_assert(sizeof(M3dJob) % 8 == 0, "the pair states follow the jobs in one block");
    const size_t block = sizeof(M3dJob) * cap * M3DREG_MAX_LEVELS + sizeof(M3dPairState) * cap;
    HIPCHK(h, m3d_malloc((void**)&h->d_jobs, block));
    HIPCHK(h, hipHostMalloc((void**)&h->h_jobs, block, hipHostMallocDefault));
    HIPCHK(h, m3d_malloc((void**)&h->d_ring, sizeof(float) * 32 * 12 * cap));
    h->d_states = reinterpret_cast<M3dPairState*>(h->d_jobs + cap * M3DREG_MAX_LEVELS);
    h->h_states = reinterpret_cast<M3dPairState*>(h->h_jobs + cap * M3DREG_MAX_LEVELS);
    if (!h->d_trace) {
        HIPCHK(h, m3d_malloc((void**)&h->d_trace, sizeof(double) * 16 * M3D_MAX_TRACE));
        HIPCHK(h, hipHostMalloc((void**)&h->h_trace, sizeof(double) * 16 * M3D_MAX_TRACE, hipHostMallocDefault));
    }
    if (!h->h_progress) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                h->h_progress = static_cast<volatile unsigned long long*>(hp);
                h->d_progress = static_cast<unsigned long long*>(dp);
                *h->h_progress = 0ull;
            } else hipHostFree(hp);
        }   // without mapped memory the library simply enqueues every iteration (still correct)
    }
    h->cap_pairs = cap;
    return M3DREG_OK;
}

int ensure_match(m3dreg_handle* h, size_t n_pairs, int max_n_src, int max_n_tgt) {
    // per-pair stride of the per-query arrays: a whole number of 256-query blocks, so that no block of one pair can ever touch
    // slots of the next (an earlier worklist layout raced across pairs with a 64-rounded stride: found with 64 identical
    // pairs giving different results; tests/test_gpu_parity.py::test_identical_pairs_in_one_batch_give_identical_results)
    const size_t stride = (size_t(max_n_src) + 255) & ~size_t(255);
    if (n_pairs * stride > h->match_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_match) hipFree(h->d_match);
        h->d_match = nullptr; h->match_cap = 0;
        const size_t cap = n_pairs * stride + n_pairs * stride / 4;
        HIPCHK(h, m3d_malloc((void**)&h->d_match, sizeof(int) * 4 * cap));   // {match, certificate word} (8 B) | cache (int64)
        h->match_cap = cap;
    }
    // (sized for either setting of m3dreg_set_latency_mode: the mode may change between batches)
    const size_t n_part = n_pairs * size_t(std::max(m3d_acc_blocks(max_n_src, int(n_pairs), 0), m3d_acc_blocks(max_n_src, int(n_pairs), 1))) * M3D_PARTIAL_STRIDE;
    if (n_part > h->partials_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_partials) hipFree(h->d_partials);
        h->d_partials = nullptr; h->partials_cap = 0;
        HIPCHK(h, m3d_malloc((void**)&h->d_partials, sizeof(long long) * (n_part + n_part / 4)));
        h->partials_cap = n_part + n_part / 4;
    }
    const size_t n_tk = size_t(std::max(m3d_ticket_words(int(n_pairs), max_n_src, 0), m3d_ticket_words(int(n_pairs), max_n_src, 1)));
    if (n_tk > h->tickets_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_tickets) hipFree(h->d_tickets);
        h->d_tickets = nullptr; h->tickets_cap = 0;
        HIPCHK(h, m3d_malloc((void**)&h->d_tickets, sizeof(unsigned int) * (n_tk + n_tk / 4)));
        h->tickets_cap = n_tk + n_tk / 4;
        HIPCHK(h, hipMemsetAsync(h->d_tickets, 0, sizeof(unsigned int) * h->tickets_cap, h->stream));   // once: every launch leaves its arrival counters at zero
    }
    h->match_stride = int(stride);
    h->match_pairs = n_pairs;
    if (h->tiles) {   // workspace of the LDS-staged search: per pair, query records per tile + the global-walk list, and their counters
        const size_t ntile = size_t(m3d_tiles_of(max_n_tgt));
        const size_t rec_stride = ntile * M3D_TILE_QCAP;   // the tiles' slabs
        const size_t cnt_stride = (ntile + 1 + 31) & ~size_t(31);
        if (n_pairs * rec_stride > h->rec_cap) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (h->d_rec) hipFree(h->d_rec);
            h->d_rec = nullptr; h->rec_cap = 0;
            const size_t cap = n_pairs * rec_stride + n_pairs * rec_stride / 8;
            HIPCHK(h, m3d_malloc((void**)&h->d_rec, (sizeof(float4) + sizeof(float)) * cap));
            h->rec_cap = cap;
        }
        if (n_pairs * cnt_stride > h->tcnt_cap) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (h->d_tcnt) hipFree(h->d_tcnt);
            h->d_tcnt = nullptr; h->tcnt_cap = 0;
            const size_t cap = n_pairs * cnt_stride + n_pairs * cnt_stride / 4;
            HIPCHK(h, m3d_malloc((void**)&h->d_tcnt, sizeof(unsigned int) * cap));
            h->tcnt_cap = cap;
            HIPCHK(h, hipMemsetAsync(h->d_tcnt, 0, sizeof(unsigned int) * cap, h->stream));   // once: every iteration leaves its counters at zero
        }
        const size_t wcap = n_pairs * (stride / 64 + 2 * ntile + 16);   // work items: at most one per 64 records plus one per tile
        if (wcap > h->witems_cap) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (h->d_witems) hipFree(h->d_witems);
            h->d_witems = nullptr; h->witems_cap = 0;
            // M3D_TILE_LISTS lists, each able to hold every item (which workgroups of k_nn_iter publish is data dependent); the
            // first 128 * M3D_TILE_LISTS bytes hold the lists' counters, one per 128-B line
            HIPCHK(h, m3d_malloc((void**)&h->d_witems, sizeof(uint2) * (wcap + wcap / 4) * M3D_TILE_LISTS + 128 * M3D_TILE_LISTS));
            h->witems_cap = wcap + wcap / 4;
            HIPCHK(h, hipMemsetAsync(h->d_witems, 0, 128 * M3D_TILE_LISTS, h->stream));
        }
        h->ntile_max = int(ntile); h->rec_stride = rec_stride; h->cnt_stride = int(cnt_stride);
    }
    return M3DREG_OK;
}

M3dNnWork nn_work(const m3dreg_handle* h, int level = -1, int it = 0) {
    M3dNnWork w{};
    w.match = reinterpret_cast<int2*>(h->d_match); w.stride = h->match_stride; w.ring = h->d_ring; w.partials = h->d_partials; w.tickets = h->d_tickets; w.states = h->d_states;
    w.cache = reinterpret_cast<long long*>(h->d_match + 2 * h->match_cap);
    w.certify = h->certify;
    w.lane_min = h->lane_min;
    w.seed_reach = h->seed_reach;
    w.rot = h->alone ? h->xcd_rot : -1;   // (icp.hip m3d_map_block: one pair per XCD only when the caller said the batch has the GPU to itself)
    w.alone = h->alone;
    // (lean per LEVEL: only on the finest level of a registration — a pyramid's coarse levels put more points into a bucket than a tile image holds,
    // their queries would all take the fallback list: config 5 with lean on every level took 6.5 instead of 5.9 ms)
    // (one-level registrations only; on the finest level of a pyramid it was measured on config 5: 4.21 vs 4.14 ms, no gain, and a crowded finest level would leave every query pending)
    const bool lean_here = h->params.n_levels == 1;
    // (tile images exist on a target's FINEST level only: on a pyramid's coarser levels nothing is ever binned, and an empty k_nn_tiles launch costs its 5 us in
    // every one of their iterations — 16 launches of a config-5 registration, 60 of config 2's)
    w.tiles = (h->tiles && (level < 0 || level == h->params.n_levels - 1)) ? 1 : 0;
    w.lean = (h->lean && h->batch_all_tiles && lean_here) ? 1 : 0;
    // k_nn_coop (icp.hip) answers the pairs whose target level is DENSE (m3d_dense_level: a coarse level of a pyramid, a map): behind k_nn_iter<false> where some
    // pairs of the batch are dense, alone where all are, not at all where none is — decided per batch and level in build_jobs from the clouds' own counts.
    w.coop_kernel = level >= 0 ? int(h->dense_level[level]) : 0;
    // a dense level's later iterations: most queries are certified (config 5: 74 % at the end of its first level, 93-98 % on the levels that start from a coarser
    // level's result) — from the 4th iteration of a registration's first level (1: per 64 queries) and the 2nd of every other one (2: per 128) the searchers
    // are compacted (k_nn_coop_list).
    // A function of the iteration number alone; the other choice costs time, never a bit (profiles/r04_dense_levels.txt).
    w.coop_list = level > 0 ? (it >= 1 ? 2 : 0) : (it >= 4 ? 1 : 0);
    w.ntile_max = h->ntile_max; w.rec = h->d_rec; w.recd = reinterpret_cast<float*>(h->d_rec + h->rec_cap);
    w.rec_stride = h->rec_stride; w.tcnt = h->d_tcnt; w.cnt_stride = h->cnt_stride;
    w.wcount = reinterpret_cast<unsigned int*>(h->d_witems); w.witems = h->d_witems ? h->d_witems + 16 * M3D_TILE_LISTS : nullptr; w.wcap = int(h->witems_cap);
    return w;
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
int build_jobs(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs, int& max_n_src, int& max_n_tgt) {
    const m3dreg_params& P = h->params;
    max_n_src = 0; max_n_tgt = 0;
    h->batch_all_tiles = h->tiles != 0;
    bool dense_any[M3DREG_MAX_LEVELS] = {}, dense_all[M3DREG_MAX_LEVELS], dense_unknown = false;
    for (int l = 0; l < M3DREG_MAX_LEVELS; l++) dense_all[l] = true;
    for (size_t i = 0; i < n_pairs; i++) {
        const m3dreg_cloud* s = pairs[i].source;
        const m3dreg_cloud* t = pairs[i].target;
        if (!s || !t) return fail(h, M3DREG_ERR_INVALID_ARG, "null cloud in pair");
        int rc = check_levels(h, t);
        if (rc) return rc;
        if (t->source_only) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "a source-only cloud (m3dreg_cloud_desc.source_only) cannot be the target of a registration");
        if (P.metric == M3DREG_POINT_TO_PLANE && !t->has_normals) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "target cloud has no normals");
        for (const m3dreg_cloud* c : { s, t })   // a cloud bucketed on ANOTHER handle's stream: this stream waits for that pipeline
            if (c->owner && c->owner != h && c->ready) HIPCHK(h, hipStreamWaitEvent(h->stream, c->ready->ev, 0));
        if (s->n > max_n_src) max_n_src = s->n;   // launch geometry only (the finite count stays on the device; results do not depend on it)
        if (t->n > max_n_tgt) max_n_tgt = t->n;
        if (!t->has_tiles) h->batch_all_tiles = false;   // (then the tile iterations keep the full k_nn_iter, which walks such a pair's searches itself)
        // the dense-level schedule: the same rule on the same numbers as the device's (k_patch_jobs: J.coop_always), when the host has them — a cloud whose geometry
        // and counts were read back (every synchronous creation call does; m3dreg_cloud_status / _grid_info / _density do) — else "unknown"
        if (s->meta_ready && t->meta_ready && !s->err && !t->err) {
            for (int l = 0; l < P.n_levels; l++) {
                const bool d = m3d_dense_level(uint32_t(t->n_valid), t->lv[l].n_cells_host, uint32_t(s->n_valid), l);
                dense_any[l] = dense_any[l] || d; dense_all[l] = dense_all[l] && d;
            }
        } else dense_unknown = true;
        for (int l = 0; l < P.n_levels; l++) {
            M3dJob& J = h->h_jobs[size_t(l) * h->cap_pairs + i];
            memset(&J, 0, sizeof(J));
            J.src = s->lv[s->n_levels - 1].src3; J.n_src = 0; J.metric = P.metric;   // n_src, tgt.g, exps, S: k_patch_jobs, from the clouds' device-side meta
            J.src_dyn = s->lv[s->n_levels - 1].dyn;
            J.src_order = s->lv[s->n_levels - 1].order; J.src_nblk = (s->n + 255) / 256;
            J.tgt = level_dev(t->lv[l], t->has_tiles && h->tiles);
            J.dmax = P.max_corr_dist[l];
            J.dmax2 = P.max_corr_dist[l] * P.max_corr_dist[l];
            J.min_corr = P.min_correspondences;
            J.last_level = (l == P.n_levels - 1) ? 1 : 0;
            J.eps_rot2 = P.eps_rot * P.eps_rot;
            J.eps_trans2 = P.eps_trans * P.eps_trans;
            J.pivot_rel_tol = P.pivot_rel_tol;
            J.st = h->d_states + i;
            J.ring = h->d_ring + size_t(i) * 32 * 12;
            J.prev_pts = (l > 0 && P.iterations[l - 1] > 0) ? t->lv[l - 1].pts : nullptr;   // (the coarser level ran: its last iteration left a match for every query)
            J.trace = (i == 0) ? h->d_trace : nullptr;
        }
        M3dPairState& S = h->h_states[i];
        memset(&S, 0, sizeof(S));
        for (int k = 0; k < 16; k++) S.T[k] = double(pairs[i].init_T[k]);
        S.status = M3DREG_MAX_ITERATIONS;
    }
    for (int l = 0; l < M3DREG_MAX_LEVELS; l++) h->dense_level[l] = uint8_t(dense_unknown ? 1 : (dense_all[l] ? 2 : (dense_any[l] ? 1 : 0)));
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

// fold the recorded events into the running totals (requires the stream to be idle). Per batch the stream holds
// k0 k1 k0 k1 ... k0 k1 end: k0/k1 bracket the dominant kernel of an iteration, and k0 -> next k0 (or end) is the whole
// iteration (search + reduction + solve) — two event records per iteration instead of four (each one is a barrier packet
// on the queue and cost ~4 us of the ~60 us iterations it was measuring).
void drain_events(m3dreg_handle* h) {
    hipEvent_t k0 = nullptr, k1 = nullptr, b0 = nullptr, c0 = nullptr;
    for (size_t i = 0; i < h->ev_used && i < h->ev_kind.size(); i++) {
        float ms = 0.f;
        const int kind = h->ev_kind[i];
        hipEvent_t e = h->ev_pool[i];
        if (kind == 3) { b0 = e; continue; }   // bucketing batch: begin / end
        if (kind == 5) { c0 = e; continue; }   // all iterations of a batch (the SHIPPED schedule, fused launches included): begin / end
        if (kind >= 6) { if (c0 && hipEventElapsedTime(&ms, c0, e) == hipSuccess) { h->prof_ms[4] += double(ms); h->prof_launches[4] += uint64_t(kind - 6); } c0 = nullptr; continue; }   // kind = 6 + iterations enqueued
        if (kind == 4) { if (b0 && hipEventElapsedTime(&ms, b0, e) == hipSuccess) { h->prof_ms[2] += double(ms); h->prof_launches[2]++; } b0 = nullptr; continue; }
        if (kind == 1) { k1 = e; if (k0 && hipEventElapsedTime(&ms, k0, e) == hipSuccess) { h->prof_ms[1] += double(ms); h->prof_launches[1]++; } continue; }
        // kind 0 (start of a bracketed iteration) or 2 (end of one): both close the iteration that k0 opened
        if (k0 && hipEventElapsedTime(&ms, k0, e) == hipSuccess) { h->prof_ms[0] += double(ms); h->prof_launches[0]++; }
        if (k0 && k1 && hipEventElapsedTime(&ms, k1, e) == hipSuccess) { h->prof_ms[3] += double(ms); h->prof_launches[3]++; }
        k0 = (kind == 0) ? e : nullptr; k1 = nullptr;
    }
    h->ev_used = 0;
    h->ev_kind.clear();
}

// roctx ranges around the stages (M3DREG_ROCTX=1): libroctx64 is looked up at run time, the library does not link it
void roctx_push(const char* name) {
    static int (*push)(const char*) = [] {
        const char* v = getenv("M3DREG_ROCTX");
        if (!v || !atoi(v)) return (int (*)(const char*))nullptr;
        void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        return lib ? (int (*)(const char*))dlsym(lib, "roctxRangePushA") : nullptr;
    }();
    if (push) push(name);
}
void roctx_pop() {
    static int (*pop)() = [] {
        const char* v = getenv("M3DREG_ROCTX");
        if (!v || !atoi(v)) return (int (*)())nullptr;
        void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        return lib ? (int (*)())dlsym(lib, "roctxRangePop") : nullptr;
    }();
    if (pop) pop();
}

void release_handle(m3dreg_handle* h) {
    hipSetDevice(h->device);
    sync_handle(h);
    if (h->done_ev) hipEventDestroy(h->done_ev);
    for (Block& b : h->pool) hipFree(b.p);
    if (h->ws.p) hipFree(h->ws.p);
    if (h->h_ws) hipHostFree(h->h_ws);
    for (void* p : { (void*)h->d_jobs, (void*)h->d_ring, (void*)h->d_trace, (void*)h->d_match, (void*)h->d_partials, (void*)h->d_tickets, (void*)h->d_rec, (void*)h->d_tcnt, (void*)h->d_witems }) if (p) hipFree(p);   // (the states live in the jobs' block)
    for (void* p : { (void*)h->h_jobs, (void*)h->h_trace, (void*)h->h_progress }) if (p) hipHostFree(p);
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->staged) hipEventDestroy(h->staged);
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
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
    return m3d_guarded(nullptr, "m3dreg_default_params", [&]() -> int {
    if (!p) return M3DREG_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    // Coarse to fine (ABI 4; was a single 0.1 m level): the node is launched WITHOUT parameters (m3d_husky_bringup.launch:13), so the defaults
    // must register what a rotating-lidar unit on a moving platform produces. Measured on the synthetic HDL-32 pairs (oracle == HIP): the
    // single level converges up to 1.0 m / 6 deg of initial offset and is lost at 1.5 m / 10 deg; 0.4 m -> 0.1 m converges up to 2.0 m /
    // 15 deg to the same 0.001 m / 0.04 deg, at the price of one more grid per cloud.
    p->n_levels = 2;
    p->leaf[0] = 0.4f; p->iterations[0] = 20; p->max_corr_dist[0] = 1.5f;
    p->leaf[1] = 0.1f; p->iterations[1] = 20; p->max_corr_dist[1] = 0.5f;
    p->metric = M3DREG_POINT_TO_PLANE; p->min_correspondences = 10;
    p->eps_rot = 1e-5; p->eps_trans = 1e-5; p->pivot_rel_tol = 1e-9;
    p->plane_ratio = 0.25f; p->normal_min_pts = 5; p->normal_leaf = 0.4f; p->normal_min_spread = 0.25f;
    return M3DREG_OK;
    });
}

int m3dreg_create(const m3dreg_params* params, int device, void* stream, m3dreg_handle** out) {
    return m3d_guarded(nullptr, "m3dreg_create", [&]() -> int {
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
    alloc_point();
    m3dreg_handle* h = new m3dreg_handle();
    h->device = device;
    h->params = *params;
    { static std::atomic<int> created{0}; h->xcd_rot = (3 * created.fetch_add(1)) & 7; }   // (a creation counter, read once: which XCD a handle's k-th pair lands on — placement, not behaviour)
    if (const char* v = getenv("M3DREG_CERTIFY")) h->certify = atoi(v) ? 1 : 0;
    if (const char* v = getenv("M3DREG_TILES")) h->tiles = atoi(v) ? 1 : 0;
    if (const char* v = getenv("M3DREG_LEAN")) h->lean = atoi(v) != 0;
    if (const char* v = getenv("M3DREG_FUSE_FROM")) { int q = atoi(v); if (q >= 0) h->fuse_from = q; }
    if (const char* v = getenv("M3DREG_TILE_ITERS")) { int q = atoi(v); if (q >= 1) h->tile_iters = q; }
    if (stream) { h->stream = static_cast<hipStream_t>(stream); h->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return M3DREG_ERR_HIP; }
        h->own_stream = true;
    }
    *out = h;
    return M3DREG_OK;
    });
}

int m3dreg_destroy(m3dreg_handle* h) {
    return m3d_guarded(nullptr, "m3dreg_destroy", [&]() -> int {
    if (!h || h->closed) return M3DREG_ERR_INVALID_ARG;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (h->target) { m3dreg_cloud* t = h->target; h->target = nullptr; free_cloud(h, t); }
    h->closed = true;
    if (h->live_clouds == 0) release_handle(h);   // else: the last m3dreg_cloud_destroy of a cloud it owns releases it
    return M3DREG_OK;
    });
}

void* m3dreg_get_stream(m3dreg_handle* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

int m3dreg_synchronize(m3dreg_handle* h) {
    return m3d_guarded(h, "m3dreg_synchronize", [&]() -> int {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return M3DREG_OK;
    });
}

static int check_input(m3dreg_handle* h, const m3dreg_cloud_desc& d, CloudInput& ci) {
    if (!d.data || d.n == 0 || d.n > 0x0FFFFFFFull) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create: bad argument (at most 2^28 - 1 points per cloud)");
    if (d.off_x + 4 > d.point_step || d.off_y + 4 > d.point_step || d.off_z + 4 > d.point_step || d.point_step > 0x7FFFFFFFull)
        return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create: field offsets outside point_step");
    ci.data = d.data; ci.n = d.n; ci.step = d.point_step; ci.ox = d.off_x; ci.oy = d.off_y; ci.oz = d.off_z;
    ci.is_device = d.data_is_device != 0;
    ci.src_only = d.source_only != 0;
    // device code reads 4-byte aligned floats; anything else is repacked on the host first
    ci.aligned = (d.point_step % 4 == 0) && (d.off_x % 4 == 0) && (d.off_y % 4 == 0) && (d.off_z % 4 == 0) &&
                 (reinterpret_cast<uintptr_t>(d.data) % 4 == 0);
    if (ci.is_device && !ci.aligned) return fail(h, M3DREG_ERR_INVALID_ARG, "device payloads must be 4-byte aligned");
    return M3DREG_OK;
}

int m3dreg_cloud_create_batch_async(m3dreg_handle* h, const m3dreg_cloud_desc* descs, size_t n_clouds, m3dreg_cloud** out) {
    return m3d_guarded(h, "m3dreg_cloud_create_batch_async", [&]() -> int {
    if (!h || !descs || !out || n_clouds == 0 || n_clouds > 4096) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_batch: bad argument");
    if (h->closed) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_batch: the handle was destroyed (it only lives on until its last cloud is released)");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<CloudInput> in(n_clouds);
    for (size_t i = 0; i < n_clouds; i++) { out[i] = nullptr; int rc = check_input(h, descs[i], in[i]); if (rc) return rc; }
    return create_clouds(h, in.data(), n_clouds, out);
    });
}

int m3dreg_cloud_create_batch(m3dreg_handle* h, const m3dreg_cloud_desc* descs, size_t n_clouds, m3dreg_cloud** out) {
    return m3d_guarded(h, "m3dreg_cloud_create_batch", [&]() -> int {
    const int rc = m3dreg_cloud_create_batch_async(h, descs, n_clouds, out);
    return rc ? rc : finish_sync(h, out, n_clouds);
    });
}

int m3dreg_cloud_status(m3dreg_handle* h, const m3dreg_cloud* c) {
    return m3d_guarded(h, "m3dreg_cloud_status", [&]() -> int {
    if (!h || !c) return M3DREG_ERR_INVALID_ARG;
    return fetch_meta(h, const_cast<m3dreg_cloud*>(c));
    });
}

int m3dreg_cloud_create(m3dreg_handle* h, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                        int data_is_device, m3dreg_cloud** out) {
    return m3d_guarded(h, "m3dreg_cloud_create", [&]() -> int {
    if (!h || !out) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create: bad argument");
    m3dreg_cloud_desc d;
    d.data = data; d.n = n; d.point_step = point_step; d.off_x = off_x; d.off_y = off_y; d.off_z = off_z;
    d.data_is_device = (data_is_device & M3DREG_CLOUD_DEVICE) ? 1 : 0;
    d.source_only = (data_is_device & M3DREG_CLOUD_SOURCE_ONLY) ? 1 : 0;
    return m3dreg_cloud_create_batch(h, &d, 1, out);
    });
}

// the whole sensor_msgs/PointCloud2 layout contract (what pcl::fromPCLPointCloud2 resolves by field name,
// m3d_aggregator.cpp:243-246), decoded on the device
int m3dreg_cloud_create_pc2(m3dreg_handle* h, const void* data, size_t data_bytes, uint32_t width, uint32_t height, uint32_t point_step,
                            uint32_t row_step, const m3dreg_point_field* fields, size_t n_fields, int is_bigendian, int data_is_device,
                            m3dreg_cloud** out) {
    return m3d_guarded(h, "m3dreg_cloud_create_pc2", [&]() -> int {
    if (!h || !out) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: bad argument");
    *out = nullptr;
    if (!data || !fields || width == 0 || height == 0 || point_step == 0) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: empty message");
    const size_t n = size_t(width) * size_t(height);
    if (n > 0x0FFFFFFFull) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: at most 2^28 - 1 points per cloud");
    if (row_step == 0) row_step = width * point_step;   // some producers leave it unset on unorganised clouds
    if (size_t(row_step) < size_t(width) * point_step || size_t(row_step) * height > data_bytes || size_t(row_step) * height > 0x7FFFFFFFull)
        return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: width * point_step / row_step * height do not fit the data");
    CloudInput ci{};
    ci.data = data; ci.n = n; ci.step = point_step; ci.is_device = (data_is_device & M3DREG_CLOUD_DEVICE) != 0; ci.aligned = true;
    ci.src_only = (data_is_device & M3DREG_CLOUD_SOURCE_ONLY) != 0;
    ci.generic = true; ci.width = width; ci.row_step = row_step; ci.data_bytes = size_t(row_step) * height; ci.bigendian = is_bigendian != 0;
    size_t* off[3] = { &ci.ox, &ci.oy, &ci.oz };
    bool have[3] = { false, false, false };
    for (size_t f = 0; f < n_fields; f++) {
        if (!fields[f].name) continue;
        const int a = !strcmp(fields[f].name, "x") ? 0 : (!strcmp(fields[f].name, "y") ? 1 : (!strcmp(fields[f].name, "z") ? 2 : -1));
        if (a < 0) continue;   // intensity, ring, rgb ...: not read by the registration path
        if (fields[f].datatype != M3DREG_FLOAT32 && fields[f].datatype != M3DREG_FLOAT64)
            return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: x/y/z must be FLOAT32 or FLOAT64");
        if (fields[f].count < 1) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: x/y/z field with count 0");
        const size_t sz = fields[f].datatype == M3DREG_FLOAT64 ? 8 : 4;
        if (size_t(fields[f].offset) + sz > point_step) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: field outside point_step");
        *off[a] = fields[f].offset; ci.f64[a] = sz == 8; have[a] = true;
    }
    if (!have[0] || !have[1] || !have[2]) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_create_pc2: the message lacks an x, y or z field");
    // the aggregator's own layout takes the fast path (one coalesced 16-B read per point)
    if (!ci.bigendian && !ci.f64[0] && !ci.f64[1] && !ci.f64[2] && row_step == width * point_step && point_step % 4 == 0 && ci.ox % 4 == 0 &&
        ci.oy % 4 == 0 && ci.oz % 4 == 0 && reinterpret_cast<uintptr_t>(data) % 4 == 0)
        ci.generic = false;
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = create_clouds(h, &ci, 1, out);
    return rc ? rc : finish_sync(h, out, 1);
    });
}

int m3dreg_cloud_destroy(m3dreg_handle* h, m3dreg_cloud* c) {
    return m3d_guarded(h, "m3dreg_cloud_destroy", [&]() -> int {
    if (!h || !c) return M3DREG_ERR_INVALID_ARG;
    free_cloud(h, c);   // the block returns to the handle's cache; later users are ordered on the same stream
    return M3DREG_OK;
    });
}

// ---- one batch on one handle, in three parts: set-up, one Gauss-Newton iteration per call, tail -----------------------------------------
static int batch_begin(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs) {
    if (!h || !pairs || n_pairs == 0 || n_pairs > 65535) return fail(h, M3DREG_ERR_INVALID_ARG, "align_batch: bad argument");
    if (h->closed) return fail(h, M3DREG_ERR_INVALID_ARG, "align_batch: the handle was destroyed");
    if (h->pending_pairs) return fail(h, M3DREG_ERR_INVALID_ARG, "align_batch_async: a batch is pending on this handle (a handle holds the state of ONE batch: call m3dreg_batch_wait, or use another handle on the same stream)");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_batch(h, n_pairs);
    if (rc) return rc;
    int max_n_src = 0, max_n_tgt = 0;
    if ((rc = build_jobs(h, pairs, n_pairs, max_n_src, max_n_tgt))) return rc;
    const m3dreg_params& P = h->params;
    if ((rc = ensure_match(h, n_pairs, max_n_src, max_n_tgt))) return rc;
    if (poison_mode().on) {   // (diagnosis: nothing is carried across registrations — per-query records, query slabs, partial sums and pose rings may hold anything; the
        // counters that every launch leaves at zero — tickets, per-tile counts, work-item counts — are part of the protocol and stay)
        HIPCHK(h, poison(h->d_match, sizeof(int) * 4 * h->match_cap, h->stream));
        HIPCHK(h, poison(h->d_partials, sizeof(long long) * h->partials_cap, h->stream));
        HIPCHK(h, poison(h->d_ring, sizeof(float) * 32 * 12 * h->cap_pairs, h->stream));
        if (h->d_rec) HIPCHK(h, poison(h->d_rec, (sizeof(float4) + sizeof(float)) * h->rec_cap, h->stream));
        if (h->d_witems) HIPCHK(h, poison(reinterpret_cast<uint8_t*>(h->d_witems) + 128 * M3D_TILE_LISTS, sizeof(uint2) * h->witems_cap * M3D_TILE_LISTS, h->stream));
    }
    HIPCHK(h, hipMemcpyAsync(h->d_jobs, h->h_jobs, sizeof(M3dJob) * h->cap_pairs * M3DREG_MAX_LEVELS + sizeof(M3dPairState) * n_pairs, hipMemcpyHostToDevice, h->stream));   // jobs + states: one block, one copy
    HIPCHK(h, m3d_launch_patch_jobs(h->stream, h->d_jobs, int(n_pairs), int(h->cap_pairs), P.n_levels));   // table geometry, device to device
    m3dreg_handle::Run& R = h->run;
    R = m3dreg_handle::Run();
    R.n_pairs = n_pairs; R.max_n_src = max_n_src;
    R.can_stop_early = h->h_progress && (P.eps_rot > 0.0 || P.eps_trans > 0.0);
    R.iters_before = h->launched_iters;
    h->chain_bracketed = false;
    if (h->profiling && (h->prof_batches++ % uint64_t(h->prof_batch_every)) == 0) {   // the whole chain of this batch's iterations as it ships
        hipEvent_t e = next_event(h);
        if (e) { h->ev_kind.push_back(5); (void)hipEventRecord(e, h->stream); h->chain_bracketed = true; }
    }
    return M3DREG_OK;
}

// enqueues the batch's next iteration: 1 = one was enqueued, 0 = there is none left, < 0 = error
static int batch_step(m3dreg_handle* h) {
    m3dreg_handle::Run& R = h->run;
    const m3dreg_params& P = h->params;
    for (;;) {
        if (R.l >= P.n_levels) return 0;
        if (R.it >= P.iterations[R.l]) { R.l++; R.it = 0; continue; }
        if (R.it == 0) R.level_first_seq = h->seq + 1;
        if (R.can_stop_early && R.it > 0) {   // nothing left to do at this level? (a stale value only delays the exit)
            if (h->throttle) {
                // The SYNCHRONOUS call (the host waits for the batch anyway): never more than M3D_AHEAD iterations ahead of the device. The host enqueues an
                // iteration in a quarter of the time the device takes to run it, so by the time a level's last useful iteration ran, a dozen more were queued —
                // launches that find the level finished and leave, ~5 us each (config 2: 29 of its 212 launches). Four queued iterations are >= 100 us of work:
                // the device never runs dry. (Not in m3dreg_align_batch_async: a caller that feeds several handles must not be held up inside one of them.)
#ifndef M3D_AHEAD_ITERS
#define M3D_AHEAD_ITERS 4
#endif
                constexpr unsigned int M3D_AHEAD = M3D_AHEAD_ITERS;
                // (bounded by TIME, ADVICE r5: a GPU that is busy with other handles' work — a batch queued ahead on a shared stream — must not cost this caller a
                // core for milliseconds per iteration: after one wait of more than 2 ms the rest of this batch is enqueued unthrottled)
                if (!R.throttle_off) {
                    const auto t0 = std::chrono::steady_clock::now();
                    for (unsigned int spin = 1;; spin++) {
                        const unsigned int done = (unsigned int)(*h->h_progress >> 32);
                        if (h->seq - done <= M3D_AHEAD || done > h->seq) break;   // (done > seq: a word of an earlier life of the counter)
                        m3d_cpu_relax();
                        if ((spin & 63u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) { R.throttle_off = true; break; }
                    }
                }
            }
            const unsigned long long v = *h->h_progress;
            if ((unsigned int)(v >> 32) >= R.level_first_seq && (unsigned int)(v >> 32) <= h->seq && (unsigned int)v == 0u) {
                h->skipped_iters += uint64_t(P.iterations[R.l] - R.it);
                R.l++; R.it = 0;
                continue;
            }
        }
        const int l = R.l, it = R.it;
        const M3dJob* dj = h->d_jobs + size_t(l) * h->cap_pairs;
        // (the fused late launches only on the finest level: a pyramid's coarser levels are the crowded ones, where the uncertified few of a late
        // iteration are long cooperative walks behind a fat workgroup's stream — config 5 with them fused from the 8th iteration: 4.12 instead of 3.82 ms)
        const int fuse_from_l = (l < P.n_levels - 1) ? 0 : h->fuse_from;
        h->seq++;
        roctx_push("m3dreg:iteration");
        hipEvent_t k0 = nullptr, k1 = nullptr;
        if (h->profiling) {   // every prof_every-th iteration is bracketed: {k0, k1, end of the iteration}
            if (R.prev_sampled) { hipEvent_t e = next_event(h); if (e) { h->ev_kind.push_back(2); (void)hipEventRecord(e, h->stream); } }
            R.prev_sampled = (h->launched_iters % uint64_t(h->prof_every)) == 0;
            if (R.prev_sampled) {
                k0 = next_event(h); if (k0) h->ev_kind.push_back(0);
                k1 = k0 ? next_event(h) : nullptr; if (k1) h->ev_kind.push_back(1);
                if (!k1) { k0 = nullptr; R.prev_sampled = false; }   // (an event could not be created: this iteration is not bracketed)
            }
        }
        const M3dNnWork nw = nn_work(h, l, it);
        int fol = it == 0 ? 1 : (it >= h->tile_iters ? ((fuse_from_l > 0 && it >= fuse_from_l) ? -2 : -1) : 0);
        // (a dense level's mostly-certified iterations as ONE launch, k_icp_late with its row-by-row walk, were measured again in round 4: config 5 1.76 -> 2.80 ms)
        HIPCHK(h, m3d_launch_icp_iteration(h->stream, dj, int(R.n_pairs), R.max_n_src, P.metric, fol, nw, h->seq, R.can_stop_early ? h->d_progress : nullptr, k0, k1));
        roctx_pop();
        h->launched_iters++;
        R.it++;
        return 1;
    }
}

static int batch_end(m3dreg_handle* h, const m3dreg_pair* pairs) {
    m3dreg_handle::Run& R = h->run;
    const size_t n_pairs = R.n_pairs;
    int rc;
    if (h->profiling && R.prev_sampled) { hipEvent_t e = next_event(h); if (e) { h->ev_kind.push_back(2); (void)hipEventRecord(e, h->stream); } }
    if (h->chain_bracketed) { hipEvent_t e = next_event(h); if (e) { h->ev_kind.push_back(6 + int(h->launched_iters - R.iters_before)); (void)hipEventRecord(e, h->stream); } h->chain_bracketed = false; }
    for (size_t i = 0; i < n_pairs; i++) {   // clouds of other handles: their owners' streams wait for this batch before the blocks are re-used
        if ((rc = note_foreign_use(h, pairs[i].source))) return rc;
        if ((rc = note_foreign_use(h, pairs[i].target))) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_states, h->d_states, sizeof(M3dPairState) * n_pairs, hipMemcpyDeviceToHost, h->stream));
    // m3dreg_batch_wait waits for THIS point of the stream, not for the stream: handles that share a stream queue batches behind each other (bench.py: two per
    // stream), and a wait for the whole stream also waited for the batch queued behind this one — the stream then ran dry until the host had enqueued the next
    if (!h->done_ev) HIPCHK(h, hipEventCreateWithFlags(&h->done_ev, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->done_ev, h->stream));
    h->done_recorded = true;
    // (the pose trace of pair 0 stays on the device: m3dreg_debug_trace fetches it when asked — a 32 KB copy per batch otherwise)
    h->pending_pairs = n_pairs;
    return M3DREG_OK;
}

int m3dreg_align_batch_async(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs) {
    return m3d_guarded(h, "m3dreg_align_batch_async", [&]() -> int {
    int rc = batch_begin(h, pairs, n_pairs);
    if (rc) return rc;
    while ((rc = batch_step(h)) > 0) {}
    if (rc < 0) return rc;
    return batch_end(h, pairs);
    });
}

int m3dreg_batch_wait(m3dreg_handle* h, float* out_T, m3dreg_stats* stats) {
    return m3d_guarded(h, "m3dreg_batch_wait", [&]() -> int {
    if (!h || h->pending_pairs == 0) return fail(h, M3DREG_ERR_INVALID_ARG, "batch_wait: nothing pending");
    if (h->done_recorded) { HIPCHK(h, hipEventSynchronize(h->done_ev)); h->done_recorded = false; }
    else HIPCHK(h, hipStreamSynchronize(h->stream));
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
    });
}

int m3dreg_set_latency_mode(m3dreg_handle* h, int on) {
    if (!h || h->pending_pairs) return M3DREG_ERR_INVALID_ARG;   // (not between an asynchronous call and its wait: the batch's grids are laid out)
    h->alone = on ? 1 : 0;
    return M3DREG_OK;
}

int m3dreg_align_batch(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs, float* out_T, m3dreg_stats* stats) {
    return m3d_guarded(h, "m3dreg_align_batch", [&]() -> int {
    if (!h || !pairs || n_pairs == 0 || n_pairs > 65535) return fail(h, M3DREG_ERR_INVALID_ARG, "align_batch: bad argument");
    // ONE launch chain per batch. (ABI 6-7 could cut a batch into internal chains on child handles' streams, m3dreg_set_batch_chains: measured to lose for 100 k-point
    // pairs in round 5 and for 10 k / 32 k-point pairs in round 6 — profiles/r05_batch_chains.txt, profiles/r06_batch_chains.txt — and removed in ABI 8.)
    h->throttle = true;   // this call waits for the batch anyway: the enqueue of a convergence-terminated batch stays a few iterations ahead of the device (batch_step)
    int rc = m3dreg_align_batch_async(h, pairs, n_pairs);
    h->throttle = false;
    if (rc) return rc;
    return m3dreg_batch_wait(h, out_T, stats);
    });
}

int m3dreg_align_clouds(m3dreg_handle* h, const m3dreg_cloud* source, const m3dreg_cloud* target, const float init_T[16], float out_T[16],
                        m3dreg_stats* stats) {
    return m3d_guarded(h, "m3dreg_align_clouds", [&]() -> int {
    if (!h || !source || !target || !init_T || !out_T) return fail(h, M3DREG_ERR_INVALID_ARG, "align_clouds: bad argument");
    m3dreg_pair p;
    p.source = source; p.target = target;
    memcpy(p.init_T, init_T, sizeof(float) * 16);
    return m3dreg_align_batch(h, &p, 1, out_T, stats);
    });
}

int m3dreg_set_target_xyz(m3dreg_handle* h, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z) {
    return m3d_guarded(h, "m3dreg_set_target_xyz", [&]() -> int {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    m3dreg_cloud* c = nullptr;
    int rc = m3dreg_cloud_create(h, data, n, point_step, off_x, off_y, off_z, 0, &c);
    if (rc) return rc;
    if (h->target) free_cloud(h, h->target);
    h->target = c;
    return M3DREG_OK;
    });
}

int m3dreg_align(m3dreg_handle* h, const void* src, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                 const float init_T[16], float out_T[16], m3dreg_stats* stats) {
    return m3d_guarded(h, "m3dreg_align", [&]() -> int {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    if (!h->target) return fail(h, M3DREG_ERR_NO_TARGET, "m3dreg_align before m3dreg_set_target_xyz");
    m3dreg_cloud* s = nullptr;
    int rc = m3dreg_cloud_create(h, src, n, point_step, off_x, off_y, off_z, 0, &s);
    if (rc) return rc;
    rc = m3dreg_align_clouds(h, s, h->target, init_T, out_T, stats);
    m3dreg_cloud_destroy(h, s);
    return rc;
    });
}

// ---- aggregation on the device --------------------------------------------------------
}  // extern "C"

struct m3dagg {
    m3dreg_handle* h = nullptr;
    double bb[6]{};
    double current_angle = 0.0;
    double angular_distance = 1.1 * 3.14159265358979323846;   // 1.1 * M_PI, m3d_aggregator.cpp:30
    bool creating = true, first_scan = true;
    double actual[4]{};
    float4* d_pts = nullptr;       // pcl::PointXYZ layout
    uint32_t* d_count = nullptr;   // {points, overflow}
    uint32_t* d_blocks = nullptr;
    uint8_t* d_stage = nullptr;    // staged message payload
    size_t capacity = 0, stage_bytes = 0, blocks_cap = 0;
    int scan_trig_float = 0;       // m3dagg_set_scan_trig: 0 = cos(double) (default), 1 = the float overload
    bool auto_rearm = true;        // m3dagg_set_rearm: m3dagg_take_cloud re-arms the aggregator itself (default) / leaves it idle until m3dagg_restart, like the reference's node
};

namespace {
// tf LinearMath (un-vendored dependency of m3d_aggregator): Matrix3x3::setRotation / getRotation and
// Quaternion::angleShortestPath restated in double with the published operation order
void tf_set_rotation(const double q[4], double m[9]) {
    const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double s = 2.0 / d;
    const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
    const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
    const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
    const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
    m[0] = 1.0 - (yy + zz); m[1] = xy - wz; m[2] = xz + wy;
    m[3] = xy + wz; m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
    m[6] = xz - wy; m[7] = yz + wx; m[8] = 1.0 - (xx + yy);
}
void tf_get_rotation(const double m[9], double q[4]) {
    const double trace = m[0] + m[4] + m[8];
    double temp[4];
    if (trace > 0.0) {
        double s = std::sqrt(trace + 1.0);
        temp[3] = s * 0.5;
        s = 0.5 / s;
        temp[0] = (m[7] - m[5]) * s; temp[1] = (m[2] - m[6]) * s; temp[2] = (m[3] - m[1]) * s;
    } else {
        const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = std::sqrt(m[3 * i + i] - m[3 * j + j] - m[3 * k + k] + 1.0);
        temp[i] = s * 0.5;
        s = 0.5 / s;
        temp[3] = (m[3 * k + j] - m[3 * j + k]) * s;
        temp[j] = (m[3 * j + i] + m[3 * i + j]) * s;
        temp[k] = (m[3 * k + i] + m[3 * i + k]) * s;
    }
    memcpy(q, temp, sizeof(temp));
}
double tf_angle_shortest_path(const double a[4], const double b[4]) {
    const double la = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    const double lb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    const double s = std::sqrt(la * lb);
    const double dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    if (dot < 0.0) return std::acos(-dot / s) * 2.0;
    return std::acos(dot / s) * 2.0;
}

int agg_add(m3dagg* a, M3dAggArgs& A, const void* host_payload, size_t payload_bytes, const double tf7[7]) {
    m3dreg_handle* h = a->h;
    if (!a->creating) return M3DREG_OK;   // addPoints :55
    HIPCHK(h, hipSetDevice(h->device));
    const double qin[4] = { tf7[3], tf7[4], tf7[5], tf7[6] };
    tf_set_rotation(qin, A.m);
    A.o[0] = tf7[0]; A.o[1] = tf7[1]; A.o[2] = tf7[2];
    double q[4];
    tf_get_rotation(A.m, q);
    memcpy(A.bb, a->bb, sizeof(A.bb));
    if (payload_bytes > a->stage_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (a->d_stage) hipFree(a->d_stage);
        a->d_stage = nullptr; a->stage_bytes = 0;
        HIPCHK(h, m3d_malloc((void**)&a->d_stage, payload_bytes + payload_bytes / 2));
        a->stage_bytes = payload_bytes + payload_bytes / 2;
    }
    const size_t nblocks = (size_t(A.n) + 255) / 256;
    if (nblocks > a->blocks_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (a->d_blocks) hipFree(a->d_blocks);
        a->d_blocks = nullptr; a->blocks_cap = 0;
        HIPCHK(h, m3d_malloc((void**)&a->d_blocks, sizeof(uint32_t) * (nblocks + nblocks / 2 + 16)));
        a->blocks_cap = nblocks + nblocks / 2 + 16;
    }
    HIPCHK(h, hipMemcpyAsync(a->d_stage, host_payload, payload_bytes, hipMemcpyHostToDevice, h->stream));
    if (A.mode == 0) A.raw = a->d_stage; else A.ranges = reinterpret_cast<const float*>(a->d_stage);
    A.out = a->d_pts; A.count = a->d_count; A.capacity = uint32_t(a->capacity); A.block_counts = a->d_blocks;
    HIPCHK(h, m3d_launch_aggregate(h->stream, A));
    // :75-87 — one quaternion per message, but the reference evaluates it once PER POINT: the first point of a
    // message sees the rotation since the previous message, every further point adds angleShortestPath(q, q)
    // (2*acos(1 +- rounding): tiny or NaN, skipped when NaN). Reproduced add by add to keep the same double.
    for (int i = 0; i < A.n; i++) {
        if (a->first_scan) { a->first_scan = false; memcpy(a->actual, q, sizeof(q)); }
        else {
            const double dd = tf_angle_shortest_path(q, a->actual);
            if (!std::isnan(dd)) a->current_angle = a->current_angle + dd;
            memcpy(a->actual, q, sizeof(q));
        }
    }
    return M3DREG_OK;
}
}  // namespace

extern "C" {

int m3dagg_create(m3dreg_handle* h, const double bbox[6], size_t capacity, m3dagg** out) {
    return m3d_guarded(h, "m3dagg_create", [&]() -> int {
    if (!h || !bbox || !out || capacity == 0 || capacity >= 0x7FFFFFFFull) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dagg_create: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    m3dagg* a = new m3dagg();
    a->h = h; a->capacity = capacity;
    memcpy(a->bb, bbox, sizeof(a->bb));
    hipError_t e = m3d_malloc((void**)&a->d_pts, sizeof(float4) * capacity);
    if (e == hipSuccess) e = m3d_malloc((void**)&a->d_count, sizeof(uint32_t) * 2);
    if (e == hipSuccess) e = hipMemsetAsync(a->d_count, 0, sizeof(uint32_t) * 2, h->stream);
    if (e != hipSuccess) { if (a->d_pts) hipFree(a->d_pts); if (a->d_count) hipFree(a->d_count); delete a; return fail(h, M3DREG_ERR_HIP, "m3dagg_create", e); }
    *out = a;
    return M3DREG_OK;
    });
}

int m3dagg_destroy(m3dagg* a) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_destroy", [&]() -> int {
    if (!a) return M3DREG_ERR_INVALID_ARG;
    hipSetDevice(a->h->device);
    hipStreamSynchronize(a->h->stream);
    for (void* p : { (void*)a->d_pts, (void*)a->d_count, (void*)a->d_blocks, (void*)a->d_stage }) if (p) hipFree(p);
    delete a;
    return M3DREG_OK;
    });
}

int m3dagg_add_cloud(m3dagg* a, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z, const double tf7[7]) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_add_cloud", [&]() -> int {
    if (!a || !data || !tf7 || n == 0 || n >= 0x7FFFFFFFull) return M3DREG_ERR_INVALID_ARG;
    if (off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step || (point_step % 4) || (off_x % 4) || (off_y % 4) || (off_z % 4))
        return fail(a->h, M3DREG_ERR_INVALID_ARG, "m3dagg_add_cloud: FLOAT32 fields must be 4-byte aligned inside point_step");
    M3dAggArgs A{};
    A.mode = 0; A.n = int(n); A.step = int(point_step); A.ox = int(off_x); A.oy = int(off_y); A.oz = int(off_z);
    return agg_add(a, A, data, n * point_step, tf7);
    });
}

int m3dagg_add_scan(m3dagg* a, const float* ranges, size_t n, float angle_min, float angle_increment, const double tf7[7]) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_add_scan", [&]() -> int {
    if (!a || !ranges || !tf7 || n == 0 || n >= 0x7FFFFFFFull) return M3DREG_ERR_INVALID_ARG;
    M3dAggArgs A{};
    A.mode = a->scan_trig_float ? 2 : 1; A.n = int(n); A.angle_min = angle_min; A.angle_inc = angle_increment;
    return agg_add(a, A, ranges, n * sizeof(float), tf7);
    });
}

int m3dagg_set_scan_trig(m3dagg* a, int float_overload) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_set_scan_trig", [&]() -> int {
    if (!a || (float_overload != 0 && float_overload != 1)) return M3DREG_ERR_INVALID_ARG;
    a->scan_trig_float = float_overload;
    return M3DREG_OK;
    });
}

int m3dagg_status(m3dagg* a, double* progress, int* ready, double* angle, size_t* n_points) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_status", [&]() -> int {
    if (!a) return M3DREG_ERR_INVALID_ARG;
    if (progress) *progress = a->creating ? 0.1 * std::floor(a->current_angle * 1000.0 / a->angular_distance) : -1.0;   // :119-124
    if (ready) *ready = a->current_angle > a->angular_distance ? 1 : 0;                                                  // :95-103
    if (angle) *angle = a->current_angle;
    if (n_points) {
        uint32_t c[2];
        HIPCHK(a->h, hipMemcpyAsync(c, a->d_count, sizeof(c), hipMemcpyDeviceToHost, a->h->stream));
        HIPCHK(a->h, hipStreamSynchronize(a->h->stream));
        if (c[1]) return fail(a->h, M3DREG_ERR_INVALID_ARG, "m3dagg: aggregate capacity exceeded");
        *n_points = c[0];
    }
    return M3DREG_OK;
    });
}

int m3dagg_restart(m3dagg* a) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_restart", [&]() -> int {
    if (!a) return M3DREG_ERR_INVALID_ARG;
    HIPCHK(a->h, hipMemsetAsync(a->d_count, 0, sizeof(uint32_t) * 2, a->h->stream));
    a->current_angle = 0.0; a->first_scan = true; a->creating = true;
    return M3DREG_OK;
    });
}

int m3dagg_take_cloud(m3dagg* a, m3dreg_cloud** out) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_take_cloud", [&]() -> int {
    if (!a || !out) return M3DREG_ERR_INVALID_ARG;
    size_t n = 0;
    int rc = m3dagg_status(a, nullptr, nullptr, nullptr, &n);
    if (rc) return rc;
    if (n == 0) return fail(a->h, M3DREG_ERR_EMPTY_CLOUD, "m3dagg_take_cloud: nothing aggregated");
    rc = m3dreg_cloud_create(a->h, a->d_pts, n, 16, 0, 4, 8, 1, out);   // bucketed in place: no PCIe transfer of the sweep
    if (rc) return rc;
    rc = m3dagg_restart(a);
    if (rc == M3DREG_OK && !a->auto_rearm) { a->creating = false; a->first_scan = false; }   // clearPointCloud (:108-114): idle until the next request (:224-229)
    return rc;
    });
}

int m3dagg_set_rearm(m3dagg* a, int automatic) {
    if (!a) return M3DREG_ERR_INVALID_ARG;
    a->auto_rearm = automatic != 0;
    return M3DREG_OK;
}

int m3dagg_download(m3dagg* a, float* xyzw, size_t cap_points, size_t* n_out) {
    return m3d_guarded((a ? a->h : nullptr), "m3dagg_download", [&]() -> int {
    if (!a || !n_out) return M3DREG_ERR_INVALID_ARG;
    size_t n = 0;
    int rc = m3dagg_status(a, nullptr, nullptr, nullptr, &n);
    if (rc) return rc;
    *n_out = n;
    const size_t k = n < cap_points ? n : cap_points;
    if (xyzw && k) {
        HIPCHK(a->h, hipMemcpyAsync(xyzw, a->d_pts, 16 * k, hipMemcpyDeviceToHost, a->h->stream));
        HIPCHK(a->h, hipStreamSynchronize(a->h->stream));
    }
    return M3DREG_OK;
    });
}

// ---- calibration cost on the device --------------------------------------------------
}  // extern "C"

struct m3dcal {
    m3dreg_handle* h = nullptr;
    int axis = 1;
    std::vector<float> pts;        // host copy {x, y, z, bits(segment)} per point
    std::vector<float> seg_T;      // 12 floats per segment: row-major linear part + translation of original_Transform
    bool dirty = true;
    float4* d_pts = nullptr; size_t pts_cap = 0;
    float* d_mm = nullptr; size_t mm_cap = 0;
    unsigned long long* d_keys = nullptr; size_t keys_cap = 0;
    uint32_t* d_cnt = nullptr; size_t cnt_cap = 0;
    long long* d_sums = nullptr; size_t sums_cap = 0;
    unsigned int* d_result = nullptr; size_t res_cap = 0;
    int* d_status = nullptr; size_t st_cap = 0;
    float* h_mm = nullptr; size_t hmm_cap = 0;                 // pinned staging
    unsigned int* h_result = nullptr; size_t hres_cap = 0;
    int* h_status = nullptr; size_t hst_cap = 0;
};

namespace {
// m3d_calibration_twiddle.cpp:202-220 — Eigen's scalar code paths restated in float, left to right, no fused multiply-add
// (this file is built with -ffp-contract=off); cosf/sinf stay on the host so both sides of the parity test share one libm.
void cal_offset_matrix(const float p[6], float ol[9], float ot[3]) {
    const float ha0 = 0.5f * p[3], ha1 = 0.5f * p[4], ha2 = 0.5f * p[5];
    const float aw = std::cos(ha0), ax = std::sin(ha0) * 1.0f, ay = std::sin(ha0) * 0.0f, az = std::sin(ha0) * 0.0f;
    const float bw = std::cos(ha1), bx = std::sin(ha1) * 0.0f, by = std::sin(ha1) * 1.0f, bz = std::sin(ha1) * 0.0f;
    const float cw = std::cos(ha2), cx = std::sin(ha2) * 0.0f, cy = std::sin(ha2) * 0.0f, cz = std::sin(ha2) * 1.0f;
    const float dw = aw * bw - ax * bx - ay * by - az * bz;
    const float dx = aw * bx + ax * bw + ay * bz - az * by;
    const float dy = aw * by + ay * bw + az * bx - ax * bz;
    const float dz = aw * bz + az * bw + ax * by - ay * bx;
    const float w = dw * cw - dx * cx - dy * cy - dz * cz;
    const float x = dw * cx + dx * cw + dy * cz - dz * cy;
    const float y = dw * cy + dy * cw + dz * cx - dx * cz;
    const float z = dw * cz + dz * cw + dx * cy - dy * cx;
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    ol[0] = 1.0f - (tyy + tzz); ol[1] = txy - twz; ol[2] = txz + twy;
    ol[3] = txy + twz; ol[4] = 1.0f - (txx + tzz); ol[5] = tyz - twx;
    ol[6] = txz - twy; ol[7] = tyz + twx; ol[8] = 1.0f - (txx + tyy);
    for (int r = 0; r < 3; r++) ot[r] = 0.0f + ((ol[3 * r] * p[0] + ol[3 * r + 1] * p[1]) + ol[3 * r + 2] * p[2]);
}
// :229 mm = original_Transform * laserOffsetMatrix
void cal_compose(const float* a, const float bl[9], const float bt[3], float* o) {
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) o[3 * r + c] = (a[3 * r] * bl[c] + a[3 * r + 1] * bl[3 + c]) + a[3 * r + 2] * bl[6 + c];
        o[9 + r] = ((a[3 * r] * bt[0] + a[3 * r + 1] * bt[1]) + a[3 * r + 2] * bt[2]) + a[9 + r];
    }
}
template <typename T> int cal_grow(m3dreg_handle* h, T*& p, size_t& cap, size_t need, bool pinned = false) {
    if (need <= cap) return M3DREG_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (p) { if (pinned) hipHostFree(p); else hipFree(p); }
    p = nullptr; cap = 0;
    const size_t c = need + need / 4 + 16;
    if (pinned) HIPCHK(h, hipHostMalloc((void**)&p, sizeof(T) * c, hipHostMallocDefault));
    else HIPCHK(h, m3d_malloc((void**)&p, sizeof(T) * c));
    cap = c;
    return M3DREG_OK;
}
int cal_eval1(m3dcal* c, const float p6[6], float* err) {
    int64_t v = 0;
    int rc = m3dcal_evaluate(c, p6, 1, &v, nullptr);
    if (rc) return rc;
    *err = float(v);
    return M3DREG_OK;
}
}  // namespace

extern "C" {

int m3dcal_create(m3dreg_handle* h, int laser_up_axis, m3dcal** out) {
    return m3d_guarded(h, "m3dcal_create", [&]() -> int {
    if (!h || !out || laser_up_axis < 0 || laser_up_axis > 2) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dcal_create: bad argument");
    m3dcal* c = new m3dcal();
    c->h = h; c->axis = laser_up_axis;
    *out = c;
    return M3DREG_OK;
    });
}

int m3dcal_destroy(m3dcal* c) {
    return m3d_guarded((c ? c->h : nullptr), "m3dcal_destroy", [&]() -> int {
    if (!c) return M3DREG_ERR_INVALID_ARG;
    hipSetDevice(c->h->device);
    hipStreamSynchronize(c->h->stream);
    for (void* p : { (void*)c->d_pts, (void*)c->d_mm, (void*)c->d_keys, (void*)c->d_cnt, (void*)c->d_sums, (void*)c->d_result, (void*)c->d_status }) if (p) hipFree(p);
    for (void* p : { (void*)c->h_mm, (void*)c->h_result, (void*)c->h_status }) if (p) hipHostFree(p);
    delete c;
    return M3DREG_OK;
    });
}

int m3dcal_add_segment(m3dcal* c, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z, const float T[16]) {
    return m3d_guarded((c ? c->h : nullptr), "m3dcal_add_segment", [&]() -> int {
    if (!c || !T || (n && !data)) return M3DREG_ERR_INVALID_ARG;
    if (off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step) return fail(c->h, M3DREG_ERR_INVALID_ARG, "m3dcal_add_segment: field offsets outside point_step");
    if (c->pts.size() / 4 + n >= 0x3FFFFFFFull) return fail(c->h, M3DREG_ERR_INVALID_ARG, "m3dcal_add_segment: too many points");
    const uint32_t seg = uint32_t(c->seg_T.size() / 12);
    for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) c->seg_T.push_back(T[4 * k + r]);   // column-major in, row-major kept
    for (int r = 0; r < 3; r++) c->seg_T.push_back(T[12 + r]);
    const uint8_t* b = static_cast<const uint8_t*>(data);
    for (size_t j = 0; j < n; j++) {
        float v[4];
        memcpy(&v[0], b + j * point_step + off_x, 4); memcpy(&v[1], b + j * point_step + off_y, 4); memcpy(&v[2], b + j * point_step + off_z, 4);
        memcpy(&v[3], &seg, 4);
        c->pts.insert(c->pts.end(), v, v + 4);
    }
    c->dirty = true;
    return M3DREG_OK;
    });
}

int m3dcal_evaluate(m3dcal* c, const float* params, size_t k, int64_t* counts, int64_t* voxels) {
    return m3d_guarded((c ? c->h : nullptr), "m3dcal_evaluate", [&]() -> int {
    if (!c || !params || !counts || k == 0 || k > 4096) return M3DREG_ERR_INVALID_ARG;
    m3dreg_handle* h = c->h;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = c->pts.size() / 4, S = c->seg_T.size() / 12;
    if (n == 0) return fail(h, M3DREG_ERR_EMPTY_CLOUD, "m3dcal_evaluate: no segments");
    int rc;
    if ((rc = cal_grow(h, c->d_pts, c->pts_cap, n))) return rc;
    if (c->dirty) { HIPCHK(h, hipMemcpyAsync(c->d_pts, c->pts.data(), sizeof(float) * 4 * n, hipMemcpyHostToDevice, h->stream)); c->dirty = false; }
    uint32_t tsize = 1024; int tbits = 10;
    while (tsize < 2u * uint32_t(n)) { tsize <<= 1; tbits++; }
    // candidates per launch: bounded by a table budget of ~2 GiB
    size_t per = (size_t(2) << 30) / (size_t(tsize) * 36u);
    if (per < 1) per = 1;
    if (per > k) per = k;
    if ((rc = cal_grow(h, c->d_keys, c->keys_cap, per * tsize))) return rc;
    if ((rc = cal_grow(h, c->d_cnt, c->cnt_cap, per * tsize))) return rc;
    if ((rc = cal_grow(h, c->d_sums, c->sums_cap, per * tsize * 3))) return rc;
    if ((rc = cal_grow(h, c->d_result, c->res_cap, per * 3))) return rc;
    if ((rc = cal_grow(h, c->d_status, c->st_cap, per))) return rc;
    if ((rc = cal_grow(h, c->h_result, c->hres_cap, per * 3, true))) return rc;
    if ((rc = cal_grow(h, c->h_status, c->hst_cap, per, true))) return rc;
    if ((rc = cal_grow(h, c->d_mm, c->mm_cap, per * S * 12))) return rc;
    if ((rc = cal_grow(h, c->h_mm, c->hmm_cap, per * S * 12, true))) return rc;
    for (size_t k0 = 0; k0 < k; k0 += per) {
        const size_t kk = std::min(per, k - k0);
        HIPCHK(h, hipStreamSynchronize(h->stream));   // the pinned staging of the previous chunk has been consumed
        for (size_t q = 0; q < kk; q++) {
            float ol[9], ot[3];
            cal_offset_matrix(params + 6 * (k0 + q), ol, ot);
            for (size_t s = 0; s < S; s++) cal_compose(&c->seg_T[12 * s], ol, ot, c->h_mm + (q * S + s) * 12);
        }
        HIPCHK(h, hipMemcpyAsync(c->d_mm, c->h_mm, sizeof(float) * kk * S * 12, hipMemcpyHostToDevice, h->stream));
        M3dCalArgs A{};
        A.pts = c->d_pts; A.n = int(n); A.n_seg = int(S); A.axis = c->axis; A.mm = c->d_mm;
        A.keys = c->d_keys; A.cnt = c->d_cnt; A.sums = c->d_sums; A.tsize = tsize; A.tshift = 64 - tbits;
        A.result = c->d_result; A.status = c->d_status;
        HIPCHK(h, m3d_launch_calibration(h->stream, A, int(kk)));
        HIPCHK(h, hipMemcpyAsync(c->h_result, c->d_result, sizeof(unsigned int) * 3 * kk, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(c->h_status, c->d_status, sizeof(int) * kk, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (size_t q = 0; q < kk; q++) {
            if (c->h_status[q]) return fail(h, M3DREG_ERR_GRID_TOO_LARGE, "m3dcal_evaluate: a transformed point lies beyond +-104 km");
            counts[k0 + q] = int64_t(c->h_result[3 * q]);
            if (voxels) { voxels[2 * (k0 + q)] = int64_t(c->h_result[3 * q + 1]); voxels[2 * (k0 + q) + 1] = int64_t(c->h_result[3 * q + 2]); }
        }
    }
    return M3DREG_OK;
    });
}

// m3d_calibration_twiddle.cpp:330-396. p and dp are std::vector<float> (:514); `p[i] - 2.0 * dp[i]` and `dp[i] * 1.1` are
// double expressions rounded back to float on assignment, exactly as written here.
int m3dcal_twiddle(m3dcal* c, int max_sweeps, float p_out[5], float* best_error_out, int* sweeps, int* evaluations) {
    return m3d_guarded((c ? c->h : nullptr), "m3dcal_twiddle", [&]() -> int {
    if (!c || !p_out) return M3DREG_ERR_INVALID_ARG;
    float p[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, dp[5] = { 0.01f, 0.01f, 0.01f, 0.01f, 0.01f };
    int evals = 0, rc;
    auto test = [&](float* err) { const float q[6] = { 0.0f, p[0], p[1], p[2], p[3], p[4] }; evals++; return cal_eval1(c, q, err); };   // :345
    float best_error;
    if ((rc = test(&best_error))) return rc;
    int n = 0;
    float incr = 100.0f;
    while (incr > 0.000001) {                                     // :348
        if (max_sweeps > 0 && n >= max_sweeps) break;
        for (int i = 0; i < 5; i++) {
            p[i] = p[i] + dp[i];
            float err;
            if ((rc = test(&err))) return rc;
            if (err < best_error) { best_error = err; dp[i] = float(dp[i] * 1.1); }
            else {
                p[i] = float(p[i] - 2.0 * dp[i]);
                if ((rc = test(&err))) return rc;
                if (err < best_error) { best_error = err; dp[i] = float(dp[i] * 1.1); }
                else { p[i] = p[i] + dp[i]; dp[i] = float(dp[i] * 0.9); }
            }
        }
        incr = 0.0f;
        for (int i = 0; i < 5; i++) incr = dp[i];                 // :376-380: only the LAST dp survives
        n++;
    }
    for (int i = 0; i < 5; i++) p_out[i] = p[i];
    if (best_error_out) *best_error_out = best_error;
    if (sweeps) *sweeps = n;
    if (evaluations) *evaluations = evals;
    return M3DREG_OK;
    });
}

// m3d_calibration_sa.cpp:280-356
int m3dcal_anneal(m3dcal* c, unsigned int seed, float p[5], float* best_error_out, int* evaluations) {
    return m3d_guarded((c ? c->h : nullptr), "m3dcal_anneal", [&]() -> int {
    if (!c || !p) return M3DREG_ERR_INVALID_ARG;
    srand(seed);                                                  // :289 (time(0) there)
    auto m_rand = []() { return -1.0f + 2 * ((float)rand()) / ((float)RAND_MAX); };   // :280-283
    int evals = 0, rc;
    float best_error;
    { const float q[6] = { 0.0f, p[0], p[1], p[2], p[3], p[4] }; evals++; if ((rc = cal_eval1(c, q, &best_error))) return rc; }   // :311
    float temperature = 1.0f;
    const float alpha = 0.99;
    while (temperature > 0.001f) {                                // :316
        float c_p[5];
        for (int i = 0; i < 5; i++) c_p[i] = float(p[i] + 0.001 * m_rand());   // :319-323
        float c_error;
        { const float q[6] = { 0.0f, c_p[0], c_p[1], c_p[2], c_p[3], c_p[4] }; evals++; if ((rc = cal_eval1(c, q, &c_error))) return rc; }
        const float ac_p = float(std::exp((best_error - c_error) / temperature));   // :328 exp(float) -> std::exp overload on float
        if (ac_p > ((float)rand()) / ((float)RAND_MAX)) {         // :329
            best_error = c_error;
            for (int i = 0; i < 5; i++) p[i] = c_p[i];
        }
        temperature = alpha * temperature;                        // :340
    }
    if (best_error_out) *best_error_out = best_error;
    if (evaluations) *evaluations = evals;
    return M3DREG_OK;
    });
}

// ---- persistent map in HBM -----------------------------------------------------------
}  // extern "C"

struct m3dmap {
    m3dreg_handle* h = nullptr;
    float leaf = 0.02f;
    size_t capacity = 0;
    float4* d_pts = nullptr;
    uint32_t* d_count = nullptr;          // {points}
    uint32_t* d_flags = nullptr;          // [4]
    unsigned long long* d_keys = nullptr; uint32_t* d_epoch = nullptr; uint32_t* d_owner = nullptr;
    uint32_t tsize = 0; int tbits = 0;
    uint32_t* d_slot_of = nullptr; size_t slot_cap = 0;
    uint32_t* d_blocks = nullptr; size_t blocks_cap = 0;
    uint32_t epoch = 0;
    size_t n_host = 0;                    // points in the map as of the last insert
};

namespace {
int map_clear_device(m3dmap* m) {
    m3dreg_handle* h = m->h;
    HIPCHK(h, hipMemsetAsync(m->d_keys, 0xFF, sizeof(unsigned long long) * m->tsize, h->stream));
    HIPCHK(h, hipMemsetAsync(m->d_epoch, 0, sizeof(uint32_t) * m->tsize, h->stream));
    HIPCHK(h, hipMemsetAsync(m->d_owner, 0xFF, sizeof(uint32_t) * m->tsize, h->stream));
    HIPCHK(h, hipMemsetAsync(m->d_count, 0, sizeof(uint32_t), h->stream));
    HIPCHK(h, hipMemsetAsync(m->d_flags, 0, sizeof(uint32_t) * 4, h->stream));
    m->epoch = 0; m->n_host = 0;
    return M3DREG_OK;
}
}  // namespace

extern "C" {

int m3dmap_create(m3dreg_handle* h, float dedup_leaf, size_t capacity, m3dmap** out) {
    return m3d_guarded(h, "m3dmap_create", [&]() -> int {
    if (!h || !out || !(dedup_leaf > 0.f) || !std::isfinite(dedup_leaf) || capacity == 0 || capacity > 0x0FFFFFFFull)
        return fail(h, M3DREG_ERR_INVALID_ARG, "m3dmap_create: bad argument (at most 2^28 - 1 points)");
    HIPCHK(h, hipSetDevice(h->device));
    m3dmap* m = new m3dmap();
    m->h = h; m->leaf = dedup_leaf; m->capacity = capacity;
    m->tsize = 1024; m->tbits = 10;
    while (m->tsize < 2u * uint32_t(capacity)) { m->tsize <<= 1; m->tbits++; }   // one voxel per kept point: load factor <= 1/2
    hipError_t e = m3d_malloc((void**)&m->d_pts, sizeof(float4) * capacity);
    if (e == hipSuccess) e = m3d_malloc((void**)&m->d_count, sizeof(uint32_t));
    if (e == hipSuccess) e = m3d_malloc((void**)&m->d_flags, sizeof(uint32_t) * 4);
    if (e == hipSuccess) e = m3d_malloc((void**)&m->d_keys, sizeof(unsigned long long) * m->tsize);
    if (e == hipSuccess) e = m3d_malloc((void**)&m->d_epoch, sizeof(uint32_t) * m->tsize);
    if (e == hipSuccess) e = m3d_malloc((void**)&m->d_owner, sizeof(uint32_t) * m->tsize);
    if (e != hipSuccess) { m3dmap_destroy(m); return fail(h, M3DREG_ERR_HIP, "m3dmap_create", e); }
    int rc = map_clear_device(m);
    if (rc) { m3dmap_destroy(m); return rc; }
    *out = m;
    return M3DREG_OK;
    });
}

int m3dmap_destroy(m3dmap* m) {
    return m3d_guarded((m ? m->h : nullptr), "m3dmap_destroy", [&]() -> int {
    if (!m) return M3DREG_ERR_INVALID_ARG;
    hipSetDevice(m->h->device);
    hipStreamSynchronize(m->h->stream);
    for (void* p : { (void*)m->d_pts, (void*)m->d_count, (void*)m->d_flags, (void*)m->d_keys, (void*)m->d_epoch, (void*)m->d_owner, (void*)m->d_slot_of, (void*)m->d_blocks }) if (p) hipFree(p);
    delete m;
    return M3DREG_OK;
    });
}

int m3dmap_clear(m3dmap* m) { return m ? map_clear_device(m) : M3DREG_ERR_INVALID_ARG; }

int m3dmap_insert(m3dmap* m, const m3dreg_cloud* scan, const float T[16], size_t* n_added) {
    return m3d_guarded((m ? m->h : nullptr), "m3dmap_insert", [&]() -> int {
    if (!m || !scan || !T) return M3DREG_ERR_INVALID_ARG;
    m3dreg_handle* h = m->h;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = size_t(scan->n), nblocks = (n + 255) / 256;
    if (n > m->slot_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (m->d_slot_of) hipFree(m->d_slot_of);
        m->d_slot_of = nullptr; m->slot_cap = 0;
        HIPCHK(h, m3d_malloc((void**)&m->d_slot_of, sizeof(uint32_t) * (n + n / 4 + 16)));
        m->slot_cap = n + n / 4 + 16;
    }
    if (nblocks > m->blocks_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (m->d_blocks) hipFree(m->d_blocks);
        m->d_blocks = nullptr; m->blocks_cap = 0;
        HIPCHK(h, m3d_malloc((void**)&m->d_blocks, sizeof(uint32_t) * (nblocks + nblocks / 4 + 16)));
        m->blocks_cap = nblocks + nblocks / 4 + 16;
    }
    if (scan->owner && scan->owner != h && scan->ready) HIPCHK(h, hipStreamWaitEvent(h->stream, scan->ready->ev, 0));
    M3dMapArgs A{};
    A.src = scan->xyz; A.n = int(n);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) A.R[3 * r + c] = T[c * 4 + r]; A.t[r] = T[12 + r]; }
    A.inv_leaf = 1.0f / m->leaf;
    A.keys = m->d_keys; A.epoch = m->d_epoch; A.owner = m->d_owner; A.tsize = m->tsize; A.tshift = 64 - m->tbits;
    A.cur_epoch = ++m->epoch;
    A.slot_of = m->d_slot_of; A.block_counts = m->d_blocks; A.out = m->d_pts; A.count = m->d_count; A.capacity = uint32_t(m->capacity); A.flags = m->d_flags;
    HIPCHK(h, hipMemsetAsync(m->d_flags, 0, sizeof(uint32_t) * 4, h->stream));
    HIPCHK(h, m3d_launch_map_insert(h->stream, A));
    { int rcf = note_foreign_use(h, scan); if (rcf) return rcf; }
    uint32_t host[5];
    HIPCHK(h, hipMemcpyAsync(host, m->d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(host + 1, m->d_flags, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_added) *n_added = size_t(host[0]) - m->n_host;
    m->n_host = host[0];
    if (host[1] || host[4]) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dmap_insert: map capacity exceeded (points beyond it were dropped)");
    if (host[2]) return fail(h, M3DREG_ERR_GRID_TOO_LARGE, "m3dmap_insert: points beyond +-2^20 dedup voxels were dropped");
    return M3DREG_OK;
    });
}

int m3dmap_size(m3dmap* m, size_t* n) {
    return m3d_guarded((m ? m->h : nullptr), "m3dmap_size", [&]() -> int {
    if (!m || !n) return M3DREG_ERR_INVALID_ARG;
    *n = m->n_host;
    return M3DREG_OK;
    });
}

int m3dmap_as_cloud(m3dmap* m, m3dreg_cloud** out) {
    return m3d_guarded((m ? m->h : nullptr), "m3dmap_as_cloud", [&]() -> int {
    if (!m || !out) return M3DREG_ERR_INVALID_ARG;
    if (m->n_host == 0) return fail(m->h, M3DREG_ERR_EMPTY_CLOUD, "m3dmap_as_cloud: the map is empty");
    return m3dreg_cloud_create(m->h, m->d_pts, m->n_host, 16, 0, 4, 8, 1, out);   // bucketed in place: no PCIe transfer of the map
    });
}

int m3dmap_download(m3dmap* m, float* xyzw, size_t cap_points, size_t* n_out) {
    return m3d_guarded((m ? m->h : nullptr), "m3dmap_download", [&]() -> int {
    if (!m || !n_out) return M3DREG_ERR_INVALID_ARG;
    *n_out = m->n_host;
    const size_t k = m->n_host < cap_points ? m->n_host : cap_points;
    if (xyzw && k) {
        HIPCHK(m->h, hipMemcpyAsync(xyzw, m->d_pts, 16 * k, hipMemcpyDeviceToHost, m->h->stream));
        HIPCHK(m->h, hipStreamSynchronize(m->h->stream));
    }
    return M3DREG_OK;
    });
}

// ---- loop-closure candidate generation (second half; include/m3dreg.h) ---------------------------
}  // extern "C"

struct m3dloop {
    m3dreg_handle* h = nullptr;
    m3dloop_params P{};
    int W = 0;                             // words per signature
    uint32_t* d_sig = nullptr;             // [max_keyframes][W]
    float4* d_pos = nullptr;               // [max_keyframes] {t, bits(popcount)}
    uint32_t* d_ov = nullptr;              // [LOOP_ROWS][max_keyframes] scores of the rows of one launch
    uint2* d_out = nullptr;                // [LOOP_ROWS][top_k]
    std::vector<const m3dreg_cloud*> clouds;
    std::vector<float> poses;              // 16 per keyframe, column-major, as given
    std::vector<m3dreg_cloud_desc> payloads;
    std::vector<uint8_t> has_payload;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0; uint64_t last_bytes = 0;
};

namespace {
constexpr int LOOP_ROWS = 256;   // query rows per launch of the scoring pass (their score rows live in d_ov)

int loop_sign(m3dloop* l, int32_t k, const m3dreg_cloud* c, const float T[16]) {
    m3dreg_handle* h = l->h;
    if (c->owner && c->owner != h && c->ready) HIPCHK(h, hipStreamWaitEvent(h->stream, c->ready->ev, 0));
    M3dLoopSignArgs A{};
    A.src = c->xyz; A.n = c->n;
    for (int r = 0; r < 3; r++) { for (int cc = 0; cc < 3; cc++) A.R[3 * r + cc] = T[cc * 4 + r]; A.t[r] = T[12 + r]; }
    A.inv_leaf = 1.0f / l->P.sig_leaf;
    A.sig = l->d_sig + size_t(k) * size_t(l->W);
    A.log2_bits = l->P.sig_log2_bits;
    HIPCHK(h, m3d_launch_loop_sign(h->stream, A, l->d_pos + k));
    return note_foreign_use(h, c);
}

// inv(T_j) * T_i in double from the float poses, fixed operation order (oracle/m3d_loop_oracle.c: orc_loop_rel)
void loop_rel(const float* Tj, const float* Ti, float out[16]) {
    double Rj[3][3], Ri[3][3], d[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { Rj[r][c] = double(Tj[c * 4 + r]); Ri[r][c] = double(Ti[c * 4 + r]); } d[r] = double(Ti[12 + r]) - double(Tj[12 + r]); }
    for (int k = 0; k < 16; k++) out[k] = 0.0f;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) out[c * 4 + r] = float((Rj[0][r] * Ri[0][c] + Rj[1][r] * Ri[1][c]) + Rj[2][r] * Ri[2][c]);
        out[12 + r] = float((Rj[0][r] * d[0] + Rj[1][r] * d[1]) + Rj[2][r] * d[2]);
    }
    out[15] = 1.0f;
}
}  // namespace

extern "C" {

int m3dloop_default_params(m3dloop_params* p) {
    if (!p) return M3DREG_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->sig_leaf = 2.0f; p->sig_log2_bits = 16; p->radius = 10.0f; p->min_gap = 10; p->top_k = 2; p->min_overlap = 0.5f; p->max_keyframes = 4096;
    return M3DREG_OK;
}

int m3dloop_destroy(m3dloop* l) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_destroy", [&]() -> int {
    if (!l) return M3DREG_ERR_INVALID_ARG;
    hipSetDevice(l->h->device);
    hipStreamSynchronize(l->h->stream);
    for (void* p : { (void*)l->d_sig, (void*)l->d_pos, (void*)l->d_ov, (void*)l->d_out }) if (p) hipFree(p);
    if (l->ev0) hipEventDestroy(l->ev0);
    if (l->ev1) hipEventDestroy(l->ev1);
    delete l;
    return M3DREG_OK;
    });
}

int m3dloop_create(m3dreg_handle* h, const m3dloop_params* P, m3dloop** out) {
    return m3d_guarded(h, "m3dloop_create", [&]() -> int {
    if (!h || !P || !out) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dloop_create: bad argument");
    *out = nullptr;
    if (!(P->sig_leaf > 0.f) || !std::isfinite(P->sig_leaf) || P->sig_log2_bits < 10 || P->sig_log2_bits > 18 || !(P->radius >= 0.f) || !std::isfinite(P->radius) ||
        P->min_gap < 1 || P->top_k < 1 || P->top_k > 16 || !(P->min_overlap >= 0.f && P->min_overlap <= 1.f) || P->max_keyframes < 1 || P->max_keyframes > (1 << 16) || P->reserved != 0)   // (65 536 keyframes: 512 MB of 8 KB signatures + 64 MB of score rows)
        return fail(h, M3DREG_ERR_INVALID_ARG, "m3dloop_create: parameter out of range (see m3dloop_params)");
    HIPCHK(h, hipSetDevice(h->device));
    alloc_point();
    m3dloop* l = new m3dloop();
    l->h = h; l->P = *P; l->W = 1 << (P->sig_log2_bits - 5);
    const size_t cap = size_t(P->max_keyframes);
    hipError_t e = m3d_malloc((void**)&l->d_sig, sizeof(uint32_t) * cap * size_t(l->W));
    if (e == hipSuccess) e = m3d_malloc((void**)&l->d_pos, sizeof(float4) * cap);
    if (e == hipSuccess) e = m3d_malloc((void**)&l->d_ov, sizeof(uint32_t) * cap * size_t(LOOP_ROWS));
    if (e == hipSuccess) e = m3d_malloc((void**)&l->d_out, sizeof(uint2) * size_t(LOOP_ROWS) * size_t(P->top_k));
    if (e == hipSuccess) e = hipEventCreate(&l->ev0);
    if (e == hipSuccess) e = hipEventCreate(&l->ev1);
    if (e != hipSuccess) { m3dloop_destroy(l); return fail(h, M3DREG_ERR_HIP, "m3dloop_create", e); }
    *out = l;
    return M3DREG_OK;
    });
}

int m3dloop_clear(m3dloop* l) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_clear", [&]() -> int {
    if (!l) return M3DREG_ERR_INVALID_ARG;
    HIPCHK(l->h, hipSetDevice(l->h->device));
    HIPCHK(l->h, hipStreamSynchronize(l->h->stream));   // (the keyframes' clouds may be released after this call)
    l->clouds.clear(); l->poses.clear(); l->payloads.clear(); l->has_payload.clear();
    return M3DREG_OK;
    });
}

int m3dloop_size(m3dloop* l, size_t* n) {
    if (!l || !n) return M3DREG_ERR_INVALID_ARG;
    *n = l->clouds.size();
    return M3DREG_OK;
}

int m3dloop_add_keyframe(m3dloop* l, const m3dreg_cloud* cloud, const float T[16], const m3dreg_cloud_desc* payload, int32_t* index) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_add_keyframe", [&]() -> int {
    if (!l || !cloud || !T) return M3DREG_ERR_INVALID_ARG;
    m3dreg_handle* h = l->h;
    if (l->clouds.size() >= size_t(l->P.max_keyframes)) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dloop_add_keyframe: max_keyframes reached");
    if (cloud->owner && cloud->owner->device != h->device) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dloop_add_keyframe: the cloud lives on another device");
    for (int k = 0; k < 16; k++) if (!std::isfinite(T[k])) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dloop_add_keyframe: non-finite pose");
    HIPCHK(h, hipSetDevice(h->device));
    alloc_point();
    const int32_t k = int32_t(l->clouds.size());
    l->clouds.reserve(size_t(k) + 1); l->poses.reserve(16 * (size_t(k) + 1)); l->payloads.reserve(size_t(k) + 1); l->has_payload.reserve(size_t(k) + 1);   // (nothing throws between the pushes below)
    int rc = loop_sign(l, k, cloud, T);
    if (rc) return rc;
    l->clouds.push_back(cloud);
    l->poses.insert(l->poses.end(), T, T + 16);
    l->payloads.push_back(payload ? *payload : m3dreg_cloud_desc{});
    l->has_payload.push_back(payload ? 1 : 0);
    if (index) *index = k;
    return M3DREG_OK;
    });
}

int m3dloop_update_pose(m3dloop* l, int32_t index, const float T[16]) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_update_pose", [&]() -> int {
    if (!l || !T || index < 0 || size_t(index) >= l->clouds.size()) return M3DREG_ERR_INVALID_ARG;
    for (int k = 0; k < 16; k++) if (!std::isfinite(T[k])) return fail(l->h, M3DREG_ERR_INVALID_ARG, "m3dloop_update_pose: non-finite pose");
    HIPCHK(l->h, hipSetDevice(l->h->device));
    int rc = loop_sign(l, index, l->clouds[size_t(index)], T);
    if (rc) return rc;
    std::copy(T, T + 16, l->poses.begin() + 16 * size_t(index));
    return M3DREG_OK;
    });
}

int m3dloop_signature(m3dloop* l, int32_t index, uint32_t* words, uint32_t* pop) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_signature", [&]() -> int {
    if (!l || index < 0 || size_t(index) >= l->clouds.size()) return M3DREG_ERR_INVALID_ARG;
    m3dreg_handle* h = l->h;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (words) HIPCHK(h, hipMemcpy(words, l->d_sig + size_t(index) * size_t(l->W), sizeof(uint32_t) * size_t(l->W), hipMemcpyDeviceToHost));
    if (pop) { float4 p; HIPCHK(h, hipMemcpy(&p, l->d_pos + index, sizeof(p), hipMemcpyDeviceToHost)); memcpy(pop, &p.w, 4); }
    return M3DREG_OK;
    });
}

int m3dloop_candidates(m3dloop* l, int32_t first, int32_t count, m3dloop_candidate* out, size_t cap, size_t* n_out) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_candidates", [&]() -> int {
    if (!l || !n_out || (cap && !out)) return M3DREG_ERR_INVALID_ARG;
    m3dreg_handle* h = l->h;
    *n_out = 0;
    const int32_t n = int32_t(l->clouds.size());
    if (first < 0 || first > n) return fail(h, M3DREG_ERR_INVALID_ARG, "m3dloop_candidates: first row out of range");
    const int32_t last = (count < 0 || first + count > n) ? n : first + count;
    l->last_ms = 0.0; l->last_bytes = 0;
    if (last <= first) return M3DREG_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const int K = l->P.top_k;
    const uint32_t thr_q16 = uint32_t(lrintf(l->P.min_overlap * 65536.0f));
    std::vector<uint2> res(size_t(last - first) * size_t(K));
    std::vector<float4> pos(static_cast<size_t>(n), make_float4(0.f, 0.f, 0.f, 0.f));
    HIPCHK(h, hipEventRecord(l->ev0, h->stream));
    for (int32_t r0 = first; r0 < last; r0 += LOOP_ROWS) {
        M3dLoopScoreArgs A{};
        A.sig = l->d_sig; A.pos = l->d_pos; A.W = l->W; A.row0 = r0; A.n_rows = std::min<int32_t>(LOOP_ROWS, last - r0);
        A.min_gap = l->P.min_gap; A.r2 = l->P.radius * l->P.radius; A.ov = l->d_ov; A.ov_stride = l->P.max_keyframes;
        HIPCHK(h, m3d_launch_loop_score(h->stream, A, l->d_out, K, thr_q16));
        HIPCHK(h, hipMemcpyAsync(res.data() + size_t(r0 - first) * size_t(K), l->d_out, sizeof(uint2) * size_t(A.n_rows) * size_t(K), hipMemcpyDeviceToHost, h->stream));
        // what the launch pair has to move: every signature it touches once, every score written and read once
        const long long jmax = std::max<long long>(0, (long long)(r0 + A.n_rows - 1) - A.min_gap + 1);
        long long cells = 0;
        for (int32_t i = r0; i < r0 + A.n_rows; i++) cells += std::max<long long>(0, (long long)i - A.min_gap + 1);
        l->last_bytes += uint64_t(jmax + A.n_rows) * uint64_t(l->W) * 4ull + uint64_t(cells) * 8ull;
        if (r0 + LOOP_ROWS < last) HIPCHK(h, hipStreamSynchronize(h->stream));   // (d_out / d_ov are re-used by the next chunk)
    }
    HIPCHK(h, hipEventRecord(l->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(pos.data(), l->d_pos, sizeof(float4) * size_t(n), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, l->ev0, l->ev1) == hipSuccess) l->last_ms = double(ms); }
    size_t found = 0;
    for (int32_t i = first; i < last; i++) {
        for (int k = 0; k < K; k++) {
            const uint2 e = res[size_t(i - first) * size_t(K) + size_t(k)];
            if (e.x == 0xFFFFFFFFu) break;
            if (found < cap) {
                m3dloop_candidate& c = out[found];
                const int32_t j = int32_t(e.x);
                c.source = i; c.target = j; c.overlap = e.y;
                memcpy(&c.pop_source, &pos[size_t(i)].w, 4); memcpy(&c.pop_target, &pos[size_t(j)].w, 4);
                const float dx = pos[size_t(i)].x - pos[size_t(j)].x, dy = pos[size_t(i)].y - pos[size_t(j)].y, dz = pos[size_t(i)].z - pos[size_t(j)].z;
                c.dist2 = std::fmaf(dz, dz, std::fmaf(dy, dy, dx * dx));
                loop_rel(&l->poses[16 * size_t(j)], &l->poses[16 * size_t(i)], c.init_T);
            }
            found++;
        }
    }
    *n_out = found;
    return M3DREG_OK;
    });
}

int m3dloop_last_profile(m3dloop* l, double* ms, uint64_t* bytes) {
    if (!l) return M3DREG_ERR_INVALID_ARG;
    if (ms) *ms = l->last_ms;
    if (bytes) *bytes = l->last_bytes;
    return M3DREG_OK;
}

int m3dloop_make_pairs(m3dloop* l, const m3dloop_candidate* cands, size_t n, m3dreg_pair* out) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_make_pairs", [&]() -> int {
    if (!l || (n && (!cands || !out))) return M3DREG_ERR_INVALID_ARG;
    const int32_t nk = int32_t(l->clouds.size());
    for (size_t i = 0; i < n; i++) {
        if (cands[i].source < 0 || cands[i].source >= nk || cands[i].target < 0 || cands[i].target >= nk) return fail(l->h, M3DREG_ERR_INVALID_ARG, "m3dloop_make_pairs: keyframe index out of range");
        out[i].source = l->clouds[size_t(cands[i].source)];
        out[i].target = l->clouds[size_t(cands[i].target)];
        memcpy(out[i].init_T, cands[i].init_T, sizeof(float) * 16);
    }
    return M3DREG_OK;
    });
}

int m3dloop_make_pair_descs(m3dloop* l, const m3dloop_candidate* cands, size_t n, m3dreg_pair_desc* out) {
    return m3d_guarded((l ? l->h : nullptr), "m3dloop_make_pair_descs", [&]() -> int {
    if (!l || (n && (!cands || !out))) return M3DREG_ERR_INVALID_ARG;
    const int32_t nk = int32_t(l->clouds.size());
    for (size_t i = 0; i < n; i++) {
        const int32_t s = cands[i].source, t = cands[i].target;
        if (s < 0 || s >= nk || t < 0 || t >= nk) return fail(l->h, M3DREG_ERR_INVALID_ARG, "m3dloop_make_pair_descs: keyframe index out of range");
        if (!l->has_payload[size_t(s)] || !l->has_payload[size_t(t)]) return fail(l->h, M3DREG_ERR_INVALID_ARG, "m3dloop_make_pair_descs: a keyframe was added without its payload descriptor");
        memset(&out[i], 0, sizeof(out[i]));
        out[i].source = l->payloads[size_t(s)]; out[i].source.source_only = 1;
        out[i].target = l->payloads[size_t(t)]; out[i].target.source_only = 0;
        memcpy(out[i].init_T, cands[i].init_T, sizeof(float) * 16);
        out[i].target_group = t + 1;   // every pair against keyframe t: one device, one upload, one bucketing
    }
    return M3DREG_OK;
    });
}

int m3dloop_gate(const m3dloop_candidate* cands, const m3dreg_stats* stats, size_t n, int64_t min_corr, double max_rms, uint8_t* accept) {
    if (n && (!stats || !accept)) return M3DREG_ERR_INVALID_ARG;
    (void)cands;
    for (size_t i = 0; i < n; i++)
        accept[i] = ((stats[i].status == M3DREG_CONVERGED || stats[i].status == M3DREG_MAX_ITERATIONS) && stats[i].n_corr >= min_corr && stats[i].rms <= max_rms) ? 1 : 0;
    return M3DREG_OK;
}

// ---- measurement ----------------------------------------------------------------------------------------
int m3dreg_profile_enable(m3dreg_handle* h, int on) {
    return m3d_guarded(h, "m3dreg_profile_enable", [&]() -> int {
    if (!h) return M3DREG_ERR_INVALID_ARG;
    h->profiling = on != 0;
    h->prof_every = on > 1 ? on : 1;
    return M3DREG_OK;
    });
}

int m3dreg_profile_batches(m3dreg_handle* h, int every) {
    return m3d_guarded(h, "m3dreg_profile_batches", [&]() -> int {
    if (!h || every < 1) return M3DREG_ERR_INVALID_ARG;
    h->prof_batch_every = every;
    return M3DREG_OK;
    });
}

int m3dreg_profile_read(m3dreg_handle* h, int what, uint64_t* n_launches, double* total_ms, int reset) {
    return m3d_guarded(h, "m3dreg_profile_read", [&]() -> int {
    if (!h || what < 0 || what > 4) return M3DREG_ERR_INVALID_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drain_events(h);
    if (n_launches) *n_launches = h->prof_launches[what];
    if (total_ms) *total_ms = h->prof_ms[what];
    if (reset) { h->prof_launches[what] = 0; h->prof_ms[what] = 0.0; }
    return M3DREG_OK;
    });
}

// ---- introspection ------------------------------------------------------------------------------------
int m3dreg_cloud_levels(const m3dreg_cloud* c) { return c ? c->n_levels : M3DREG_ERR_INVALID_ARG; }

int m3dreg_cloud_grid_info(m3dreg_handle* h, const m3dreg_cloud* c, int level, m3dreg_grid_info* out) {
    return m3d_guarded(h, "m3dreg_cloud_grid_info", [&]() -> int {
    if (!h || !c || !out || level < 0 || level >= c->n_levels) return fail(h, M3DREG_ERR_INVALID_ARG, "grid_info: bad argument");
    if (c->source_only && level < c->n_levels - 1) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "grid_info: the coarser levels of a source-only cloud are not built");
    { int rc = fetch_meta(h, const_cast<m3dreg_cloud*>(c)); if (rc) return rc; }
    const DevLevel& L = c->lv[level];
    memset(out, 0, sizeof(*out));
    out->n = c->n; out->n_valid = c->n_valid;
    out->n_cells = int32_t(L.n_cells_host);
    for (int a = 0; a < 3; a++) {
        out->dims[a] = L.grid.dims[a]; out->bits[a] = L.bits[a]; out->mn[a] = L.grid.mn[a]; out->mx[a] = L.mx[a]; out->center[a] = L.grid.center[a];
    }
    out->leaf = L.grid.leaf; out->inv_leaf = L.grid.inv_leaf; out->lbound = L.lbound; out->has_normals = c->has_normals ? 1 : 0;
    return M3DREG_OK;
    });
}

int m3dreg_cloud_density(m3dreg_handle* h, const m3dreg_cloud* c, int level, double* out) {
    return m3d_guarded(h, "m3dreg_cloud_density", [&]() -> int {
    if (!h || !c || !out || level < 0 || level >= c->n_levels) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_density: bad argument");
    if (c->source_only && level < c->n_levels - 1) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "cloud_density: the coarser levels of a source-only cloud are not built");
    { int rc = fetch_meta(h, const_cast<m3dreg_cloud*>(c)); if (rc) return rc; }
    *out = c->n_valid > 0 ? double(c->lv[level].sumsq_host) / double(c->n_valid) : 0.0;
    return M3DREG_OK;
    });
}

int m3dreg_debug_cloud_raw(m3dreg_handle* h, const m3dreg_cloud* c, int level, int what, void* out, size_t cap, size_t* bytes) {
    return m3d_guarded(h, "m3dreg_debug_cloud_raw", [&]() -> int {
    if (!h || !c || !bytes || level < 0 || level >= c->n_levels) return fail(h, M3DREG_ERR_INVALID_ARG, "debug_cloud_raw: bad argument");
    { int rc = fetch_meta(h, const_cast<m3dreg_cloud*>(c)); if (rc) return rc; }
    const DevLevel& L = c->lv[level];
    const size_t nt = size_t(m3d_tiles_of(c->n)), ni = nt + size_t(m3d_tile_pool(int(nt)));
    const void* p = nullptr; size_t sz = 0;
    switch (what) {
    case 0: p = L.htab; sz = sizeof(M3dBucket) * (size_t(L.grid.hmask) + 1); break;
    case 1: p = L.thdr; sz = sizeof(M3dTileHdr) * nt; break;
    case 2: p = L.timg; sz = size_t(M3D_TILE_IMG_BYTES) * ni; break;
    case 3: p = L.timeta; sz = sizeof(M3dTileImgMeta) * ni; break;
    case 4: p = L.occ; sz = size_t(1) << (M3D_OCC_BITS - 3); break;
    case 5: p = L.dyn; sz = sizeof(M3dLevelMeta); break;
    case 6: p = L.order; sz = 4 * ((size_t(c->n) + 255) / 256); break;
    default: return fail(h, M3DREG_ERR_INVALID_ARG, "debug_cloud_raw: unknown structure");
    }
    if (!p) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "debug_cloud_raw: the cloud has no such structure at this level");
    *bytes = sz;
    if (out && cap >= sz) { HIPCHK(h, hipSetDevice(h->device)); HIPCHK(h, hipMemcpy(out, p, sz, hipMemcpyDeviceToHost)); }
    return M3DREG_OK;
    });
}

int m3dreg_cloud_export(m3dreg_handle* h, const m3dreg_cloud* c, int level, uint32_t* keys, uint32_t* sorted_keys, int32_t* perm,
                        float* sorted_xyz, float* normals) {
    return m3d_guarded(h, "m3dreg_cloud_export", [&]() -> int {
    if (!h || !c || level < 0 || level >= c->n_levels) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_export: bad argument");
    if (c->source_only && level < c->n_levels - 1) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "cloud_export: the coarser levels of a source-only cloud are not built");
    if (normals && !c->has_normals) return fail(h, M3DREG_ERR_INVALID_ARG, "cloud_export: cloud has no normals");
    { int rc = fetch_meta(h, const_cast<m3dreg_cloud*>(c)); if (rc) return rc; }
    const DevLevel& L = c->lv[level];
    const size_t n = size_t(c->n);
    HIPCHK(h, hipSetDevice(h->device));
    // A coarser level of a pyramid keeps its points in the finest level's order inside every voxel (M3dBuild::fine: what makes its chunk boxes
    // compact); the spec's order — stable, i.e. input order inside a voxel — is restored here, on the host: this is the introspection path.
    const bool reorder = c->n_levels > 1 && level < c->n_levels - 1 && (perm || sorted_xyz || normals);
    std::vector<uint32_t> sk_tmp; std::vector<int32_t> pm_tmp;
    uint32_t* sk_host = sorted_keys; int32_t* pm_host = perm;
    if (reorder) {
        if (!sk_host) { sk_tmp.resize(n); sk_host = sk_tmp.data(); }
        if (!pm_host) { pm_tmp.resize(n); pm_host = pm_tmp.data(); }
    }
    if (keys) HIPCHK(h, hipMemcpyAsync(keys, L.keys, 4 * n, hipMemcpyDeviceToHost, h->stream));
    if (sk_host) HIPCHK(h, hipMemcpyAsync(sk_host, L.skey, 4 * n, hipMemcpyDeviceToHost, h->stream));
    if (pm_host) HIPCHK(h, hipMemcpyAsync(pm_host, L.perm, 4 * n, hipMemcpyDeviceToHost, h->stream));
    float *dx = nullptr, *dn = nullptr;
    if (sorted_xyz || normals) {
        HIPCHK(h, m3d_malloc((void**)&dx, 12 * n));
        if (normals) { hipError_t e = m3d_malloc((void**)&dn, 12 * n); if (e != hipSuccess) { hipFree(dx); return fail(h, M3DREG_ERR_HIP, "hipMalloc", e); } }
        hipError_t e = m3d_launch_export_sorted(h->stream, L.pts, normals ? L.nrm : nullptr, int(n), dx, dn);
        if (e == hipSuccess && sorted_xyz) e = hipMemcpyAsync(sorted_xyz, dx, 12 * n, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && normals) e = hipMemcpyAsync(normals, dn, 12 * n, hipMemcpyDeviceToHost, h->stream);
        hipError_t e2 = hipStreamSynchronize(h->stream);
        hipFree(dx); if (dn) hipFree(dn);
        if (e != hipSuccess) return fail(h, M3DREG_ERR_HIP, "cloud_export", e);
        if (e2 != hipSuccess) return fail(h, M3DREG_ERR_HIP, "cloud_export", e2);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (reorder) {
        std::vector<uint32_t> ord(n);
        for (size_t i = 0; i < n; i++) ord[i] = uint32_t(i);
        for (size_t a = 0; a < n;) {   // every run of equal keys: by input index
            size_t b = a + 1;
            while (b < n && sk_host[b] == sk_host[a]) b++;
            std::sort(ord.begin() + a, ord.begin() + b, [&](uint32_t x, uint32_t y) { return pm_host[x] < pm_host[y]; });
            a = b;
        }
        std::vector<int32_t> p2(n);
        for (size_t i = 0; i < n; i++) p2[i] = pm_host[ord[i]];
        for (float* arr : { sorted_xyz, normals }) {
            if (!arr) continue;
            std::vector<float> t(arr, arr + 3 * n);
            for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) arr[3 * i + k] = t[3 * size_t(ord[i]) + k];
        }
        if (perm) memcpy(perm, p2.data(), 4 * n);
    }
    return M3DREG_OK;
    });
}

int m3dreg_debug_nn(m3dreg_handle* h, const m3dreg_cloud* target, int level, const float* queries_xyz, size_t nq, float max_corr_dist,
                    int32_t* out_idx, float* out_d2) {
    return m3d_guarded(h, "m3dreg_debug_nn", [&]() -> int {
    if (!h || !target || !queries_xyz || !out_idx || !out_d2 || level < 0 || level >= target->n_levels || nq == 0 || nq >= 0x7FFFFFFFull)
        return fail(h, M3DREG_ERR_INVALID_ARG, "debug_nn: bad argument");
    if (target->source_only) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "debug_nn: a source-only cloud has no bucket table");
    { int rc = fetch_meta(h, const_cast<m3dreg_cloud*>(target)); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    float* dq = nullptr; int32_t* di = nullptr; float* dd = nullptr;
    HIPCHK(h, m3d_malloc((void**)&dq, 12 * nq));
    hipError_t e = m3d_malloc((void**)&di, 4 * nq);
    if (e == hipSuccess) e = m3d_malloc((void**)&dd, 4 * nq);
    if (e == hipSuccess) e = hipMemcpyAsync(dq, queries_xyz, 12 * nq, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = m3d_launch_debug_nn(h->stream, level_dev(target->lv[level]), dq, int(nq), max_corr_dist * max_corr_dist, di, dd);
    if (e == hipSuccess) e = hipMemcpyAsync(out_idx, di, 4 * nq, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_d2, dd, 4 * nq, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    hipFree(dq); if (di) hipFree(di); if (dd) hipFree(dd);
    if (e != hipSuccess) return fail(h, M3DREG_ERR_HIP, "debug_nn", e);
    if (e2 != hipSuccess) return fail(h, M3DREG_ERR_HIP, "debug_nn", e2);
    return M3DREG_OK;
    });
}

int m3dreg_debug_candidates(m3dreg_handle* h, const m3dreg_cloud* target, int level, const float* queries_xyz, size_t nq, int32_t* out_count) {
    return m3d_guarded(h, "m3dreg_debug_candidates", [&]() -> int {
    if (!h || !target || !queries_xyz || !out_count || level < 0 || level >= target->n_levels || nq == 0 || nq >= 0x7FFFFFFFull)
        return fail(h, M3DREG_ERR_INVALID_ARG, "debug_candidates: bad argument");
    if (target->source_only) return fail(h, M3DREG_ERR_LEVEL_MISMATCH, "debug_candidates: a source-only cloud has no bucket table");
    { int rc = fetch_meta(h, const_cast<m3dreg_cloud*>(target)); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    float* dq = nullptr; int32_t* di = nullptr;
    HIPCHK(h, m3d_malloc((void**)&dq, 12 * nq));
    hipError_t e = m3d_malloc((void**)&di, 4 * nq);
    if (e == hipSuccess) e = hipMemcpyAsync(dq, queries_xyz, 12 * nq, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = m3d_launch_debug_candidates(h->stream, level_dev(target->lv[level]), dq, int(nq), di);
    if (e == hipSuccess) e = hipMemcpyAsync(out_count, di, 4 * nq, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    hipFree(dq); if (di) hipFree(di);
    if (e != hipSuccess) return fail(h, M3DREG_ERR_HIP, "debug_candidates", e);
    if (e2 != hipSuccess) return fail(h, M3DREG_ERR_HIP, "debug_candidates", e2);
    return M3DREG_OK;
    });
}

int m3dreg_debug_accumulate(m3dreg_handle* h, const m3dreg_cloud* source, const m3dreg_cloud* target, int level, const float T[16],
                            int64_t sums[M3DREG_NSUMS], int32_t exps[6]) {
    return m3d_guarded(h, "m3dreg_debug_accumulate", [&]() -> int {
    if (!h || !source || !target || !T || !sums || !exps || level < 0 || level >= h->params.n_levels)
        return fail(h, M3DREG_ERR_INVALID_ARG, "debug_accumulate: bad argument");
    if (h->pending_pairs) return fail(h, M3DREG_ERR_INVALID_ARG, "debug_accumulate: a batch is pending on this handle (it uses the same job / state slots: call m3dreg_batch_wait first)");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_batch(h, 1);
    if (rc) return rc;
    m3dreg_pair p;
    p.source = source; p.target = target;
    memcpy(p.init_T, T, sizeof(float) * 16);
    int max_n_src = 0, max_n_tgt = 0;
    if ((rc = build_jobs(h, &p, 1, max_n_src, max_n_tgt))) return rc;
    if ((rc = ensure_match(h, 1, max_n_src, max_n_tgt))) return rc;
    h->h_jobs[size_t(level) * h->cap_pairs].prev_pts = nullptr;   // (no coarser level ran before this launch: the match array holds nothing to seed from)
    const M3dJob* hj = &h->h_jobs[size_t(level) * h->cap_pairs];
    HIPCHK(h, hipMemcpyAsync(h->d_jobs, hj, sizeof(M3dJob), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_states, h->h_states, sizeof(M3dPairState), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, m3d_launch_patch_jobs(h->stream, h->d_jobs, 1, int(h->cap_pairs), 1));
    HIPCHK(h, m3d_launch_accumulate_only(h->stream, h->d_jobs, 1, max_n_src, h->params.metric, nn_work(h)));
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
    // the exponents k_patch_jobs derived on the device, restated from the target's meta (same shared function)
    if ((rc = fetch_meta(h, const_cast<m3dreg_cloud*>(target)))) return rc;
    float S[6];
    m3d_fixed_exps(target->lv[level].lbound, h->params.max_corr_dist[level], exps, S);
    return M3DREG_OK;
    });
}

int m3dreg_debug_counters(m3dreg_handle* h, uint64_t out[2]) {
    return m3d_guarded(h, "m3dreg_debug_counters", [&]() -> int {
    if (!h || !out || !h->h_states) return M3DREG_ERR_INVALID_ARG;
    out[0] = h->h_states[0].ctr[0]; out[1] = h->h_states[0].ctr[1];
    return M3DREG_OK;
    });
}

int m3dreg_debug_trace(m3dreg_handle* h, double* poses, size_t cap, size_t* n_out) {
    return m3d_guarded(h, "m3dreg_debug_trace", [&]() -> int {
    if (!h || !n_out) return M3DREG_ERR_INVALID_ARG;
    if (h->pending_pairs) return fail(h, M3DREG_ERR_INVALID_ARG, "debug_trace: a batch is pending on this handle (call m3dreg_batch_wait first)");
    size_t k = h->last_trace_n < cap ? h->last_trace_n : cap;
    if (poses && k) {
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipMemcpyAsync(h->h_trace, h->d_trace, sizeof(double) * 16 * k, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        memcpy(poses, h->h_trace, sizeof(double) * 16 * k);
    }
    *n_out = h->last_trace_n;
    return M3DREG_OK;
    });
}

#ifdef M3D_CHECKED
}  // extern "C"
extern "C" hipError_t m3d_chk_read_icp(unsigned int* out, int reset);
extern "C" hipError_t m3d_chk_read_bucket(unsigned int* out, int reset);
extern "C" {
#endif
// The diagnosis build's report (libm3dreg_checked.so: -DM3D_CHECKED): out[0..3] = {offences, site, index, bound} of the iteration kernels (icp.hip), out[4..7] of the
// bucketing pipeline (bucket.hip); the first offence of each is kept. The shipped library has no checks compiled in and answers M3DREG_ERR_INVALID_ARG.
int m3dreg_debug_checks(m3dreg_handle* h, uint32_t out[8], int reset) {
    return m3d_guarded(h, "m3dreg_debug_checks", [&]() -> int {
    if (!h || !out) return M3DREG_ERR_INVALID_ARG;
#ifdef M3D_CHECKED
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, m3d_chk_read_icp(out, reset));
    HIPCHK(h, m3d_chk_read_bucket(out + 4, reset));
    return M3DREG_OK;
#else
    (void)reset;
    for (int i = 0; i < 8; i++) out[i] = 0u;
    return fail(h, M3DREG_ERR_INVALID_ARG, "m3dreg_debug_checks: this is not the checked build (make checked: libm3dreg_checked.so)");
#endif
    });
}
int m3dreg_debug_fail_alloc(int nth) { g_fail_alloc.store(nth > 0 ? nth : 0); return M3DREG_OK; }
int m3dreg_debug_throw(int kind) {
    return m3d_guarded(nullptr, "m3dreg_debug_throw", [&]() -> int {
        if (kind == 0) throw std::bad_alloc();
        throw 42;
    });
}

}  // extern "C"

// ---- one process, several devices (SURVEY.md §8 rows b / e) ------------------------------------------------------
// One HOST THREAD per listed device ("one host thread + one HIP stream per device"): the worker owns its handle, uploads,
// buckets and registers its shard and collects its results, so the devices' uploads and enqueues run side by side instead of one
// after the other on the caller's thread. Host payloads reach the device through PINNED memory: a payload that already is pinned
// (m3dreg_host_alloc / m3dreg_host_register, or any hipHostMalloc'ed / registered range) is handed to the copy engine as it is — a
// truly asynchronous hipMemcpyAsync —, a pageable one is first copied into the worker's pinned staging block (a pageable
// hipMemcpyAsync is synchronous and staged by the runtime anyway; here the staging of one device overlaps the others').
struct MultiWorker {
    m3dreg_handle* h = nullptr;
    int device = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, quit = false, busy = false;
    void* pinned = nullptr; size_t pinned_bytes = 0;   // staging for pageable payloads (grown on demand, reused by every call)
    void run() {
        hipSetDevice(device);
        for (;;) {
            std::function<void()> j;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return has_job || quit; }); if (quit && !has_job) return; j = std::move(job); has_job = false; }
            j();   // (never throws: the job catches everything)
            { std::lock_guard<std::mutex> lk(mu); busy = false; }
            cv.notify_all();
        }
    }
    void post(std::function<void()> j) { { std::lock_guard<std::mutex> lk(mu); job = std::move(j); has_job = true; busy = true; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !busy; }); }
};
struct m3dreg_multi {
    std::vector<std::unique_ptr<MultiWorker>> workers;   // one per entry of `devices` (a device may appear more than once: several streams on it)
    std::vector<int> devices;
    std::string err;
    size_t last_clouds = 0;   // clouds uploaded and bucketed by the last m3dreg_multi_align (m3dreg_debug_multi_clouds)
};

namespace {
// Longest-processing-time-first with a capacity per device (what mandala_mapping_amd/sharding.py lpt_assign does for the ranks of a
// torchrun job): heaviest pair first, onto the least loaded device that still has room; ties to the lower index.
// A UNIT is what must stay together: one pair, or all pairs of a target group (size[u] pairs). A unit that no device has room for under the
// capacity (a group larger than ceil(n_pairs / n_devices)) goes to the least loaded device regardless.
void lpt_assign(const std::vector<double>& cost, const std::vector<size_t>& size, int n_dev, size_t capacity, std::vector<int>& dev_of) {
    std::vector<size_t> order(cost.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
    std::vector<double> load(size_t(n_dev), 0.0);
    std::vector<size_t> count(size_t(n_dev), 0);
    dev_of.assign(cost.size(), 0);
    for (size_t i : order) {
        int best = -1, any = 0;
        for (int d = 0; d < n_dev; d++) {
            if (load[size_t(d)] < load[size_t(any)]) any = d;
            if (count[size_t(d)] + size[i] <= capacity && (best < 0 || load[size_t(d)] < load[size_t(best)])) best = d;
        }
        if (best < 0) best = any;
        dev_of[i] = best;
        load[size_t(best)] += cost[i];
        count[size_t(best)] += size[i];
    }
}
bool same_payload(const m3dreg_cloud_desc& a, const m3dreg_cloud_desc& b) {
    return a.data == b.data && a.n == b.n && a.point_step == b.point_step && a.off_x == b.off_x && a.off_y == b.off_y && a.off_z == b.off_z &&
           (a.data_is_device != 0) == (b.data_is_device != 0);
}
int mfail(m3dreg_multi* m, int code, const std::string& msg) { if (m) { try { m->err = msg; } catch (...) {} } return code; }

bool host_ptr_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (an ordinary malloc'ed pointer: "invalid value")
    return a.type == hipMemoryTypeHost;
}

// what one worker does with its shard; everything it allocates is released on every path out, and its handle is idle when it returns
struct ShardResult { int rc = M3DREG_OK; std::string msg; std::vector<float> T; std::vector<m3dreg_stats> st; size_t n_clouds = 0; };
void run_shard(MultiWorker* w, const m3dreg_pair_desc* pairs, const std::vector<size_t>& idx, ShardResult& R) {
    m3dreg_handle* h = w->h;
    std::vector<m3dreg_cloud*> clouds;
    bool touched = false;   // something was enqueued on the handle's stream that may still read host payloads / hold a pending batch
    auto fail_with = [&](int rc, const char* what) { R.rc = rc; try { R.msg = std::string("device ") + std::to_string(w->device) + ": " + what; } catch (...) {} };
    try {
        const size_t k = idx.size();
        // the shard's clouds: every pair's source, and every DISTINCT target — the pairs of a target group (m3dreg_pair_desc.target_group > 0) share
        // one cloud, uploaded and bucketed once; src_of / tgt_of: the cloud of pair j
        std::vector<m3dreg_cloud_desc> descs;
        std::vector<size_t> src_of(k), tgt_of(k);
        std::unordered_map<int32_t, size_t> group_cloud;   // target group -> its (one) cloud of this shard
        descs.reserve(2 * k);
        for (size_t j = 0; j < k; j++) {
            const m3dreg_pair_desc& P = pairs[idx[j]];
            m3dreg_cloud_desc d = P.source; d.source_only = 1;
            src_of[j] = descs.size(); descs.push_back(d);
            size_t t = size_t(-1);
            if (P.target_group > 0) { const auto f = group_cloud.find(P.target_group); if (f != group_cloud.end()) t = f->second; }
            if (t == size_t(-1)) {
                d = P.target; d.source_only = 0; t = descs.size(); descs.push_back(d);
                if (P.target_group > 0) group_cloud[P.target_group] = t;
            }
            tgt_of[j] = t;
        }
        const size_t nc = descs.size();
        // pageable host payloads go through this worker's pinned staging block
        size_t need = 0;
        std::vector<size_t> off(nc, size_t(-1));
        for (size_t i = 0; i < nc; i++) {
            const m3dreg_cloud_desc& d = descs[i];
            if (!d.data_is_device && d.data && d.n && !host_ptr_is_pinned(d.data)) { off[i] = need; need += (d.n * d.point_step + 255) & ~size_t(255); }
        }
        if (need > w->pinned_bytes) {
            if (w->pinned) { hipStreamSynchronize(h->stream); hipHostFree(w->pinned); w->pinned = nullptr; w->pinned_bytes = 0; }
            alloc_point();
            if (hipHostMalloc(&w->pinned, need + need / 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); w->pinned = nullptr; }
            else w->pinned_bytes = need + need / 4;
        }
        if (w->pinned && need <= w->pinned_bytes)
            for (size_t i = 0; i < nc; i++)
                if (off[i] != size_t(-1)) {
                    uint8_t* dst = static_cast<uint8_t*>(w->pinned) + off[i];
                    memcpy(dst, descs[i].data, descs[i].n * descs[i].point_step);
                    descs[i].data = dst;
                }   // (no pinned memory to be had: the payloads go as they are — a synchronous pageable copy, still correct)
        clouds.assign(nc, nullptr);
        R.n_clouds = nc;
        touched = true;
        int rc = m3dreg_cloud_create_batch_async(h, descs.data(), descs.size(), clouds.data());
        if (rc != M3DREG_OK) fail_with(rc, h->err.c_str());
        else {
            std::vector<m3dreg_pair> pr(k);
            for (size_t j = 0; j < k; j++) {
                pr[j].source = clouds[src_of[j]]; pr[j].target = clouds[tgt_of[j]];
                memcpy(pr[j].init_T, pairs[idx[j]].init_T, sizeof(float) * 16);
            }
            R.T.assign(16 * k, 0.f); R.st.assign(k, m3dreg_stats{});
            rc = m3dreg_align_batch_async(h, pr.data(), pr.size());
            if (rc == M3DREG_OK) rc = m3dreg_batch_wait(h, R.T.data(), R.st.data());
            if (rc != M3DREG_OK) fail_with(rc, h->err.c_str());
        }
    } catch (const std::bad_alloc&) { fail_with(M3DREG_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    catch (...) { fail_with(M3DREG_ERR_HIP, "unexpected exception"); }
    // leave the handle idle and give everything back, whatever happened above (a throw in the middle left earlier devices
    // with a pending batch, leaked the shard's clouds and returned while copies could still read the caller's payloads)
    if (touched) {
        if (h->pending_pairs) { hipStreamSynchronize(h->stream); if (h->ev_used) drain_events(h); h->pending_pairs = 0; }
        else hipStreamSynchronize(h->stream);
    }
    for (m3dreg_cloud* c : clouds) if (c) free_cloud(h, c);
}
}  // namespace

extern "C" {

const char* m3dreg_multi_last_error(const m3dreg_multi* m) { return m ? m->err.c_str() : "null context"; }

int m3dreg_multi_destroy(m3dreg_multi* m) {
    return m3d_guarded(nullptr, "m3dreg_multi_destroy", [&]() -> int {
    if (!m) return M3DREG_ERR_INVALID_ARG;
    for (auto& w : m->workers) {
        if (w->th.joinable()) { { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; } w->cv.notify_all(); w->th.join(); }
        if (w->h) { hipSetDevice(w->device); if (w->pinned) { hipStreamSynchronize(w->h->stream); hipHostFree(w->pinned); } m3dreg_destroy(w->h); }
    }
    delete m;
    return M3DREG_OK;
    });
}

int m3dreg_multi_create(const m3dreg_params* params, const int* devices, int n_devices, m3dreg_multi** out) {
    return m3d_guarded(nullptr, "m3dreg_multi_create", [&]() -> int {
    if (!out) return M3DREG_ERR_INVALID_ARG;
    *out = nullptr;
    if (!params || !devices || n_devices < 1 || n_devices > 64) return M3DREG_ERR_INVALID_ARG;
    alloc_point();
    m3dreg_multi* m = new m3dreg_multi();
    try {
        for (int d = 0; d < n_devices; d++) {
            m->workers.emplace_back(new MultiWorker());
            MultiWorker* w = m->workers.back().get();
            w->device = devices[d];
            const int rc = m3dreg_create(params, devices[d], nullptr, &w->h);
            if (rc != M3DREG_OK) { m3dreg_multi_destroy(m); return rc; }
            m->devices.push_back(devices[d]);
            w->th = std::thread([w] { w->run(); });
        }
    } catch (...) { m3dreg_multi_destroy(m); throw; }
    *out = m;
    return M3DREG_OK;
    });
}

int m3dreg_multi_align(m3dreg_multi* m, const m3dreg_pair_desc* pairs, size_t n_pairs, float* out_T, m3dreg_stats* stats, int32_t* device_of_pair) {
    if (!m || !pairs || !out_T || n_pairs == 0 || n_pairs > 65535) return mfail(m, M3DREG_ERR_INVALID_ARG, "multi_align: bad argument");
    const int n_dev = int(m->workers.size());
    std::vector<std::vector<size_t>> idx;
    std::vector<ShardResult> res;
    int posted = 0;
    int rc = M3DREG_OK;
    std::string msg;
    try {
        alloc_point();
        // units of the assignment: a pair, or a whole target group (its target counted once: it is uploaded and bucketed once)
        std::vector<size_t> unit_of(n_pairs), usize;
        std::vector<double> cost;
        std::vector<size_t> ufirst;
        std::unordered_map<int32_t, size_t> unit_of_group;
        for (size_t i = 0; i < n_pairs; i++) {
            const int32_t gid = pairs[i].target_group;
            if (gid < 0) return mfail(m, M3DREG_ERR_INVALID_ARG, "multi_align: negative target_group");
            if (pairs[i].reserved != 0) return mfail(m, M3DREG_ERR_INVALID_ARG, "multi_align: m3dreg_pair_desc.reserved must be 0 (zero-initialise the struct: code written for ABI 4 leaves target_group and reserved undefined)");
            size_t u = size_t(-1);
            if (gid > 0) { const auto f = unit_of_group.find(gid); if (f != unit_of_group.end()) u = f->second; }
            if (u == size_t(-1)) {
                u = cost.size(); cost.push_back(double(pairs[i].target.n)); usize.push_back(0); ufirst.push_back(i);
                if (gid > 0) unit_of_group[gid] = u;
            }
            else if (!same_payload(pairs[ufirst[u]].target, pairs[i].target)) return mfail(m, M3DREG_ERR_INVALID_ARG, "multi_align: the pairs of a target_group must name the same target payload");
            unit_of[i] = u; cost[u] += double(pairs[i].source.n); usize[u]++;
        }
        std::vector<int> dev_of;
        lpt_assign(cost, usize, n_dev, (n_pairs + size_t(n_dev) - 1) / size_t(n_dev), dev_of);
        idx.assign(size_t(n_dev), std::vector<size_t>());
        res.assign(size_t(n_dev), ShardResult());
        for (size_t i = 0; i < n_pairs; i++) idx[size_t(dev_of[unit_of[i]])].push_back(i);
        // every device's worker takes its shard: uploads, bucketing, registrations and the wait for them run side by side
        for (int d = 0; d < n_dev; d++, posted++) {
            if (idx[size_t(d)].empty()) continue;
            MultiWorker* w = m->workers[size_t(d)].get();
            const std::vector<size_t>* ix = &idx[size_t(d)];
            ShardResult* R = &res[size_t(d)];
            w->post([w, pairs, ix, R] { run_shard(w, pairs, *ix, *R); });
        }
    } catch (const std::bad_alloc&) { rc = M3DREG_ERR_OUT_OF_MEMORY; }
    catch (...) { rc = M3DREG_ERR_HIP; }
    // whatever was posted is waited for — also on the error paths: the workers read the caller's payloads and write into res
    for (int d = 0; d < posted && d < n_dev; d++) m->workers[size_t(d)]->wait();
    if (rc != M3DREG_OK) return mfail(m, rc, rc == M3DREG_ERR_OUT_OF_MEMORY ? "multi_align: host allocation failed" : "multi_align: unexpected exception");
    m->last_clouds = 0;
    for (int d = 0; d < n_dev; d++) {
        const ShardResult& R = res[size_t(d)];
        m->last_clouds += R.n_clouds;
        if (R.rc != M3DREG_OK) { if (rc == M3DREG_OK) { rc = R.rc; msg = R.msg; } continue; }
        for (size_t k = 0; k < idx[size_t(d)].size(); k++) {
            const size_t i = idx[size_t(d)][k];
            memcpy(out_T + 16 * i, R.T.data() + 16 * k, sizeof(float) * 16);
            if (stats) stats[i] = R.st[k];
            if (device_of_pair) device_of_pair[i] = m->devices[size_t(d)];
        }
    }
    if (rc != M3DREG_OK) return mfail(m, rc, msg);
    return M3DREG_OK;
}

int m3dreg_debug_multi_clouds(const m3dreg_multi* m) { return m ? int(m->last_clouds) : M3DREG_ERR_INVALID_ARG; }

// Pinned host memory for callers that do not link HIP: a payload that lives in it reaches the device by a truly asynchronous copy
// (every cloud_create* entry point and m3dreg_multi_align recognise pinned ranges; pageable ones work too, at the price of a
// synchronous, staged copy). m3dreg_host_register pins an existing allocation in place until m3dreg_host_unregister.
int m3dreg_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return M3DREG_ERR_INVALID_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return M3DREG_ERR_HIP; }
    return M3DREG_OK;
}
int m3dreg_host_free(void* p) { if (!p) return M3DREG_ERR_INVALID_ARG; return hipHostFree(p) == hipSuccess ? M3DREG_OK : M3DREG_ERR_HIP; }
int m3dreg_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) return M3DREG_ERR_INVALID_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return M3DREG_ERR_HIP; }
    return M3DREG_OK;
}
int m3dreg_host_unregister(void* p) { if (!p) return M3DREG_ERR_INVALID_ARG; return hipHostUnregister(p) == hipSuccess ? M3DREG_OK : M3DREG_ERR_HIP; }

}  // extern "C"
