// bucket.hip — SURVEY.md §8 rows a2 (decode), a3 (voxel keys), a4 (bucketing) and a9 (normals) as
// hand-written HIP for gfx950. No reference source exists for these stages (the reference's
// gpu_6dslam is an empty submodule); the nearest in-tree analogue is the pcl::VoxelGrid(0.1) use at
// /root/reference/m3d/m3d_calibration/src/m3d_calibration_twiddle.cpp:279-286. The normative
// arithmetic is DESIGN.md §Spec; oracle/m3d_oracle.c restates it on the CPU.
//
// Everything is BATCHED: one launch serves every cloud / every grid of a batch (blockIdx.y selects the
// descriptor), because a single 100k-point cloud cannot fill 256 CUs and ~50 tiny launches per cloud
// were costing more than the work itself (profiles/r01_v4_*). All of this is HBM-bound integer/byte
// work: coalesced 4-B/16-B streams, LDS histograms and wave64 ballots for the stable in-block ranks;
// no MFMA (nothing here is a contraction).
#include "m3d_kernels.h"
M3D_CHK_READER(m3d_chk_read_bucket)

// Workgroup -> (row, block) of the batched launches below (a row = one cloud or one grid of the batch; 1-D grids of rows * bpr workgroups).
// Consecutive workgroups of a row go to consecutive XCDs (the dispatcher deals workgroups round-robin by linear id). Round 4 measured the
// opposite — with a multiple of 8 rows, every row's workgroups on ONE XCD, so that a radix scatter's partly written lines or a finalize pass's
// 1.6 MB gather set meet in one L2 —: every kernel of the pipeline got SLOWER, the pure streams too (k_voxel_keys 15.7 -> 21.6 us, k_finalize_level
// 41 -> 77, k_tile_build 47 -> 74, k_rs_scatter 19.8 -> 23.9; bucketing 0.32 -> 0.49 ms per step, profiles/r04_xcd_rows.txt): eight XCDs walking
// eight far-apart address streams lose more in DRAM locality than the L2s gain.
struct M3dRB { int row, blk; };
// The batched launches below have one ROW per LEVEL build (round 5): a point-to-plane cloud's first build, the normal grid, carries geometry and a voxel table but is
// never keyed, sorted or tabled — as a row of these launches it was a third to a half of their workgroups, every one of which left at once (a dispatch slot each:
// ~0.4 ns, 50 000 of them per launch with 64 pairs in one chain). `rows` packs the batch's shape: levels per cloud | builds per cloud << 16.
__device__ __forceinline__ int m3d_row_build(int row, int rows) {
    const int lv = rows & 0xFFFF, gpc = rows >> 16;
    return (row / lv) * gpc + (gpc - lv) + (row % lv);
}
__device__ __forceinline__ M3dRB m3d_row_block(int rows, int bpr, int id = (int)blockIdx.x) {
    M3D_ENTRY_JITTER();
    M3dRB r;
    r.row = id / bpr; r.blk = id - r.row * bpr;
    (void)rows;
    return r;
}
// SLICED rows (round 5), for the kernels whose workgroups gather or scatter with spatial locality: a row is launched with 8 * ceil(bpr / 8) workgroups and
// XCD x (the dispatcher deals workgroups round-robin by linear id; a padded row starts at a multiple of 8) takes the x-th CONTIGUOUS eighth of the row's
// blocks. A cloud is still spread over all eight XCDs, but each XCD's L2 sees one compact stretch of the sorted order — one region of space, whose points
// also sit close together in the input (scan) order — instead of every eighth block of the whole cloud. blk >= bpr: a padding workgroup, nothing to do.
__device__ __forceinline__ M3dRB m3d_row_block_sliced(int bpr, int sliced, int id = (int)blockIdx.x) {
    M3D_ENTRY_JITTER();
    const int per = (bpr + 7) >> 3, w = per << 3;
    M3dRB r;
    r.row = id / w;
    const int k = id - r.row * w;
    r.blk = sliced ? (k & 7) * per + (k >> 3) : k;
    return r;
}
static inline int m3d_sliced_grid(int bpr) { return ((bpr + 7) >> 3) << 3; }
// (M3DREG_SLICED: bit 0 = k_finalize_level, 1 = k_rs_scatter, 2 = the tile workgroups of k_tiles_normals; default OFF. Measured, profiles/r05_sliced.txt: sliced,
// k_finalize_level fetches 50 MB per 16-cloud step instead of 109, k_rs_scatter writes 42 instead of 54, the tile build fetches 40 instead of 58 — and nothing gets
// faster: alone the kernels are latency-bound, with 64 pairs in one chain k_finalize_level takes 48 instead of 38 us per 8 pairs and the tile build 4 us more (eight XCDs
// walking eight far-apart stretches lose more in DRAM page locality than their L2s gain, as round 4 found for whole rows), the headline is within +-0.4 %.)
static inline int m3d_sliced_on() { static const int on = [] { const char* v = getenv("M3DREG_SLICED"); return v ? atoi(v) : 0; }(); return on; }

#define RS_THREADS 256
#define RS_ROUNDS 8
#define RS_TILE (RS_THREADS * RS_ROUNDS)
#define RS_WAVES (RS_THREADS / 64)

// ---- a2: PointCloud2 payload -> one float4 per point (input order) + exact AABB -------------------
__device__ __forceinline__ uint32_t ord_f32(float f) {   // order-preserving float -> uint map
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unord_f32(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    union { uint32_t u; float f; } c; c.u = u; return c.f;
}
float m3d_unord_f32(uint32_t u) { return unord_f32(u); }

// one x/y/z field of a general PointCloud2 point: bytes -> (FLOAT32 | FLOAT64 rounded to nearest float), any alignment
__device__ __forceinline__ float decode_field(const uint8_t* p, int is_f64, int bigendian) {
    const int nb = is_f64 ? 8 : 4;
    unsigned long long v = 0;
    for (int b = 0; b < nb; b++) v |= (unsigned long long)p[b] << (8 * (bigendian ? (nb - 1 - b) : b));
    return is_f64 ? (float)__longlong_as_double((long long)v) : __uint_as_float((uint32_t)v);
}

// aabb (zero-initialised): [0..2] = max of ~ordered (i.e. the minimum), [3..5] = max of ordered, [6] = finite points
__global__ __launch_bounds__(256) void k_decode_aabb(const M3dDecode* __restrict__ descs, int rows, int bpr) {
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dDecode D = descs[rb.row];
    const int n = D.n;
    uint32_t mn[3] = { 0u, 0u, 0u }, mx[3] = { 0u, 0u, 0u };
    uint32_t cnt = 0;
    for (int i = rb.blk * (int)blockDim.x + (int)threadIdx.x; i < n; i += bpr * (int)blockDim.x) {
        float px, py, pz;
        if (!D.generic) {
            const uint8_t* p = D.raw + (size_t)i * D.step;
            px = *reinterpret_cast<const float*>(p + D.ox);
            py = *reinterpret_cast<const float*>(p + D.oy);
            pz = *reinterpret_cast<const float*>(p + D.oz);
        } else {   // any field offset / FLOAT32 or FLOAT64 / either byte order / padded rows, on the device
            const uint8_t* p = D.raw + (size_t)(i / D.width) * (size_t)D.row_step + (size_t)(i % D.width) * (size_t)D.step;
            px = decode_field(p + D.ox, D.f64[0], D.bigendian);
            py = decode_field(p + D.oy, D.f64[1], D.bigendian);
            pz = decode_field(p + D.oz, D.f64[2], D.bigendian);
        }
        D.xyz[i] = make_float4(px, py, pz, 0.f);
        if (m3d_finite3(px, py, pz)) {
            const uint32_t a = ord_f32(px), b = ord_f32(py), c = ord_f32(pz);
            mn[0] = max(mn[0], ~a); mx[0] = max(mx[0], a);
            mn[1] = max(mn[1], ~b); mx[1] = max(mx[1], b);
            mn[2] = max(mn[2], ~c); mx[2] = max(mx[2], c);
            cnt++;
        }
    }
    // wave64 shuffle reduction -> LDS across the 4 waves -> one set of integer atomics per block
    // (exact and order-independent; per-wave atomics made this kernel contention-bound: 128 us vs 9 us)
    for (int o = 32; o > 0; o >>= 1) {
        for (int a = 0; a < 3; a++) {
            mn[a] = max(mn[a], (uint32_t)__shfl_down((int)mn[a], o));
            mx[a] = max(mx[a], (uint32_t)__shfl_down((int)mx[a], o));
        }
        cnt += __shfl_down((int)cnt, o);
    }
    __shared__ uint32_t red[4][7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { for (int a = 0; a < 3; a++) { red[wave][a] = mn[a]; red[wave][3 + a] = mx[a]; } red[wave][6] = cnt; }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        uint32_t v = red[0][a];
        for (int w = 1; w < 4; w++) v = (a < 6) ? max(v, red[w][a]) : v + red[w][a];
        if (a < 6) { if (v) atomicMax(&D.aabb[a], v); } else if (v) atomicAdd(&D.aabb[6], v);
    }
}

// ---- Spec §Grid on the device: one thread per CLOUD derives the geometry of all of its grids from the exact AABB -----------
// (k_decode_aabb's words), the number of radix passes each needs and the cloud's error state — what the host used to do behind
// a stream synchronisation. A cloud in error (no finite point, or a grid that needs more than 31 key bits) gets n = 0 in
// every build: the pipeline below does nothing for it, and k_patch_jobs ends any registration that names it.
__global__ void k_grid_params(M3dBuild* __restrict__ builds, int n_clouds, int gpc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clouds) return;
    M3dBuild* B0 = builds + (size_t)c * gpc;
    const uint32_t* ab = B0->aabb;
    const int32_t n_valid = (int32_t)ab[6];
    float mn[3], mx[3];
    int err = n_valid == 0 ? M3D_ERR_EMPTY_CLOUD : 0;
    for (int a = 0; a < 3; a++) {   // unord_f32, spelled with XOR masks: the select-then-pun form crashes the gfx950 instruction selector of ROCm 7.2 here
        const uint32_t lo = ~ab[a], hi = ab[3 + a];
        mn[a] = __uint_as_float(lo ^ (((lo >> 31) - 1u) | 0x80000000u));
        mx[a] = __uint_as_float(hi ^ (((hi >> 31) - 1u) | 0x80000000u));
    }
    for (int gi = 0; gi < gpc && !err; gi++) {
        M3dBuild& B = B0[gi];
        M3dLevelMeta* M = reinterpret_cast<M3dLevelMeta*>(B.dyn);
        M3dGrid g; int32_t bits[3]; float lbound;
        err = m3d_make_grid(mn, mx, B.grid.leaf, n_valid, g, bits, lbound);
        if (err) break;
        B.grid = g;
        const int kb = bits[0] + bits[1] + bits[2] + 3;
        B.sort_passes = (B.n == 0) ? 0 : ((n_valid != B.n) ? 4 : (kb + 7) / 8);   // the 0xFFFFFFFF keys of non-finite points must end up last; n = 0: a build the host switched off
        M->g = g; M->lbound = lbound;
        for (int a = 0; a < 3; a++) { M->mx[a] = mx[a]; M->bits[a] = bits[a]; }
        if (B.nkeys) {   // a normal grid: its voxel table is allocated for the worst case (as many occupied voxels as points: a power of two > 1.25 n); it cannot hold more voxels
            // than the grid has cells either — a 2 M-point map in a 40 x 30 x 6 m room has 112 k normal-grid cells: 262 144 slots instead of 4 194 304 (436 MB touched
            // sparsely; the keys' clear alone was 32 MB per bucketing). Every user of the table reads ncap / nshift from this descriptor on the device.
            const unsigned long long cells = (unsigned long long)g.dims[0] * (unsigned long long)g.dims[1] * (unsigned long long)g.dims[2];
            const unsigned long long bound = cells < (unsigned long long)n_valid ? cells : (unsigned long long)n_valid;
            uint32_t c = 16u; int lg = 4;
            while ((unsigned long long)c < bound + bound / 4ull + 16ull && c < B.ncap) { c <<= 1; lg++; }
            if (c < B.ncap) { B.ncap = c; B.nshift = 32 - lg; }
        }
    }
    for (int gi = 0; gi < gpc; gi++) {
        M3dBuild& B = B0[gi];
        M3dLevelMeta* M = reinterpret_cast<M3dLevelMeta*>(B.dyn);
        M->err = err;
        if (err) {   // a cloud in error: every word of the meta is still defined (k_patch_jobs and fetch_meta copy it as it stands)
            B.n = 0; B.ntiles = 0; B.sort_passes = 0; B.grid.n_valid = 0;
            M3dGrid z; memset(&z, 0, sizeof(z)); z.leaf = B.grid.leaf;
            M->g = z; M->lbound = 0.f;
            for (int a = 0; a < 3; a++) { M->mx[a] = 0.f; M->bits[a] = 0; }
            for (int k = 0; k < 8; k++) M->dyn[k] = 0u;
        }
    }
}

// ---- a3: voxel key per point ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_voxel_keys(const M3dBuild* __restrict__ builds, int rows, int bpr) {
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    const int i = rb.blk * (int)blockDim.x + (int)threadIdx.x;
    if (i >= B.n) return;
    const float4 pi = B.xyz[i];
    const float px = pi.x, py = pi.y, pz = pi.z;
    uint32_t key = M3D_INVALID_KEY;
    if (m3d_finite3(px, py, pz)) {
        const int ix = (int)m3d_cell_f(px, B.grid.mn[0], B.grid.inv_leaf);
        const int iy = (int)m3d_cell_f(py, B.grid.mn[1], B.grid.inv_leaf);
        const int iz = (int)m3d_cell_f(pz, B.grid.mn[2], B.grid.inv_leaf);
        key = m3d_voxel_key(B.grid.cb, ix, iy, iz);
    }
    B.keys[i] = key;   // (pass 0 of the sort reads this array and takes the position as the value: no second copy of the keys, no identity array — 8 bytes per point less
                       //  written here and 4 less read there; a coarser level of a pyramid is keyed again in the finest level's order: k_rekey fills ka / va)
}

// a coarser level of a pyramid starts its sort from the finest level's ORDER (M3dBuild::fine): the LSD passes are stable, so inside a
// coarse voxel the points then lie in the finest level's Morton order
__global__ __launch_bounds__(256) void k_rekey(const M3dBuild* __restrict__ builds, int rows, int bpr) {
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    if (B.fine < 0) return;
    const int i = rb.blk * (int)blockDim.x + (int)threadIdx.x;
    if (i >= B.n) return;
    const uint32_t v = M3D_CHK(201, builds[B.fine].perm_out[i], B.n);
    B.ka[i] = B.keys[v]; B.va[i] = v;
}

// ---- a4: stable LSD radix sort, 8-bit digits ------------------------------------------------------
// pass = histogram per tile -> exclusive scan over [digit][tile] -> stable scatter. A build whose keys
// need fewer passes simply skips the later ones (its result then sits in the buffer of its own parity).
__device__ __forceinline__ void sort_buffers(const M3dBuild& B, int pass, const uint32_t*& kin, const uint32_t*& vin, uint32_t*& kout,
                                             uint32_t*& vout) {
    if (pass & 1) { kin = B.kb; vin = B.vb; kout = B.ka; vout = B.va; }
    else { kin = B.ka; vin = B.va; kout = B.kb; vout = B.vb; }
    if (pass == 0 && B.fine < 0) { kin = B.keys; vin = nullptr; }   // the keys in input order; value = position
    if (pass == B.sort_passes - 1) { kout = B.skey_out; vout = B.perm_out; }   // the last pass scatters straight into the cloud's own arrays
}
__device__ __forceinline__ const uint32_t* sorted_keys(const M3dBuild& B) { return B.skey_out; }
__device__ __forceinline__ const uint32_t* sorted_vals(const M3dBuild& B) { return B.perm_out; }
// ---- a9 without a second sort (round 5) -----------------------------------------------------------------------------------------------
// The normal-estimation grid (leaf = normal_leaf over the same AABB) used to be bucketed like a level: keyed, radix-sorted, tabled, gathered — a third
// of a point-to-plane target's bucketing — only to add up ten integers per voxel. Its voxels are SHORT RUNS of the finest level's sorted order (that
// order follows a Morton curve of 2x2x2-voxel buckets; a coarser voxel is a handful of consecutive buckets, give or take the points the two grids'
// float arithmetic rounds differently), and the moments are exact integers, so they can be added in any order: the runs' heads insert the voxel into an
// open-addressing table (k_finalize_level, fused), the runs store or add their moments to the slot (k_post_finalize),
// the solve workgroups of k_tiles_normals sum the 27 neighbours of every occupied voxel and solve, k_nrm_handout looks every point's voxel up, in every level's order. Same
// integers, same doubles, same bits as the sorted version and as the oracle's per-point loop (oracle/m3d_oracle.c: grid_normals).
__device__ __forceinline__ uint32_t nrm_voxel_key(const M3dGrid& g, float x, float y, float z) {
    const int ix = (int)m3d_cell_f(x, g.mn[0], g.inv_leaf), iy = (int)m3d_cell_f(y, g.mn[1], g.inv_leaf), iz = (int)m3d_cell_f(z, g.mn[2], g.inv_leaf);
    return (uint32_t)ix | ((uint32_t)iy << (g.cb[0] + 1)) | ((uint32_t)iz << (g.cb[0] + g.cb[1] + 2));   // (sum of the widths = cb0 + cb1 + cb2 + 3 <= 31: never M3D_INVALID_KEY)
}
// slot of a voxel that IS in the table
__device__ __forceinline__ uint32_t nrm_slot_of(const uint32_t* __restrict__ nkeys, uint32_t mask, int shift, uint32_t key) {
    uint32_t h = (key * 0x9E3779B1u) >> shift;
    while (nkeys[h] != key) h = (h + 1u) & mask;
    return h;
}

// (fused = the batch's clouds are small enough — at most RS_FUSED_TILES tiles — for every scatter workgroup to scan the counters it
// needs itself: no k_rs_scan launch, counters stored [tile][digit] so that those reads coalesce)
#define RS_FUSED_TILES 128
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const M3dBuild* __restrict__ builds, int pass, int fused, int phase, int rows, int bpr) {
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    if (pass >= B.sort_passes || rb.blk >= B.ntiles || (B.fine >= 0) != (phase == 1)) return;   // phase 1: the grids that wait for their cloud's finest level
    const uint32_t *kin, *vin; uint32_t *kout, *vout;
    sort_buffers(B, pass, kin, vin, kout, vout);
    const int n = B.n, shift = 8 * pass;
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = rb.blk * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = base + r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(kin[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (fused) B.hist[rb.blk * 256 + threadIdx.x] = h[threadIdx.x];
    else B.hist[threadIdx.x * B.ntiles + rb.blk] = h[threadIdx.x];
}

// Clouds of more than RS_FUSED_TILES sort tiles (262 k points: a map): exclusive scan of a build's 256 * ntiles counters, in place, by
// RS_SCAN_CHUNKS workgroups in two launches — chunk sums, then every chunk scans itself behind the sum of the chunks before it, all
// loads coalesced. (One workgroup of 1024 threads, each summing a contiguous 183-counter stretch with a stride of 732 bytes between
// neighbouring lanes, took 223 us per pass for the 1.5 M-point map of config 5: 1.8 of the 2.8 ms its bucketing took.)
#define RS_SCAN_CHUNKS 64
__device__ __forceinline__ int rs_scan_chunk_len(int total) { return ((total + RS_SCAN_CHUNKS - 1) / RS_SCAN_CHUNKS + 255) & ~255; }
__global__ __launch_bounds__(256) void k_rs_scan_sums(const M3dBuild* __restrict__ builds, int pass, int phase, int rows) {
    const M3dBuild& B = builds[m3d_row_build((int)blockIdx.y, rows)];
    if (pass >= B.sort_passes || (B.fine >= 0) != (phase == 1)) return;
    const int total = 256 * B.ntiles, len = rs_scan_chunk_len(total);
    const int b0 = blockIdx.x * len, e0 = min(b0 + len, total);
    uint32_t s = 0;
    for (int i = b0 + (int)threadIdx.x; i < e0; i += 256) s += B.hist[i];
    __shared__ uint32_t w[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) B.hist[total + blockIdx.x] = w[0] + w[1] + w[2] + w[3];   // (the workspace holds 256 spare counters behind the tiles')
}
__global__ __launch_bounds__(256) void k_rs_scan_apply(const M3dBuild* __restrict__ builds, int pass, int phase, int rows) {
    const M3dBuild& B = builds[m3d_row_build((int)blockIdx.y, rows)];
    if (pass >= B.sort_passes || (B.fine >= 0) != (phase == 1)) return;
    const int total = 256 * B.ntiles, len = rs_scan_chunk_len(total);
    const int b0 = blockIdx.x * len, e0 = min(b0 + len, total);
    __shared__ uint32_t w[4], s_run;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) {   // sum of the chunks before this one
        uint32_t v = ((int)threadIdx.x < (int)blockIdx.x) ? B.hist[total + threadIdx.x] : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (threadIdx.x == 0) s_run = v;
    }
    __syncthreads();
    uint32_t run = s_run;
    for (int i0 = b0; i0 < e0; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        const uint32_t v = i < e0 ? B.hist[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        __syncthreads();   // (w is read below in the previous trip)
        if (lane == 63) w[wave] = inc;
        __syncthreads();
        uint32_t off = run;
        for (int k = 0; k < wave; k++) off += w[k];
        if (i < e0) B.hist[i] = off + inc - v;
        run += w[0] + w[1] + w[2] + w[3];
    }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const M3dBuild* __restrict__ builds, int pass, int fused, int phase, int rows, int bpr, int sliced) {
    const M3dRB rb = m3d_row_block_sliced(bpr, sliced);
    const int tile = rb.blk;
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    if (tile >= bpr || pass >= B.sort_passes || tile >= B.ntiles || (B.fine >= 0) != (phase == 1)) return;
    const uint32_t *kin, *vin; uint32_t *kout, *vout;
    sort_buffers(B, pass, kin, vin, kout, vout);
    const int n = B.n, shift = 8 * pass, ntiles = B.ntiles;
    const uint32_t* scanned = B.hist;
    __shared__ uint32_t cnt[RS_ROUNDS * RS_WAVES][256];   // 32 KiB: per (round, wave) digit counts
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t goff;   // where this tile's keys of digit t go: after all keys of smaller digits and this digit's keys of the tiles before
    // (the four wave sums of the scan below ride in cnt itself — wave w's in cnt[w][64 w + 63], a word only wave w's lanes write afterwards: with 16 bytes of
    //  their own the kernel's LDS was 32 784 B, four workgroups per CU instead of five)
    uint32_t total = 0, inc = 0;
    if (fused) {
        // the scan a separate single-workgroup launch did (10 us per pass, launch gap included), redone by every workgroup for its own
        // tile: digit t's counters of all tiles (coalesced, eight independent loads per trip), then a scan of the 256 digit totals
        // (shuffles inside a wave, the four wave sums through LDS at the barrier below)
        uint32_t before = 0;
        int j = 0;
        // (32 loads in flight, then 8, then the rest: a 100 k-point cloud has 49 tiles — seven trips of eight were seven round trips at the head of every scatter workgroup)
        for (; j + 32 <= ntiles; j += 32) {
            uint32_t v[32];
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = scanned[(j + k) * 256 + t];
#pragma unroll
            for (int k = 0; k < 32; k++) { total += v[k]; before += (j + k < tile) ? v[k] : 0u; }
        }
        {   // up to 31 left: 32 predicated loads in ONE trip
            uint32_t v[32];
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = (j + k < ntiles) ? scanned[(j + k) * 256 + t] : 0u;
#pragma unroll
            for (int k = 0; k < 32; k++) { total += v[k]; before += (j + k < tile) ? v[k] : 0u; }
        }
        inc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        for (int s = 0; s < RS_ROUNDS * RS_WAVES; s++) cnt[s][t] = 0;
        if (lane == 63) cnt[wave][t] = inc;   // (behind this thread's own zeroing of that word)
        goff = before;
    } else {
        goff = scanned[t * ntiles + tile];
        for (int s = 0; s < RS_ROUNDS * RS_WAVES; s++) cnt[s][t] = 0;
    }
    __syncthreads();
    if (fused) {   // (uniform)
        for (int w = 0; w < wave; w++) goff += cnt[w][64 * w + 63];
        goff += inc - total;
        __syncthreads();   // every wave has read the sums
        if (lane == 63) cnt[wave][t] = 0;   // (LDS operations of one wave complete in order: the counts this wave writes below come behind it)
    }
    const int base = tile * RS_TILE;
    uint32_t key[RS_ROUNDS], val[RS_ROUNDS], rank[RS_ROUNDS];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = base + r * RS_THREADS + t;
        const bool ok = i < n;
        key[r] = ok ? kin[i] : 0u;
        val[r] = ok ? (vin ? vin[i] : (uint32_t)i) : 0u;
        const uint32_t d = (key[r] >> shift) & 255u;
        // lanes of this wave holding the same digit: 8 ballots (wave64 "match_any")
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(ok && bit);
            m &= bit ? bal : ~bal;
        }
        rank[r] = (uint32_t)__popcll(m & lt);
        if (ok && rank[r] == 0) cnt[r * RS_WAVES + wave][d] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    {   // thread t owns digit t: exclusive scan over the (round, wave) slots, seeded with the global offset
        uint32_t run = goff;
        for (int s = 0; s < RS_ROUNDS * RS_WAVES; s++) { uint32_t v = cnt[s][t]; cnt[s][t] = run; run += v; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = base + r * RS_THREADS + t;
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t pos = cnt[r * RS_WAVES + wave][d] + rank[r];
            { const uint32_t pc = M3D_CHK(202, pos, n); kout[pc] = key[r]; vout[pc] = val[r]; }
        }
    }
}

// ---- a4: bucket heads -> exact-sized hash table, sorted float4 points -----------------------------
// dyn[0] = occupied voxels, dyn[1] = hmask, dyn[2] = hshift, dyn[3] = occupied buckets, dyn[4] = big buckets
// (table geometry is derived on the device: no host round trip between the sort and the table build)
// (occupied voxels and buckets are counted per 256-position block, without atomics: contended returning atomics on one
// address retire at ~5 per microsecond on this chip — 1500 of them were the whole cost of every variant that used them)
__global__ __launch_bounds__(256) void k_count_cells(const M3dBuild* __restrict__ builds, int rows, int bpr) {
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    const int n = B.n;
    if (rb.blk * 256 >= n) return;
    const uint32_t* skey = sorted_keys(B);
    const int j = rb.blk * 256 + (int)threadIdx.x;
    bool vh = false, bh = false;
    uint32_t k = M3D_INVALID_KEY;
    if (j < n) {
        k = skey[j];
        if (k != M3D_INVALID_KEY) {
            const uint32_t kp = j ? skey[j - 1] : M3D_INVALID_KEY;
            vh = (j == 0) || (kp != k);
            bh = (j == 0) || ((kp >> 3) != (k >> 3));
        }
    }
    __shared__ uint32_t red[4][2];
    __shared__ float redf[4];
    const unsigned long long bv = __ballot(vh), bb = __ballot(bh);
    // population of every voxel whose head lies in this block, squared and summed (M3dLevelMeta::sumsq): the next head inside the wave ends the run;
    // the wave's last run is measured by a bisection of the sorted keys (one lane per wave)
    float sq = 0.f;
    {
        const int lane = (int)(threadIdx.x & 63u);
        // Only the wave's LAST run can reach beyond the wave: how many lanes of this wave it covers (its key is the last head's; the keys are sorted, so
        // they are contiguous), and — when it covers the wave's last lane — how many of the NEXT 64 positions continue it: every lane looks at one of them
        // (one coalesced load). Only a run that covers all of those too (a crowded voxel) gallops on from there, one lane.
        const int last_head = bv ? 63 - __clzll((long long)bv) : 0;
        const uint32_t k_last = (uint32_t)__shfl((int)k, last_head);
        const unsigned long long eqm = __ballot(bv != 0ull && k == k_last);            // (valid keys only: a head's key is valid)
        const bool reaches_end = (eqm >> 63) != 0ull;
        const int jn = j + 64;
        const bool same = reaches_end && jn < n && skey[jn] == k_last;
        const unsigned long long sm = __ballot(same);
        const int ext = (sm == ~0ull) ? 64 : __ffsll((long long)~sm) - 1;               // leading positions of the next 64 that continue the run
        if (vh) {
            const unsigned long long later = lane == 63 ? 0ull : (bv >> (lane + 1));
            int len;
            if (later) len = __ffsll((long long)later);
            else {
                len = (int)__popcll(eqm) + ext;                                          // (this lane IS the last head: the run's lanes are exactly eqm's bits)
                if (ext == 64) {   // (rare) gallop on: skey[lo] == k, hi = a position known not to hold k
                    int lo = j + len - 1, step = 64;
                    while (lo + step < n && skey[lo + step] == k) { lo += step; step *= 2; }
                    int hi = lo + step < n ? lo + step : n;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (skey[mid] == k) lo = mid; else hi = mid; }
                    len = hi - j;
                }
            }
            sq = (float)len * (float)len;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = (uint32_t)__popcll(bv); red[threadIdx.x >> 6][1] = (uint32_t)__popcll(bb); redf[threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {   // B.hist (free after the sort): [2 blk] = voxel heads, [2 blk + 1] = bucket heads of this block; behind them, [2 nblk + blk] = its sum of squares
        B.hist[2 * rb.blk] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        B.hist[2 * rb.blk + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        B.hist[2 * ((n + 255) / 256) + rb.blk] = __float_as_uint(((redf[0] + redf[1]) + redf[2]) + redf[3]);
    }
}

__host__ __device__ inline void m3d_table_size(uint32_t n_buckets, uint32_t hcap, uint32_t& hmask, int& hshift) {
    uint32_t hs = 16; int hb = 4;
    while (hs < 2u * n_buckets && hs < hcap) { hs <<= 1; hb++; }
    hmask = hs - 1; hshift = 32 - hb;
}

// Longest-processing-time-first order of a cloud's 256-point blocks (it is the SOURCE of a registration that uses it): a block that
// spans few voxels is a crowded stretch of the cloud (a surface close to the sensor) — its queries meet crowded target voxels and
// walk several times as many candidates as the rest. Started last, such blocks were the tail that set k_nn_iter's duration; started
// first, they run while the other blocks fill the machine. k_table_params (one workgroup per grid, it has the per-block voxel
// counts in hand) ranks the blocks by occupied voxels, ascending.
#define M3D_ORDER_CAP 8192

// one workgroup per build: totals of the per-block counts (dyn[0] voxels, dyn[3] buckets), exclusive prefix of the voxel
// heads in place (where each block of k_finalize_level appends its heads), table geometry
__global__ __launch_bounds__(256) void k_table_params(const M3dBuild* __restrict__ builds, int rows) {
    const M3dBuild& B = builds[m3d_row_build((int)blockIdx.x, rows)];
    const int nblk = (B.n + 255) / 256;
    __shared__ uint32_t sh[2][4];
    __shared__ uint32_t wv[M3D_ORDER_CAP];   // occupied voxels of every 256-point block (k_count_cells), for the block order below
    const bool want_order = B.order != nullptr && nblk <= M3D_ORDER_CAP;
    uint32_t carryV = 0, carryB = 0;
    const int t = threadIdx.x;
    float sq = 0.f;   // M3dLevelMeta::sumsq: the blocks' sums in a fixed order (a thread's blocks ascending, then a fixed reduction tree)
    for (int base = 0; base < nblk; base += 256) {
        const int b = base + t;
        if (b < nblk) sq += __uint_as_float(B.hist[2 * nblk + b]);
        const uint32_t vV = b < nblk ? B.hist[2 * b] : 0u, vB = b < nblk ? B.hist[2 * b + 1] : 0u;
        if (want_order && b < nblk) wv[b] = vV;
        // inclusive scans of the two counts over the 256 threads: shuffles inside a wave, the wave totals through LDS (sixteen barriers of a
        // Hillis-Steele scan in LDS before)
        uint32_t iV = vV, iB = vB;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t uV = __shfl_up(iV, o), uB = __shfl_up(iB, o); if ((t & 63) >= o) { iV += uV; iB += uB; } }
        if ((t & 63) == 63) { sh[0][t >> 6] = iV; sh[1][t >> 6] = iB; }
        __syncthreads();
        uint32_t bV = 0, bB = 0, totV = 0, totB = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t a = sh[0][w], c = sh[1][w]; if (w < (t >> 6)) { bV += a; bB += c; } totV += a; totB += c; }
        iV += bV; iB += bB;
        if (b < nblk) B.hist[2 * b] = carryV + iV - vV;
        carryV += totV; carryB += totB;
        __syncthreads();
    }
    {
        __shared__ float sqw[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if ((t & 63) == 0) sqw[t >> 6] = sq;
        __syncthreads();
        if (t == 0) reinterpret_cast<M3dLevelMeta*>(B.dyn)->sumsq = ((sqw[0] + sqw[1]) + sqw[2]) + sqw[3];
    }
    if (t == 0) {
        uint32_t hmask; int hshift;
        m3d_table_size(carryB, B.hcap, hmask, hshift);
        B.dyn[0] = carryV; B.dyn[3] = carryB; B.dyn[5] = carryV; B.dyn[4] = 0u; B.dyn[6] = 0u;   // every word this pipeline reads is written here: no memset needed (dyn[6]: pool images handed out by the tile workgroups of k_tiles_normals)
        B.dyn[1] = hmask; B.dyn[2] = (uint32_t)hshift;
    }
    if (B.order) {   // (the loop above ended with a barrier: wv is complete)
        if (!want_order) {   // a cloud of more than 2 M points: natural order
            for (int b = t; b < nblk; b += 256) B.order[b] = (uint32_t)b;
            return;
        }
        // counting sort on the voxel count (1 .. 256; blocks of equal count in any order — the order only schedules work). Ranking
        // every block against every other one took 16 of this kernel's 21 us.
        __shared__ uint32_t bin[257];
        bin[t] = 0u; if (t == 0) bin[256] = 0u;
        __syncthreads();
        for (int b = t; b < nblk; b += 256) atomicAdd(&bin[min(wv[b], 256u)], 1u);
        __syncthreads();
        {   // exclusive scan of the 257 bins: shuffles inside a wave, the four wave totals through LDS (one thread walking the bins was 257 dependent LDS
            // round trips: 8 of this kernel's 10 us)
            const uint32_t c = bin[t];
            uint32_t inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
            __shared__ uint32_t wtot[4];
            if ((t & 63) == 63) wtot[t >> 6] = inc;
            __syncthreads();
            uint32_t before = 0;
            for (int w = 0; w < (t >> 6); w++) before += wtot[w];
            bin[t] = before + inc - c;
            if (t == 255) bin[256] = before + inc;   // (bin 256 — blocks of 256 and more occupied voxels — starts behind all others; its own count is not needed)
        }
        __syncthreads();
        for (int b = t; b < nblk; b += 256) B.order[atomicAdd(&bin[min(wv[b], 256u)], 1u)] = (uint32_t)b;
    }
}

__global__ __launch_bounds__(256) void k_clear_table(const M3dBuild* __restrict__ builds, int rows, int bpr) {
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    if (B.nrm_feed) {   // the level that feeds its cloud's normal grid clears that grid's voxel table (keys only: a slot's moments are zeroed by the thread that takes it)
        const M3dBuild& G = builds[B.nrm_build];
        uint4* t4 = reinterpret_cast<uint4*>(G.nkeys);   // [ncap] keys, then [ncap] "further runs of this voxel" counters
        const uint32_t n4 = G.ncap >> 2;   // (a power of two >= 16)
        for (uint32_t i = (uint32_t)rb.blk * blockDim.x + threadIdx.x; i < 2u * n4; i += (uint32_t)bpr * blockDim.x) {
            const uint32_t f = i < n4 ? M3D_INVALID_KEY : 0u;
            t4[i] = make_uint4(f, f, f, f);
        }
    }
    if (!B.htab) return;   // a source-only cloud has no table
    const uint32_t T2 = 2u * (B.dyn[1] + 1u);
    uint4* t = reinterpret_cast<uint4*>(B.htab);
    for (uint32_t i = (uint32_t)rb.blk * blockDim.x + threadIdx.x; i < T2; i += (uint32_t)bpr * blockDim.x)
        t[i] = (i & 1u) ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(M3D_INVALID_KEY, 0u, 0u, 0u);
    const uint32_t nb = 8u * B.bigcap;
    for (uint32_t i = (uint32_t)rb.blk * blockDim.x + threadIdx.x; i < nb; i += (uint32_t)bpr * blockDim.x) B.bigcum[i] = 0u;
    if (B.occ) {   // occupancy bitmap of the bucket positions: one bit per bucket key, when the grid is small enough
        const int kb = B.grid.cb[0] + B.grid.cb[1] + B.grid.cb[2];
        const bool ok = kb <= M3D_OCC_BITS;
        if (rb.blk == 0 && threadIdx.x == 0) B.dyn[7] = ok ? 1u : 0u;
        if (ok) {
            const uint32_t nw = kb > 5 ? (1u << (kb - 5)) : 1u;
            for (uint32_t i = (uint32_t)rb.blk * blockDim.x + threadIdx.x; i < nw; i += (uint32_t)bpr * blockDim.x) B.occ[i] = 0u;
        }
    }
}

__device__ __forceinline__ uint32_t bucket_key_of_point(const M3dGrid& g, const float4& p) {
    const int ix = (int)m3d_cell_f(p.x, g.mn[0], g.inv_leaf), iy = (int)m3d_cell_f(p.y, g.mn[1], g.inv_leaf),
              iz = (int)m3d_cell_f(p.z, g.mn[2], g.inv_leaf);
    return m3d_bucket_key(g, ix >> 1, iy >> 1, iz >> 1);
}

__global__ __launch_bounds__(256) void k_finalize_level(const M3dBuild* __restrict__ builds, int rows, int bpr, int sliced) {
    const M3dRB rb = m3d_row_block_sliced(bpr, sliced);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    const int j = rb.blk * (int)blockDim.x + (int)threadIdx.x;
    const int n = B.n;
    if (rb.blk >= bpr || rb.blk * (int)blockDim.x >= n) return;   // block-uniform
    const bool inb = j < n;
    const uint32_t* skey = sorted_keys(B);
    const uint32_t* sval = sorted_vals(B);
    const uint32_t hmask = B.dyn[1];
    const int hshift = (int)B.dyn[2];
    const uint32_t k = inb ? skey[j] : M3D_INVALID_KEY;
    const uint32_t oi = inb ? M3D_CHK(203, sval[j], n) : 0u;
    const bool valid = k != M3D_INVALID_KEY;
    const uint32_t kp = (inb && j) ? skey[j - 1] : M3D_INVALID_KEY;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inb) {
        p = B.xyz[oi];    // one 16-B gather per point (three 4-B gathers from SoA arrays touched three cache lines)
        p.w = __uint_as_float(oi);   // bits of the input index (< 2^28): the tie-break key of the NN search, and the way back to input order
        B.pts[j] = p;
        if (B.src3) { B.src3[3 * (size_t)j] = p.x; B.src3[3 * (size_t)j + 1] = p.y; B.src3[3 * (size_t)j + 2] = p.z; }
    }
    if (B.nrm_feed) {   // (block-uniform) the finest level of a cloud with normals: the heads of the normal-grid voxels' runs take the voxels' slots
        const M3dBuild& G = builds[B.nrm_build];
        const uint32_t nmask = G.ncap - 1u;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t nk = M3D_INVALID_KEY;
        if (valid) nk = nrm_voxel_key(G.grid, p.x, p.y, p.z);
        uint32_t nkp = (uint32_t)__shfl_up((int)nk, 1);
        if (lane == 0) {   // (the wave's first position: its predecessor's point once more — finite, like every point before a finite one)
            nkp = M3D_INVALID_KEY;
            if (valid && j > 0) { const float4 pp = B.xyz[sval[j - 1]]; nkp = nrm_voxel_key(G.grid, pp.x, pp.y, pp.z); }
        }
        bool won = false;
        uint32_t h = 0u;
        if (valid && (j == 0 || nkp != nk)) {
            h = (nk * 0x9E3779B1u) >> G.nshift;
            for (;;) {   // the first head of a voxel to arrive owns its slot; a later run of the same voxel finds it
                const uint32_t old = atomicCAS(&G.nkeys[h], M3D_INVALID_KEY, nk);
                if (old == M3D_INVALID_KEY) { won = true; break; }
                if (old == nk) { atomicAdd(&G.nkeys[G.ncap + h], 1u); break; }   // a further run of a voxel that has its slot: k_post_finalize then ADDS this voxel's runs instead of storing its one
                h = (h + 1u) & nmask;
            }
            if (won) {
#pragma unroll
                for (int i = 0; i < 10; i++) G.mom[10 * (size_t)h + i] = 0;   // (k_post_finalize, the next launch, stores or adds)
            }
        }
        // the slots this block took, as a list (ranks: wave64 ballots + LDS; no atomics): the work items of the normals' solve
        __shared__ uint32_t s_wc[4];
        const unsigned long long bw = __ballot(won);
        if (lane == 0) s_wc[wave] = (uint32_t)__popcll(bw);
        __syncthreads();
        uint32_t off = 0u, tot = 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) { if (w < wave) off += s_wc[w]; tot += s_wc[w]; }
        if (won) G.nlist[(size_t)rb.blk * 256u + M3D_CHK(204, off + (uint32_t)__popcll(bw & ((1ull << lane) - 1ull)), 256u)] = h;
        if (threadIdx.x == 0) G.nvcnt[rb.blk] = tot;
    }
    if (!inb) return;
    const bool bhead = valid && (j == 0 || (kp >> 3) != (k >> 3));
    if (bhead && B.htab) {
        const uint32_t bk = bucket_key_of_point(B.grid, p);
        uint32_t h = m3d_hash_slot(bk, hshift);
        for (;;) {   // bucket keys are unique here, so a successful CAS owns the slot
            uint32_t old = atomicCAS(&B.htab[h].key, M3D_INVALID_KEY, bk);
            if (old == M3D_INVALID_KEY) {
                B.htab[h].start = (uint32_t)j;
                // more than 65535 points in this bucket? (the sorted keys say so: position j + 65535 still belongs to it) Then its cumulative populations are 32-bit rows
                // of bigcum, row = j >> 16: two such buckets start at least 65536 positions apart, so the rows are distinct without a counter
                const bool big = (uint32_t)j + 65535u < (uint32_t)n && (skey[(uint32_t)j + 65535u] >> 3) == (k >> 3);
                B.htab[h].big = big ? 1u + ((uint32_t)j >> 16) : 0u;
                break;
            }
            h = (h + 1) & hmask;
        }
        if (B.occ && B.grid.cb[0] + B.grid.cb[1] + B.grid.cb[2] <= M3D_OCC_BITS) atomicOr(&B.occ[bk >> 5], 1u << (bk & 31u));
    }
}

// (the per-voxel populations of the bucket table, the chunk boxes and the normal grid's moments: k_post_finalize, behind the normals' helpers below)

// ---- target tiles: what a workgroup of the LDS-staged search (icp.hip: k_nn_tiles) holds in LDS -------------------------------------
// One workgroup per tile (m3d_device.h): the tile's own buckets are the bucket heads among its M3D_TILE_PTS sorted positions; every
// occupied bucket within one bucket of an own one (27 probes of the level's hash table per own bucket) joins them in an LDS hash set
// (CAS on the key: duplicates fall out). The set is then numbered in slot order, the populations are scanned into LDS positions, and
// the image is written: directory slots, entries, and the sorted position of every staged point. Built once per target cloud and
// level; every Gauss-Newton iteration of every registration against it re-uses it.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_w, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += t; }
    __syncthreads();                       // s_w may still be read from a previous call
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) off += s_w[w]; }
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return off + incl - v;
}

#define M3D_TILE_CS 4096   // tile_build_role: slots of the candidate-position set (bucket positions around the own buckets, occupied or not)
#ifdef M3D_TB_STAMPS   // diagnosis build: wall-clock stamps (100 MHz) at the phase boundaries of the first 4096 working workgroups
__device__ unsigned long long g_tb_stamp[4096][10];
__device__ unsigned int g_tb_n = 0;
extern "C" hipError_t m3d_debug_read_tb(unsigned long long* out, unsigned int* n) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tb_stamp), sizeof(unsigned long long) * 4096 * 10);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(n, HIP_SYMBOL(g_tb_n), sizeof(unsigned int));
    unsigned int z = 0; if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_tb_n), &z, sizeof(z));
    return e;
}
#define TB_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0 && tb_slot < 4096u) g_tb_stamp[tb_slot][k] = wall_clock64(); } while (0)
#else
#define TB_STAMP(k) ((void)0)
#endif
__device__ __forceinline__ void tile_build_role(const M3dBuild* __restrict__ builds, int row_stride, int row_first, int bpr, int sliced, int bid) {
    const M3dRB rb = m3d_row_block_sliced(bpr, sliced, bid);
    if (rb.blk >= bpr) return;
    const M3dBuild& B = builds[rb.row * row_stride + row_first];   // (a row per cloud: its last build — tiles on finest levels only)
    if (!B.thdr || !B.htab) return;
    const int nv = B.grid.n_valid;
    const int t = rb.blk, p0 = t * M3D_TILE_PTS;
    if (p0 >= nv) return;
    const int p1 = min(p0 + M3D_TILE_PTS, nv);
    // LDS: the candidate set is dead once the staged-bucket list exists: the voxel directory of the image being written lives there
    __shared__ uint32_t s_a[M3D_TILE_CS];            // phase 2/3: candidate bucket keys | phase 6: voxel directory {keys[VS], values[VS]}
    static_assert(M3D_TILE_CS >= 2 * M3D_TILE_VS, "the voxel directory re-uses the candidate set's LDS");
    uint32_t* s_ck = s_a; uint32_t* s_vk = s_a; uint32_t* s_vv = s_a + M3D_TILE_VS;
    __shared__ uint32_t s_lk[M3D_TILE_ECAP], s_lg[M3D_TILE_ECAP], s_lp[M3D_TILE_ECAP];   // staged buckets: key, slot in the level's table, points | voxels << 16 (later image | offset << 8)
    __shared__ uint32_t s_head[M3D_TILE_PTS];
    __shared__ uint32_t s_src[M3D_TILE_PCAP];     // sorted position of every staged point of the image being written
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_ip[M3D_TILE_MAXIMG];   // points | voxels << 16 of every image
    __shared__ uint32_t s_cnt, s_over, s_nv, s_crowd;
    const int tid = threadIdx.x;
    const M3dGrid& g = B.grid;
    const uint32_t hmask = B.dyn[1];
    const int hshift = (int)B.dyn[2];
    const uint32_t* skey = B.skey_out;
    const uint4* tab = reinterpret_cast<const uint4*>(B.htab);
    const bool occ_ok = B.occ != nullptr && g.cb[0] + g.cb[1] + g.cb[2] <= M3D_OCC_BITS;
#ifdef M3D_TB_STAMPS
    __shared__ unsigned int s_tbslot;
    if (threadIdx.x == 0) s_tbslot = atomicAdd(&g_tb_n, 1u);
    __syncthreads();
    const unsigned int tb_slot = s_tbslot;
#endif
    TB_STAMP(0);
    // (the set holds occupied positions only when the bitmap exists: at most ECAP = 512 of them may be staged, 1024 slots do; without it every
    // candidate position goes in: 4096)
    const int cs_bits = occ_ok ? 10 : 12;
    const uint32_t cs_mask = (1u << cs_bits) - 1u;
    for (int i = tid; i <= (int)cs_mask; i += 256) s_ck[i] = M3D_INVALID_KEY;
    if (tid == 0) { s_cnt = 0u; s_over = 0u; s_crowd = 0u; }
    // 1. own bucket heads, in sorted order; their bucket coordinates (one point load per head)
    uint32_t nheads = 0;
#pragma unroll
    for (int r = 0; r < (M3D_TILE_PTS + 255) / 256; r++) {   // (any tile size: the last round is partial)
        const int j = p0 + r * 256 + tid;
        bool head = false;
        if (j < p1) { const uint32_t k = skey[j]; head = (j == 0) || ((skey[j - 1] >> 3) != (k >> 3)); }
        uint32_t tot;
        const uint32_t pos = block_excl_scan_256(head ? 1u : 0u, s_w, tot);
        if (head) s_head[nheads + pos] = (uint32_t)j;
        nheads += tot;
    }
    __syncthreads();
    for (uint32_t hd = (uint32_t)tid; hd < nheads; hd += 256u) {
        const float4 p = B.pts[M3D_CHK(208, s_head[hd], B.n)];
        // (packed with the grid's own bit widths — the bucket key: sum of the widths <= 28; fixed 11 / 11 / 10-bit fields lost the top bit of z on a grid
        // of more than 2048 voxels in z)
        s_head[hd] = m3d_bucket_key(g, (int)m3d_cell_f(p.x, g.mn[0], g.inv_leaf) >> 1, (int)m3d_cell_f(p.y, g.mn[1], g.inv_leaf) >> 1, (int)m3d_cell_f(p.z, g.mn[2], g.inv_leaf) >> 1);
    }
    __syncthreads();
    TB_STAMP(1);
    // 2. the 27 bucket positions around every own bucket, as a set (LDS only: neighbouring own buckets share most of them)
    const int nb0 = (g.dims[0] + 1) >> 1, nb1 = (g.dims[1] + 1) >> 1, nb2 = (g.dims[2] + 1) >> 1;
    // (with the occupancy bitmap: only OCCUPIED positions enter the set — eight bitmap words in flight per thread and trip; an empty
    // position costs one 4-byte load and no LDS traffic)
    for (uint32_t item0 = (uint32_t)tid; item0 < nheads * 27u; item0 += 8u * 256u) {
        uint32_t key[8], wbit[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t item = item0 + 256u * (uint32_t)u;
            key[u] = M3D_INVALID_KEY; wbit[u] = 0xFFFFFFFFu;
            if (item >= nheads * 27u) continue;
            const uint32_t hd = item / 27u, d = item - hd * 27u;
            const uint32_t c = s_head[hd];
            const int cx = (int)(c & ((1u << g.cb[0]) - 1u)) + (int)(d % 3u) - 1, cy = (int)((c >> g.cb[0]) & ((1u << g.cb[1]) - 1u)) + (int)((d / 3u) % 3u) - 1,
                      cz = (int)(c >> (g.cb[0] + g.cb[1])) + (int)(d / 9u) - 1;
            if (cx < 0 || cy < 0 || cz < 0 || cx >= nb0 || cy >= nb1 || cz >= nb2) continue;
            key[u] = m3d_bucket_key(g, cx, cy, cz);
            if (occ_ok) wbit[u] = B.occ[key[u] >> 5];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t k = key[u];
            if (k == M3D_INVALID_KEY || !((wbit[u] >> (occ_ok ? (k & 31u) : 0u)) & 1u)) continue;
            uint32_t h = (k * 0x9E3779B1u) >> (32 - cs_bits);
            static_assert(M3D_TILE_CS == 4096, "candidate set hash: at most 12 bits");
            int tries = 0;
            for (; tries <= (int)cs_mask; tries++) {
                const uint32_t old = atomicCAS(&s_ck[h], M3D_INVALID_KEY, k);
                if (old == M3D_INVALID_KEY || old == k) break;
                h = (h + 1u) & cs_mask;
            }
            if (tries > (int)cs_mask) s_over = 1u;   // (set full: more occupied positions than can be staged anyway)
        }
    }
    __syncthreads();
    TB_STAMP(2);
    // 3. the occupied candidate positions form the staged list. With the occupancy bitmap: sixteen 4-byte loads per thread answer
    // "empty" for most candidates at once, the few occupied ones are then looked up in the level's table for their entry; without it
    // (a grid of more than 2^23 bucket positions): one probe chain per candidate, four in flight.
    if (occ_ok) {
        uint32_t key[4];
#pragma unroll
        for (int r = 0; r < 4; r++) key[r] = s_ck[256 * r + tid];   // (the set's 1024 slots hold occupied positions only)
        // the occupied candidates form the list first (LDS), THEN every thread looks one of them up in the level's table: a thread that
        // found three occupied ones among its sixteen made three dependent probes while most of the others had none (12 of this
        // kernel's 28 us per workgroup)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (key[r] == M3D_INVALID_KEY) continue;
            const uint32_t e = atomicAdd(&s_cnt, 1u);
            if (e < (uint32_t)M3D_TILE_ECAP) s_lk[e] = key[r];
        }
        __syncthreads();
        const uint32_t n_occ = min(s_cnt, (uint32_t)M3D_TILE_ECAP);
        for (uint32_t e = (uint32_t)tid; e < n_occ; e += 256u) {   // (keeping both halves of the entry in registers for step 4 cost more occupancy than the round trip it saves)
            const uint32_t k = s_lk[e];
            uint32_t slot = m3d_hash_slot(k, hshift);
            uint4 lo = tab[2 * (size_t)slot];
            while (lo.x != k) { slot = (slot + 1u) & hmask; lo = tab[2 * (size_t)slot]; }   // (the bucket exists)
            s_lg[e] = slot;
        }
    } else {
        for (int s0 = 0; s0 < M3D_TILE_CS; s0 += 1024) {
            uint32_t key[4], slot[4]; uint4 lo[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                key[r] = s_ck[s0 + 256 * r + tid];
                slot[r] = m3d_hash_slot(key[r], hshift);
                lo[r] = make_uint4(M3D_INVALID_KEY, 0u, 0u, 0u);
                if (key[r] != M3D_INVALID_KEY) lo[r] = tab[2 * (size_t)slot[r]];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (key[r] == M3D_INVALID_KEY) continue;
                while (lo[r].x != key[r] && lo[r].x != M3D_INVALID_KEY) { slot[r] = (slot[r] + 1u) & hmask; lo[r] = tab[2 * (size_t)slot[r]]; }
                if (lo[r].x != key[r]) continue;
                const uint32_t e = atomicAdd(&s_cnt, 1u);
                if (e < (uint32_t)M3D_TILE_ECAP) { s_lk[e] = key[r]; s_lg[e] = slot[r]; }
            }
        }
    }
    __syncthreads();
    TB_STAMP(3);
    M3dTileHdr* H = B.thdr + t;
    const uint32_t n_e = s_cnt;
    if (s_over != 0u || n_e > (uint32_t)M3D_TILE_ECAP) { if (tid == 0) *H = M3dTileHdr{ 0u, 0u, M3D_TILE_OVERSIZE, 0u }; return; }   // block-uniform
    // 4. populations of the staged buckets (thread tid: entries tid and tid + 256)
    constexpr int SPT = M3D_TILE_ECAP / 256;
    uint32_t kk[SPT], cnt[SPT], nvx[SPT], gst[SPT]; uint4 hi[SPT];
    bool occ[SPT];
    bool anybig = false, crowd = false;
    uint32_t vmax = 0u;   // largest voxel population among this thread's staged buckets
    uint32_t my_p = 0u, my_v = 0u;
#pragma unroll
    for (int q = 0; q < SPT; q++) {
        const uint32_t e = (uint32_t)(tid + 256 * q);
        occ[q] = e < n_e; kk[q] = 0u; cnt[q] = 0u; nvx[q] = 0u; gst[q] = 0u; hi[q] = make_uint4(0u, 0u, 0u, 0u);
        if (occ[q]) {
            kk[q] = s_lk[e];
            const uint32_t gs = s_lg[e];
            const uint4 lo = tab[2 * (size_t)gs];
            hi[q] = tab[2 * (size_t)gs + 1];
            cnt[q] = lo.z; gst[q] = lo.y;
            anybig = anybig || lo.w != 0u || cnt[q] > (uint32_t)M3D_TILE_PCAP;
            const unsigned long long cumA = ((unsigned long long)hi[q].y << 32) | hi[q].x, cumB = ((unsigned long long)hi[q].w << 32) | hi[q].z;
            uint32_t c0 = 0u;
            for (int sub = 0; sub < 8; sub++) {
                const uint32_t c1 = (uint32_t)(((sub < 4) ? cumA : cumB) >> (16 * (sub & 3))) & 0xFFFFu;
                nvx[q] += c1 > c0 ? 1u : 0u; crowd = crowd || (c1 - c0 > (uint32_t)M3D_LONG_ROW); vmax = max(vmax, c1 - c0); c0 = c1;
            }
        }
        my_p += cnt[q]; my_v += nvx[q];
    }
    uint32_t tot_p, tot_v;
    const uint32_t l0 = block_excl_scan_256(my_p, s_w, tot_p);
    (void)block_excl_scan_256(my_v, s_w, tot_v);
    __syncthreads();
    if (anybig) s_over = 1u;
    __syncthreads();
    if (s_over != 0u) { if (tid == 0) *H = M3dTileHdr{ 0u, 0u, M3D_TILE_OVERSIZE, 0u }; return; }
    TB_STAMP(4);
    // 5. which image every staged bucket goes to, and where: one image when everything fits (nearly always); else a greedy cut in list
    // order by one thread (a crowded stretch: a few tiles per cloud)
    uint32_t im[SPT], off[SPT];
    {
        uint32_t l = l0;
#pragma unroll
        for (int q = 0; q < SPT; q++) { im[q] = 0u; off[q] = l; l += cnt[q]; }
    }
    uint32_t n_img = 1u;
    if (tot_p > (uint32_t)M3D_TILE_PCAP || tot_v > (uint32_t)M3D_TILE_VCAP) {
#pragma unroll
        for (int q = 0; q < SPT; q++) if (occ[q]) s_lp[tid + 256 * q] = cnt[q] | (nvx[q] << 16);
        __syncthreads();
        if (tid == 0) {
            uint32_t img_i = 0u, p = 0u, v = 0u;
            for (uint32_t e = 0; e < n_e; e++) {
                const uint32_t w = s_lp[e], bp = w & 0xFFFFu, bv = w >> 16;
                if (p + bp > (uint32_t)M3D_TILE_PCAP || v + bv > (uint32_t)M3D_TILE_VCAP) { s_ip[img_i < M3D_TILE_MAXIMG ? img_i : 0] = p | (v << 16); img_i++; p = 0u; v = 0u; }
                s_lp[e] = (img_i & 0xFFu) | (p << 8);
                p += bp; v += bv;
            }
            s_ip[img_i < M3D_TILE_MAXIMG ? img_i : 0] = p | (v << 16);
            s_cnt = img_i + 1u;
        }
        __syncthreads();
        n_img = s_cnt;
#pragma unroll
        for (int q = 0; q < SPT; q++) if (occ[q]) { const uint32_t w = s_lp[tid + 256 * q]; im[q] = w & 0xFFu; off[q] = w >> 8; }
        __syncthreads();
        if (n_img > (uint32_t)M3D_TILE_MAXIMG) { if (tid == 0) *H = M3dTileHdr{ 0u, 0u, M3D_TILE_OVERSIZE, 0u }; return; }
        if (tid == 0) {   // the extra images come from the level's pool
            const uint32_t base = atomicAdd(&B.dyn[6], n_img - 1u);
            s_cnt = (base + n_img - 1u <= (uint32_t)m3d_tile_pool(m3d_tiles_of(B.n))) ? (uint32_t)m3d_tiles_of(B.n) + base : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (s_cnt == 0xFFFFFFFFu) { if (tid == 0) *H = M3dTileHdr{ 0u, 0u, M3D_TILE_OVERSIZE, 0u }; return; }
    } else if (tid == 0) { s_ip[0] = tot_p | (tot_v << 16); s_cnt = 0u; }
    // (a word of its own, zeroed at the top. Rounds 5-6 re-used s_over here: on the one-image path NO barrier lies between the test of s_over above and this
    // atomic, so a wave that was late to that test could read a faster wave's maximum, take the tile for oversize and LEAVE — its share of the image's stores
    // never happened (the block kept whatever it held before: an older image of the same tile, or anything), and when it owned staged buckets the others
    // copied points from LDS positions nobody had written. One step in ~15 000 of the pipelined schedule, crowded tiles only: profiles/r06_fault_hunt.txt.)
    if (crowd) atomicMax(&s_crowd, vmax);   // some voxel of the tile is crowded: the tile's largest voxel population (> M3D_LONG_ROW)
    __syncthreads();
    const uint32_t crowd_max = s_crowd;
    const bool crowded = crowd_max != 0u;
    // How crowded (round 5; rounds 2-4: one flag = eight lanes per record): the lanes that share a record of k_nn_tiles split each voxel's points, so their number follows the
    // largest voxel — 2 lanes up to 64 points, 4 up to 160, 8 beyond — and with it the records per work item (512 / lanes): an item stages the whole image whatever
    // it holds, and a tile with one 40-point voxel used to be cut into eight items of 64 records.
#ifndef M3D_CROWD_T1
#define M3D_CROWD_T1 64u
#define M3D_CROWD_T2 160u
#endif
    const uint32_t crowd_level = !crowded ? 0u : (crowd_max > M3D_CROWD_T2 ? 3u : (crowd_max > M3D_CROWD_T1 ? 2u : 1u));
    const uint32_t extra = s_cnt;
    const int sh1 = g.cb[0] + 1, sh2 = g.cb[0] + g.cb[1] + 2;
    // 6. the images: per image the list of its voxels {key, LDS position | population - 1 | staged bucket} and the copies of its points; per tile
    // (in its first image) the table sorted position - LDS position of every staged bucket
    TB_STAMP(5);
    {
        int32_t* tdelta = reinterpret_cast<int32_t*>(B.timg + (size_t)t * M3D_TILE_IMG_BYTES + M3D_TILE_IMG_DELTA);
#pragma unroll
        for (int q = 0; q < SPT; q++) if (occ[q]) tdelta[tid + 256 * q] = (int32_t)gst[q] - (int32_t)off[q];
    }
    for (uint32_t j = 0; j < n_img; j++) {
        const uint32_t image = j == 0u ? (uint32_t)t : extra + j - 1u;
        uint8_t* img = B.timg + (size_t)M3D_CHK(209, image, m3d_tiles_of(B.n) + m3d_tile_pool(m3d_tiles_of(B.n))) * M3D_TILE_IMG_BYTES;
        float4* ipts = reinterpret_cast<float4*>(img + M3D_TILE_IMG_PTS);
        if (tid == 0) s_nv = 0u;
        __syncthreads();
        // the occupied voxels of this image's buckets into its list (any order: k_nn_tiles hashes them); the sorted position of every staged point
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            if (!occ[q] || im[q] != j) continue;
            const uint32_t cx = kk[q] & ((1u << g.cb[0]) - 1u), cy = (kk[q] >> g.cb[0]) & ((1u << g.cb[1]) - 1u), cz = kk[q] >> (g.cb[0] + g.cb[1]);
            const unsigned long long cumA = ((unsigned long long)hi[q].y << 32) | hi[q].x, cumB = ((unsigned long long)hi[q].w << 32) | hi[q].z;
            uint32_t c0 = 0u;
            for (int sub = 0; sub < 8; sub++) {
                const uint32_t c1 = (uint32_t)(((sub < 4) ? cumA : cumB) >> (16 * (sub & 3))) & 0xFFFFu;
                if (c1 > c0) {
                    const uint32_t vkey = (2u * cx + (uint32_t)(sub & 1)) | ((2u * cy + (uint32_t)((sub >> 1) & 1)) << sh1) | ((2u * cz + (uint32_t)(sub >> 2)) << sh2);
                    const uint32_t vi = M3D_CHK(206, atomicAdd(&s_nv, 1u), M3D_TILE_VCAP);   // (at most VCAP per image: the cut above)
                    s_vk[vi] = vkey;
                    s_vv[vi] = (off[q] + c0) | ((c1 - c0 - 1u) << 11) | ((uint32_t)(tid + 256 * q) << 22);
                }
                c0 = c1;
            }
            for (uint32_t k = 0; k < cnt[q]; k++) s_src[M3D_CHK(207, off[q] + k, M3D_TILE_PCAP)] = gst[q] + k;   // (LDS: the copy below is then one coalesced pass of the whole workgroup)
        }
        __syncthreads();
        if (j == 0u) TB_STAMP(6);
        {
            const uint32_t np = s_ip[j] & 0xFFFFu;
            const M3D_GLOBAL m3d_f32x4* gpts = (const M3D_GLOBAL m3d_f32x4*)(const void M3D_GLOBAL*)B.pts;
            M3D_GLOBAL m3d_f32x4* gout = (M3D_GLOBAL m3d_f32x4*)(void M3D_GLOBAL*)ipts;
            for (uint32_t q0 = 0; q0 < np; q0 += 1024u) {   // four independent gathers in flight per thread (vector registers: as an array of float4 structs the four
                // points went through 64 bytes of scratch per lane, 80 with the rest: the kernel's only private memory)
                const uint32_t pa = q0 + (uint32_t)tid, pb = pa + 256u, pc = pa + 512u, pd = pa + 768u;
                const uint32_t ga = pa < np ? s_src[pa] : 0u, gb = pb < np ? s_src[pb] : 0u, gc = pc < np ? s_src[pc] : 0u, gd = pd < np ? s_src[pd] : 0u;
                const m3d_f32x4 va = gpts[M3D_CHK(205, ga, B.n)], vb = gpts[M3D_CHK(205, gb, B.n)], vc = gpts[M3D_CHK(205, gc, B.n)], vd = gpts[M3D_CHK(205, gd, B.n)];
                if (pa < np) gout[pa] = va;
                if (pb < np) gout[pb] = vb;
                if (pc < np) gout[pc] = vc;
                if (pd < np) gout[pd] = vd;
            }
        }
        if (j == 0u) TB_STAMP(7);
        uint2* vlist = reinterpret_cast<uint2*>(img);
        const uint32_t nvx = s_nv;
        for (uint32_t i = (uint32_t)tid; i < nvx; i += 256u) vlist[i] = make_uint2(s_vk[i], s_vv[i]);
        if (tid == 0) B.timeta[image] = M3dTileImgMeta{ s_ip[j] & 0xFFFFu, nvx | (crowded ? 0x80000000u : 0u) };
        __syncthreads();
    }
    TB_STAMP(8);
    if (tid == 0) *H = M3dTileHdr{ extra, n_img, n_e << 16, (s_ip[0] & 0xFFFFu) | (crowd_level << 30) };
#ifdef M3D_TB_STAMPS
    if (tid == 0 && tb_slot < 4096u) g_tb_stamp[tb_slot][9] = ((unsigned long long)n_img << 32) | n_e;
#endif
}

// ---- a9: normals from the 27-voxel neighbourhood of the normal grid --------------------------------
__device__ __forceinline__ void sym3_square(const double m[6], double o[6]) {
    o[0] = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
    o[1] = m[0] * m[1] + m[1] * m[3] + m[2] * m[4];
    o[2] = m[0] * m[2] + m[1] * m[4] + m[2] * m[5];
    o[3] = m[1] * m[1] + m[3] * m[3] + m[4] * m[4];
    o[4] = m[1] * m[2] + m[3] * m[4] + m[4] * m[5];
    o[5] = m[2] * m[2] + m[4] * m[4] + m[5] * m[5];
}
__device__ __forceinline__ double sym3_maxabs(const double m[6]) {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) { double v = fabs(m[i]); if (v > a) a = v; }
    return a;
}
__device__ __forceinline__ double pow2_recip(double x) {   // spec §Normals v2: 2^-e, e = binary exponent of x (positive, normal): an exact scale factor
    const unsigned long long b = (unsigned long long)(2046u - (unsigned)((((unsigned long long)__double_as_longlong(x)) >> 52) & 0x7FFull)) << 52;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ double det_rsqrt(double x) {   // spec: bit-trick seed + 5 Newton steps, no sqrt
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    b = 0x5FE6EB50C7B537A9ull - (b >> 1);
    double y = __longlong_as_double((long long)b);
    const double hx = 0.5 * x;
#pragma unroll
    for (int i = 0; i < 5; i++) y = y * (1.5 - hx * y * y);
    return y;
}

// Spec §Normals (v2): positions quantised to 1/65536 voxel inside their voxel, exact int64 moments.
// Pass 1: per-voxel moments {n, S[3], P[6]} by a segmented wave64 scan over the sorted points (a run of
// equal keys is a voxel), one set of 64-bit integer atomics per run and wave, stored at the voxel's
// first sorted position.
#define M3D_NQ 65536
__device__ __forceinline__ int m3d_quant_frac(float v, float mn, float inv_leaf) {
    const float sv = (v - mn) * inv_leaf;
    const float fr = sv - floorf(sv);
    return (int)rintf(fr * 65536.0f);
}

// Pass 1: the moments {n, S[3], P[6]} of every run of equal normal-grid voxels in the finest level's sorted order, into the voxel's slot. Device-scope
// atomics are slow on this chip (ten 64-bit adds into one line per run: 57 us for 8 clouds when every run used them), so a run that is the voxel's ONLY one
// (no head lost the race for the slot in k_finalize_level: the counter behind the keys) and lies inside one wave's 256 positions STORES its sums; only
// runs cut by a wave's edge and the runs of voxels that have several are added with atomics (any grouping of a voxel's points gives the same integers).
// A thread walks NRM_PPT consecutive positions: pieces that begin and end inside it are whole runs; its first piece, when a second follows, is handed to
// the thread before it (whose last piece it continues, or which then knows it is a run of its own); the threads' last pieces are merged by a segmented wave64
// scan — a quarter of the cross-lane traffic of a position per lane.
#define NRM_PPT 4
__device__ __forceinline__ void nrm_flush(const M3dBuild& G, uint32_t key, const long long (&v)[10], bool whole) {
    const uint32_t h = nrm_slot_of(G.nkeys, G.ncap - 1u, G.nshift, key);
    long long* m = &G.mom[10 * (size_t)h];
    if (whole && G.nkeys[G.ncap + h] == 0u) {
#pragma unroll
        for (int i = 0; i < 10; i++) m[i] = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < 10; i++) atomicAdd(reinterpret_cast<unsigned long long*>(&m[i]), (unsigned long long)v[i]);
    }
}
// the moments part of k_post_finalize: this thread's NRM_PPT consecutive positions j0 ... of the finest level L (points p, finite ones first: nv of them) into the
// table of the cloud's normal grid G. Every lane of the wave calls it.
__device__ __forceinline__ void nrm_moments_part(const M3dBuild& G, const M3dBuild& L, const int j0, const float4 (&p)[NRM_PPT], const int nv) {
    const M3dGrid& g = G.grid;
    const int lane = threadIdx.x & 63;
    // the voxels of the positions next to the wave's 256: does its first run begin here, does its last run end here?
    uint32_t k_left = M3D_INVALID_KEY, k_right = M3D_INVALID_KEY;
    if (lane == 0 && j0 > 0 && j0 - 1 < nv) { const float4 q = L.pts[j0 - 1]; k_left = nrm_voxel_key(g, q.x, q.y, q.z); }
    if (lane == 63 && j0 + NRM_PPT < nv) { const float4 q = L.pts[j0 + NRM_PPT]; k_right = nrm_voxel_key(g, q.x, q.y, q.z); }
    uint32_t key = M3D_INVALID_KEY;   // voxel of the piece being summed (the thread's last, in the end); M3D_INVALID_KEY: none yet / past the end
    uint32_t kf = M3D_INVALID_KEY;    // voxel of the thread's FIRST piece when a second one followed
    bool multi = false;
    long long v[10], vF[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { v[i] = 0; vF[i] = 0; }
#pragma unroll
    for (int u = 0; u < NRM_PPT; u++) {
        if (j0 + u >= nv) break;
        const uint32_t ku = nrm_voxel_key(g, p[u].x, p[u].y, p[u].z);
        if (ku != key) {
            if (key != M3D_INVALID_KEY) {
                if (!multi) {   // the thread's first piece ends here: kept for the hand-over below
                    multi = true; kf = key;
#pragma unroll
                    for (int i = 0; i < 10; i++) vF[i] = v[i];
                } else nrm_flush(G, key, v, true);   // a piece between two boundaries inside the thread: a whole run
            }
            key = ku;
#pragma unroll
            for (int i = 0; i < 10; i++) v[i] = 0;
        }
        const long long qx = m3d_quant_frac(p[u].x, g.mn[0], g.inv_leaf), qy = m3d_quant_frac(p[u].y, g.mn[1], g.inv_leaf),
                        qz = m3d_quant_frac(p[u].z, g.mn[2], g.inv_leaf);
        v[0] += 1; v[1] += qx; v[2] += qy; v[3] += qz;
        v[4] += qx * qx; v[5] += qx * qy; v[6] += qx * qz; v[7] += qy * qy; v[8] += qy * qz; v[9] += qz * qz;
    }
    const bool ok = key != M3D_INVALID_KEY;
    if (!multi) kf = key;   // (one piece: first == last)
    // hand-over: lane t + 1's first piece (multi) continues lane t's last piece when the voxels agree — lane t takes its sums; otherwise lane t + 1 flushes it
    // itself, as a whole run (it begins where lane t's piece of another voxel ends) unless it is the wave's first lane and the run began before the wave
    const uint32_t kl_prev = (uint32_t)__shfl_up((int)key, 1);        // last voxel of the lane before (lane 0: unused)
    const uint32_t kf_next = (uint32_t)__shfl_down((int)kf, 1);       // first voxel of the lane after (lane 63: unused)
    const bool multi_next = __shfl_down((int)multi, 1) != 0;
    if (__ballot(multi) != 0ull) {   // (wave-uniform; nearly always some lane has a boundary inside)
        const bool take_next = lane < 63 && multi_next && ok && kf_next == key;
#pragma unroll
        for (int i = 0; i < 10; i++) { const long long t = __shfl_down(vF[i], 1); if (take_next) v[i] += t; }
        if (multi) {
            const bool given = lane > 0 && kl_prev == kf;
            if (!given) nrm_flush(G, kf, vF, lane > 0 || k_left != kf);
        }
    }
    // the threads' last pieces: a run continues from lane t - 1 into lane t when lane t has ONE piece of the same voxel
    const bool head = (lane == 0) || multi || (kl_prev != key);
    const bool left_whole = lane == 0 ? (multi || k_left != key) : true;   // at a head lane: does the run begin inside the wave?
    const unsigned long long heads = __ballot(head);
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int seg_start = 63 - __clzll((long long)(heads & le));
    const bool tail = (lane == 63) || ((heads >> (lane + 1)) & 1ull);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const bool take = (lane - o) >= seg_start;
        if (__ballot(take) == 0ull) break;   // wave-uniform: every run of this wave is already complete
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const long long t = __shfl_up(v[i], o);
            if (take) v[i] += t;
        }
    }
    const bool lw = __shfl((int)left_whole, seg_start) != 0;
    // (a tail lane's run ends inside the wave when the next lane starts another voxel — or took over nothing of it: the next lane's FIRST voxel decides)
    const bool right_whole = lane == 63 ? (k_right != key) : (kf_next != key || multi_next);
    if (ok && tail) nrm_flush(G, key, v, lw && right_whole);
}

// ---- behind k_finalize_level: ONE pass over every level's sorted points (round 5; three launches before, each re-reading the points) ------------------
// A thread takes NRM_PPT = 4 consecutive sorted positions (its 64 bytes of points and five keys are loaded once) and
//   * bucket table: the last point of every VOXEL writes the cumulative population of its voxel and of the empty voxels that follow it inside the bucket
//     (leading empty voxels keep the cleared value 0); the last point of a BUCKET writes the bucket's population. A bucket of more than 65535 points keeps
//     32-bit cumulative rows (k_finalize_level's bucket head saw it in the sorted keys and left 1 + row in the entry: no second pass);
//   * chunk boxes: the exact AABB of every M3D_CHUNK = 16 consecutive finite points = four threads (min / max of floats: exact, order-independent);
//   * normals: the moments of the normal-grid voxels' runs (nrm_moments_part), on the level that feeds the cloud's normal grid.
__global__ __launch_bounds__(256) void k_post_finalize(const M3dBuild* __restrict__ builds, int rows, int bpr) {
    static_assert(NRM_PPT == 4 && M3D_CHUNK == 16, "a chunk = four threads' positions");
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    const int n = B.n;
    if (rb.blk * 256 * NRM_PPT >= n) return;   // (block-uniform; n = 0: a build the pipeline skips)
    const int nv = B.grid.n_valid;
    const int j0 = (rb.blk * 256 + (int)threadIdx.x) * NRM_PPT;
    const uint32_t* skey = B.skey_out;
    float4 p[NRM_PPT];
    uint32_t k[NRM_PPT + 1];
#pragma unroll
    for (int u = 0; u < NRM_PPT; u++) p[u] = (j0 + u < n) ? B.pts[j0 + u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u <= NRM_PPT; u++) k[u] = (j0 + u < n) ? skey[j0 + u] : M3D_INVALID_KEY;
    if (B.htab) {   // (block-uniform) the bucket table's populations
        const uint32_t hmask = B.dyn[1];
        const int hshift = (int)B.dyn[2];
        const uint4* tab = reinterpret_cast<const uint4*>(B.htab);
        // the (up to four) voxel ends among this thread's positions: their table probes first, all in flight together, then the stores
        bool last[NRM_PPT]; uint32_t bk[NRM_PPT], hs[NRM_PPT]; uint4 lo[NRM_PPT];
#pragma unroll
        for (int u = 0; u < NRM_PPT; u++) {
            last[u] = k[u] != M3D_INVALID_KEY && k[u + 1] != k[u];
            bk[u] = 0u; hs[u] = 0u; lo[u] = make_uint4(0u, 0u, 0u, 0u);
            if (last[u]) { bk[u] = bucket_key_of_point(B.grid, p[u]); hs[u] = m3d_hash_slot(bk[u], hshift); lo[u] = tab[2 * (size_t)hs[u]]; }
        }
#pragma unroll
        for (int u = 0; u < NRM_PPT; u++) {
            if (!last[u]) continue;
            while (lo[u].x != bk[u]) { hs[u] = (hs[u] + 1u) & hmask; lo[u] = tab[2 * (size_t)hs[u]]; }   // (the bucket exists; a collision is rare)
            const uint32_t ku = k[u], kn = k[u + 1], h = hs[u];
            const bool same_bucket = (kn != M3D_INVALID_KEY) && ((kn >> 3) == (ku >> 3));
            const int s0 = (int)(ku & 7u), s1 = same_bucket ? (int)(kn & 7u) : 8;
            const uint32_t v = (uint32_t)(j0 + u) - lo[u].y + 1u;       // lo = {key, first sorted position, -, big: 0 or 1 + row (k_finalize_level)}
            if (lo[u].w == 0u) { for (int t = s0; t < s1; t++) B.htab[h].cum[t] = (uint16_t)v; }
            else { const uint32_t row = lo[u].w - 1u; if (row < B.bigcap) for (int t = s0; t < s1; t++) B.bigcum[8 * (size_t)row + t] = v; }
            if (!same_bucket) B.htab[h].count = v;                       // last point of the bucket
        }
    }
    if (B.cbox) {   // (block-uniform) chunk boxes: this thread's quarter of chunk j0 / 16, merged over its four lanes (every lane of the wave shuffles)
        const float inf = __uint_as_float(0x7F800000u);
        float mnx = inf, mny = inf, mnz = inf, mxx = -inf, mxy = -inf, mxz = -inf;
#pragma unroll
        for (int u = 0; u < NRM_PPT; u++) {
            if (j0 + u < nv) {
                mnx = fminf(mnx, p[u].x); mny = fminf(mny, p[u].y); mnz = fminf(mnz, p[u].z);
                mxx = fmaxf(mxx, p[u].x); mxy = fmaxf(mxy, p[u].y); mxz = fmaxf(mxz, p[u].z);
            }
        }
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, o)); mny = fminf(mny, __shfl_xor(mny, o)); mnz = fminf(mnz, __shfl_xor(mnz, o));
            mxx = fmaxf(mxx, __shfl_xor(mxx, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o));
        }
        const int c = j0 / M3D_CHUNK;
        if ((threadIdx.x & 3u) == 0u && c * M3D_CHUNK < nv) {
            B.cbox[2 * c] = make_float4(mnx, mny, mnz, 0.f);
            B.cbox[2 * c + 1] = make_float4(mxx, mxy, mxz, 0.f);
        }
    }
    if (B.nrm_feed) nrm_moments_part(builds[B.nrm_build], B, j0, p, nv);   // (block-uniform)
}

// Pass 2: once per occupied VOXEL (every point of a voxel sees the same 27 voxels, hence the same sums and the same normal): add the (shifted)
// moments of the 27 voxels around it and take the smallest eigenvector of the covariance. Work is dealt BY VOXEL: every workgroup scans the per-block
// counts k_finalize_level left (a few hundred words) into LDS, the voxels are then numbered across the blocks' lists and taken 32 at a time, grid-stride
// (a workgroup per group of blocks, the first shape, ran 93 us for 8 clouds: a far, sparse stretch of a scan has a voxel per point, a near one two per
// thousand points). Eight lanes per voxel for the sums (27 table probes shared out), then one lane per voxel solves.
#define NRM_PREF_CAP 8192   // blocks whose counts one prefix covers (2 M points); larger clouds are taken in stretches of that many blocks
__device__ __forceinline__ void nrm_solve_role(const M3dBuild* __restrict__ builds, int grids_per_cloud, float plane_ratio, int min_pts, float min_spread, int bpr, int bid) {
    const M3dRB rb = m3d_row_block(0, bpr, bid);
    const M3dBuild& G = builds[rb.row * grids_per_cloud];
    if (!G.nkeys) return;
    const M3dBuild& L = builds[rb.row * grids_per_cloud + grids_per_cloud - 1];
    const int nblk = (L.n + 255) / 256;   // blocks of k_finalize_level that wrote their counts
    if (nblk == 0) return;
    // two levels: the prefix of the counts of every SIXTEEN blocks in LDS (2 KB: the kernel's occupancy is the double-precision solve's registers, not this),
    // the sixteen counts themselves re-read (64 bytes, cached) by the lanes that look a voxel up
    __shared__ uint32_t s_pref[NRM_PREF_CAP / 16 + 1];
    __shared__ uint32_t s_w[4];
    __shared__ long long s_sum[32][10];
    __shared__ uint32_t s_vs[32];
    const int tid = (int)threadIdx.x;
    const uint32_t* nlist = G.nlist;
    const uint32_t* nvcnt = G.nvcnt;
    for (int sb0 = 0; sb0 < nblk; sb0 += NRM_PREF_CAP) {   // (one stretch for clouds of up to 2 M points)
    const int sbn = min(nblk - sb0, NRM_PREF_CAP), sgn = (sbn + 15) / 16;
    __syncthreads();   // (the previous stretch's prefix is no longer read)
    uint32_t run = 0u;
    for (int base = 0; base < sgn; base += 256) {   // exclusive prefix over the groups of sixteen blocks
        const int gq = base + tid;
        uint32_t c = 0u;
        if (gq < sgn) {
            const uint4* c4 = reinterpret_cast<const uint4*>(nvcnt + sb0 + 16 * gq);   // (256-byte aligned array, sb0 a multiple of 16; the array is padded to whole groups)
            const uint4 a = c4[0], b = c4[1], d = c4[2], e = c4[3];
            const uint32_t cc[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w, e.x, e.y, e.z, e.w };
#pragma unroll
            for (int q = 0; q < 16; q++) c += (16 * gq + q < sbn) ? cc[q] : 0u;   // (the words behind the last block's count are padding: never written)
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(c, s_w, tot);
        if (gq < sgn) s_pref[gq] = run + ex;
        run += tot;
    }
    if (tid == 0) s_pref[sgn] = run;
    __syncthreads();
    const uint32_t nV = s_pref[sgn];
    const M3dGrid& g = G.grid;
    const int sh1 = g.cb[0] + 1, sh2 = g.cb[0] + g.cb[1] + 2;
    const uint32_t nmask = G.ncap - 1u;
    const int nshift = G.nshift;
    const uint32_t* nkeys = G.nkeys;
    const long long* mom = G.mom;
    const int li = tid >> 3, part = tid & 7;
    for (uint32_t vb = (uint32_t)rb.blk * 32u; vb < nV; vb += (uint32_t)bpr * 32u) {   // (block-uniform: barriers and shuffles inside)
        const uint32_t v = vb + (uint32_t)li;
        const bool act = v < nV;
        uint32_t slot = 0u;
        if (act) {   // voxel v of the stretch: the group of sixteen blocks that holds it (bisection of the prefix in LDS), the block inside the group, its place in that block's list
            int lo = 0, hi = sgn;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[mid] <= v) lo = mid; else hi = mid; }
            const uint4* c4 = reinterpret_cast<const uint4*>(nvcnt + sb0 + 16 * lo);
            const uint4 a = c4[0], b = c4[1], d = c4[2], e = c4[3];
            const uint32_t cc[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w, e.x, e.y, e.z, e.w };
            uint32_t r = v - s_pref[lo], bsel = 0u, before = 0u, acc = 0u;
#pragma unroll
            for (int q = 0; q < 16; q++) { if (acc <= r) { bsel = (uint32_t)q; before = acc; } acc += (16 * lo + q < sbn) ? cc[q] : 0u; }   // the last block whose prefix is <= r (empty blocks in between are skipped: their prefix equals their successor's)
            slot = nlist[(size_t)(sb0 + 16 * lo + (int)bsel) * 256u + (r - before)];
        }
        const uint32_t key = act ? nkeys[slot] : 0u;
        const int icx = (int)(key & ((1u << sh1) - 1u)), icy = (int)((key >> sh1) & ((1u << (g.cb[1] + 1)) - 1u)), icz = (int)(key >> sh2);
        long long k = 0, s0 = 0, s1 = 0, s2 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
        // this lane's neighbours: part, part + 8, part + 16, part + 24 of the 27; their probes first (independent), then the moments of the ones that exist
        uint32_t hh[4], nkq[4], kq[4]; bool fnd[4]; int ddx[4], ddy[4], ddz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int nb = part + 8 * u;
            ddx[u] = nb % 3 - 1; ddy[u] = (nb / 3) % 3 - 1; ddz[u] = nb / 9 - 1;
            const int x = icx + ddx[u], y = icy + ddy[u], z = icz + ddz[u];
            fnd[u] = act && nb < 27 && x >= 0 && y >= 0 && z >= 0 && x < g.dims[0] && y < g.dims[1] && z < g.dims[2];
            nkq[u] = (uint32_t)x | ((uint32_t)y << sh1) | ((uint32_t)z << sh2);
            hh[u] = (nkq[u] * 0x9E3779B1u) >> nshift;
            kq[u] = fnd[u] ? nkeys[hh[u]] : M3D_INVALID_KEY;   // the four first probes are in flight together
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {   // (linear probing past the rare collision)
            while (kq[u] != nkq[u] && kq[u] != M3D_INVALID_KEY) { hh[u] = (hh[u] + 1u) & nmask; kq[u] = nkeys[hh[u]]; }
            fnd[u] = fnd[u] && kq[u] == nkq[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!fnd[u]) continue;
            const long long* m = &mom[10 * (size_t)hh[u]];
            const long long n = m[0], Sx = m[1], Sy = m[2], Sz = m[3];
            const long long Dx = (long long)ddx[u] * M3D_NQ, Dy = (long long)ddy[u] * M3D_NQ, Dz = (long long)ddz[u] * M3D_NQ;
            k += n;
            s0 += Sx + n * Dx; s1 += Sy + n * Dy; s2 += Sz + n * Dz;
            q0 += m[4] + 2 * Dx * Sx + n * Dx * Dx;
            q1 += m[5] + Dx * Sy + Dy * Sx + n * Dx * Dy;
            q2 += m[6] + Dx * Sz + Dz * Sx + n * Dx * Dz;
            q3 += m[7] + 2 * Dy * Sy + n * Dy * Dy;
            q4 += m[8] + Dy * Sz + Dz * Sy + n * Dy * Dz;
            q5 += m[9] + 2 * Dz * Sz + n * Dz * Dz;
        }
        {   // lane 8g collects the sums of its group (every lane of the wave shuffles)
            long long* acc[10] = { &k, &s0, &s1, &s2, &q0, &q1, &q2, &q3, &q4, &q5 };
#pragma unroll
            for (int a = 0; a < 10; a++) {
#pragma unroll
                for (int o = 4; o >= 1; o >>= 1) *acc[a] += __shfl_down(*acc[a], o);
            }
        }
        if (part == 0) {
            s_sum[li][0] = k; s_sum[li][1] = s0; s_sum[li][2] = s1; s_sum[li][3] = s2; s_sum[li][4] = q0;
            s_sum[li][5] = q1; s_sum[li][6] = q2; s_sum[li][7] = q3; s_sum[li][8] = q4; s_sum[li][9] = q5;
            s_vs[li] = slot;
        }
        __syncthreads();
        if (tid < 32 && vb + (uint32_t)tid < nV) {
            const long long k = s_sum[tid][0], s0 = s_sum[tid][1], s1 = s_sum[tid][2], s2 = s_sum[tid][3], q0 = s_sum[tid][4],
                            q1 = s_sum[tid][5], q2 = s_sum[tid][6], q3 = s_sum[tid][7], q4 = s_sum[tid][8], q5 = s_sum[tid][9];
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
            do {
                if (k < (long long)min_pts || k < 3) break;
                const double inv = 1.0 / (double)k;
                const double m0 = (double)s0 * inv, m1 = (double)s1 * inv, m2 = (double)s2 * inv;
                double c[6] = { (double)q0 * inv - m0 * m0, (double)q1 * inv - m0 * m1, (double)q2 * inv - m0 * m2,
                                (double)q3 * inv - m1 * m1, (double)q4 * inv - m1 * m2, (double)q5 * inv - m2 * m2 };
                const double cm = sym3_maxabs(c);
                if (!(cm > 0.0)) break;
                const double sc = pow2_recip(cm);
#pragma unroll
                for (int i = 0; i < 6; i++) c[i] = c[i] * sc;
                const double a[6] = { c[3] * c[5] - c[4] * c[4], c[2] * c[4] - c[1] * c[5], c[1] * c[4] - c[2] * c[3],
                                      c[0] * c[5] - c[2] * c[2], c[1] * c[2] - c[0] * c[4], c[0] * c[3] - c[1] * c[1] };
                const double am = sym3_maxabs(a);
                if (!(am > 1e-12)) break;
                double p[6], t2[6];
                {
                    const double sa = pow2_recip(am);
#pragma unroll
                    for (int i = 0; i < 6; i++) p[i] = a[i] * sa;
                }
                for (int it = 0; it < 5; it++) {
                    sym3_square(p, t2);
                    const double tm = sym3_maxabs(t2);
                    const double st = pow2_recip(tm);
#pragma unroll
                    for (int i = 0; i < 6; i++) p[i] = t2[i] * st;
                }
                double v0, v1, v2;
                if (p[0] >= p[3] && p[0] >= p[5]) { v0 = p[0]; v1 = p[1]; v2 = p[2]; }
                else if (p[3] >= p[5]) { v0 = p[1]; v1 = p[3]; v2 = p[4]; }
                else { v0 = p[2]; v1 = p[4]; v2 = p[5]; }
                const double nn = v0 * v0 + v1 * v1 + v2 * v2;
                if (!(nn > 0.0)) break;
                const double rn = det_rsqrt(nn);
                v0 = v0 * rn; v1 = v1 * rn; v2 = v2 * rn;
                const double cv0 = c[0] * v0 + c[1] * v1 + c[2] * v2, cv1 = c[1] * v0 + c[3] * v1 + c[4] * v2, cv2 = c[2] * v0 + c[4] * v1 + c[5] * v2;
                double l3 = v0 * cv0 + v1 * cv1 + v2 * cv2;
                const double av0 = a[0] * v0 + a[1] * v1 + a[2] * v2, av1 = a[1] * v0 + a[3] * v1 + a[4] * v2, av2 = a[2] * v0 + a[4] * v1 + a[5] * v2;
                const double pr = v0 * av0 + v1 * av1 + v2 * av2;
                const double sm = (c[0] + c[3] + c[5]) - l3;
                if (l3 < 0.0) l3 = 0.0;
                const double mth = l3 / (double)plane_ratio;
                if (!((mth <= 0.5 * sm) && ((mth * mth - sm * mth) + pr >= 0.0))) break;
                const double spread_q = (double)min_spread * 65536.0;
                const double mw = (spread_q * spread_q) * sc;   // the threshold in the units c was scaled to
                if (!((mw <= 0.5 * sm) && ((mw * mw - sm * mw) + pr >= 0.0))) break;
                int im = 0;
                double vm = fabs(v0);
                if (fabs(v1) > vm) { im = 1; vm = fabs(v1); }
                if (fabs(v2) > vm) { im = 2; }
                const double lead = im == 0 ? v0 : (im == 1 ? v1 : v2);
                if (lead < 0.0) { v0 = -v0; v1 = -v1; v2 = -v2; }
                out = make_float4((float)v0, (float)v1, (float)v2, 0.f);
            } while (0);
            G.nnrm[s_vs[tid]] = out;
        }
        __syncthreads();   // (the sums are overwritten by the next trip)
    }
    }
}

// The tile images (tile_build_role) and the normals' solve (nrm_solve_role) both follow k_post_finalize and touch nothing of each other's: ONE launch, the first
// n_tile_blocks workgroups build tiles, the rest solve voxels. Alone on the GPU either is a latency chain that leaves most of the chip idle (44 and 39 us for 16
// clouds): side by side they take what the longer one takes. (Two streams do the same on paper; measured, profiles/r05_side_stream.txt, the fork / join cost
// what the overlap saved and the headline lost 11 %.)
__global__ __launch_bounds__(256) void k_tiles_normals(const M3dBuild* __restrict__ builds, int row_stride, int row_first, int tile_bpr, int sliced, int n_tile_blocks,
                                                       int grids_per_cloud, float plane_ratio, int min_pts, float min_spread, int solve_bpr) {
    if ((int)blockIdx.x < n_tile_blocks) tile_build_role(builds, row_stride, row_first, tile_bpr, sliced, (int)blockIdx.x);
    else nrm_solve_role(builds, grids_per_cloud, plane_ratio, min_pts, min_spread, solve_bpr, (int)blockIdx.x - n_tile_blocks);
}

// Pass 3: every point takes the normal of its normal-grid voxel — in every level's sorted order, one coalesced pass per level (one table probe per
// point; neighbours share their voxel's slot). Non-finite points (sorted last) get {0,0,0,0}.
__global__ __launch_bounds__(256) void k_nrm_handout(const M3dBuild* __restrict__ builds, int rows, int bpr) {
    const M3dRB rb = m3d_row_block(rows, bpr);
    const M3dBuild& B = builds[m3d_row_build(rb.row, rows)];
    if (!B.nrm_sorted || B.nrm_build < 0) return;
    const int j = rb.blk * (int)blockDim.x + (int)threadIdx.x;
    if (j >= B.n) return;
    const M3dBuild& G = builds[B.nrm_build];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < B.grid.n_valid) {
        const float4 p = B.pts[j];
        o = G.nnrm[nrm_slot_of(G.nkeys, G.ncap - 1u, G.nshift, nrm_voxel_key(G.grid, p.x, p.y, p.z))];
    }
    B.nrm_sorted[j] = o;
}

// ---- export helpers (introspection API) -----------------------------------------------------------
__global__ void k_export_sorted(const float4* __restrict__ pts, const float4* __restrict__ nrm, int n, float* __restrict__ xyz,
                                float* __restrict__ nxyz) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float4 p = pts[j];
    xyz[3 * j] = p.x; xyz[3 * j + 1] = p.y; xyz[3 * j + 2] = p.z;
    if (nrm && nxyz) { const float4 q = nrm[j]; nxyz[3 * j] = q.x; nxyz[3 * j + 1] = q.y; nxyz[3 * j + 2] = q.z; }
}

// ---- host-side launchers ----------------------------------------------------------------------------
#define HIP_TRY(e) do { hipError_t _e = (e); if (_e != hipSuccess) return _e; } while (0)

int m3d_sort_tiles(int n) { return (n + RS_TILE - 1) / RS_TILE; }

hipError_t m3d_launch_decode_aabb(hipStream_t s, const M3dDecode* d_descs, int n_clouds, int max_n) {
    int blocks = (max_n + 1023) / 1024;   // ~4 points per thread, one atomic set per block
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_decode_aabb, dim3(blocks * n_clouds), dim3(256), 0, s, d_descs, n_clouds, blocks);
    M3D_DBG(s, "k_decode_aabb");
    return hipGetLastError();
}

// the whole bucketing pipeline of n_builds grids (dyn counters must be zeroed by the caller)
hipError_t m3d_launch_bucket_batch(hipStream_t s, M3dBuild* d_builds, int n_clouds, int grids_per_cloud, int max_n, bool any_normals,
                                   int any_tiles, float plane_ratio, int min_pts, float min_spread, bool pyramid) {
    const int lv = grids_per_cloud - (any_normals ? 1 : 0);   // level builds per cloud: the rows of the batched launches (m3d_row_build)
    const int n_rows = n_clouds * lv, rows = lv | (grids_per_cloud << 16);
    const int max_passes = 4;   // a build whose keys need fewer skips the later ones on the device
    hipLaunchKernelGGL(k_grid_params, dim3((n_clouds + 63) / 64), dim3(64), 0, s, d_builds, n_clouds, grids_per_cloud);
    M3D_DBG(s, "k_grid_params");
    const int blocks = (max_n + 255) / 256;
    const int ntiles = m3d_sort_tiles(max_n);
    const int cb = blocks > 256 ? 256 : blocks;
    hipLaunchKernelGGL(k_voxel_keys, dim3(blocks * n_rows), dim3(256), 0, s, d_builds, rows, blocks);
    M3D_DBG(s, "k_voxel_keys");
    for (int phase = 0; phase < (pyramid ? 2 : 1); phase++) {
        if (phase == 1) {   // the coarser levels of pyramids: keyed in their cloud's finest-level order, then sorted (stable) by their own keys
            hipLaunchKernelGGL(k_rekey, dim3(blocks * n_rows), dim3(256), 0, s, d_builds, rows, blocks);
            M3D_DBG(s, "k_rekey");
        }
        for (int pass = 0; pass < max_passes; pass++) {
            const int fused = ntiles <= RS_FUSED_TILES ? 1 : 0;   // (ntiles = the batch's largest cloud)
            hipLaunchKernelGGL(k_rs_hist, dim3(ntiles * n_rows), dim3(RS_THREADS), 0, s, d_builds, pass, fused, phase, rows, ntiles);
            M3D_DBG(s, "k_rs_hist");
            if (!fused) {
                hipLaunchKernelGGL(k_rs_scan_sums, dim3(RS_SCAN_CHUNKS, n_rows), dim3(256), 0, s, d_builds, pass, phase, rows);
                hipLaunchKernelGGL(k_rs_scan_apply, dim3(RS_SCAN_CHUNKS, n_rows), dim3(256), 0, s, d_builds, pass, phase, rows);
                M3D_DBG(s, "k_rs_scan");
            }
            hipLaunchKernelGGL(k_rs_scatter, dim3(m3d_sliced_grid(ntiles) * n_rows), dim3(RS_THREADS), 0, s, d_builds, pass, fused, phase, rows, ntiles, (m3d_sliced_on() >> 1) & 1);
            M3D_DBG(s, "k_rs_scatter");
        }
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_count_cells, dim3(blocks * n_rows), dim3(256), 0, s, d_builds, rows, blocks);
    M3D_DBG(s, "k_count_cells");
    hipLaunchKernelGGL(k_table_params, dim3(n_rows), dim3(256), 0, s, d_builds, rows);
    M3D_DBG(s, "k_table_params");
    hipLaunchKernelGGL(k_clear_table, dim3(cb * n_rows), dim3(256), 0, s, d_builds, rows, cb);
    M3D_DBG(s, "k_clear_table");
    hipLaunchKernelGGL(k_finalize_level, dim3(m3d_sliced_grid(blocks) * n_rows), dim3(256), 0, s, d_builds, rows, blocks, m3d_sliced_on() & 1);
    M3D_DBG(s, "k_finalize_level");
    const int nb_p = (blocks + NRM_PPT - 1) / NRM_PPT;
    hipLaunchKernelGGL(k_post_finalize, dim3(nb_p * n_rows), dim3(256), 0, s, d_builds, rows, nb_p);   // table populations, chunk boxes, normal-grid moments: one pass over the sorted points
    M3D_DBG(s, "k_post_finalize");
    {
        const int tile_bpr = m3d_tiles_of(max_n), n_tile_blocks = any_tiles ? m3d_sliced_grid(tile_bpr) * n_clouds : 0;
        const int nb_s = std::min(blocks, 256), n_solve_blocks = any_normals ? nb_s * n_clouds : 0;   // voxels are taken 32 at a time, grid-stride (their number is only known on the device)
        if (n_tile_blocks + n_solve_blocks > 0) {
            hipLaunchKernelGGL(k_tiles_normals, dim3(n_tile_blocks + n_solve_blocks), dim3(256), 0, s, d_builds, grids_per_cloud, grids_per_cloud - 1, tile_bpr, (m3d_sliced_on() >> 2) & 1,
                               n_tile_blocks, grids_per_cloud, plane_ratio, min_pts, min_spread, nb_s);
            M3D_DBG(s, "k_tiles_normals");
        }
    }
    if (any_normals) {
        hipLaunchKernelGGL(k_nrm_handout, dim3(blocks * n_rows), dim3(256), 0, s, d_builds, rows, blocks);
        M3D_DBG(s, "k_nrm_handout");
    }
    return hipGetLastError();
}

hipError_t m3d_launch_export_sorted(hipStream_t s, const float4* pts, const float4* nrm, int n, float* xyz, float* nxyz) {
    hipLaunchKernelGGL(k_export_sorted, dim3((n + 255) / 256), dim3(256), 0, s, pts, nrm, n, xyz, nxyz);
    M3D_DBG(s, "k_export_sorted");
    return hipGetLastError();
}
