// m3d_device.h — device-visible data layout of libm3dreg (gfx950 only; no CUDA/dual path).
//
// HBM layout of one bucketed cloud level (DESIGN.md §Data layout):
//   pts   float4[n]   points sorted by voxel key {x, y, z, bits(input_index)}: one
//                     16-B gather per candidate. Voxels are grouped in 2x2x2 BUCKETS; buckets follow a
//                     compact Morton curve, the 8 voxels of a bucket are consecutive runs.
//   htab  M3dBucket[T] open-addressing hash of occupied buckets, 32 B per entry (two 16-B halves):
//                     {bucket key, first sorted position, population, big index} + 8 cumulative
//                     per-voxel populations (uint16). T = smallest power of two >= 2 * occupied buckets,
//                     derived on the device after the sort (no host round trip).
//                     A query's 27 voxels live in at most 2x2x2 buckets: 8 probes instead of 27.
//   nrm   float4[n]   unit normals in the level's sorted order ({0,0,0,0} = no usable normal), point-to-plane only
// The source side of a registration streams the source cloud's own sorted float4 array (16 B per lane,
// 1 KiB per wave instruction: fully coalesced), so the 64 lanes of a wave hold spatially adjacent queries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define M3D_NSUMS 29
#define M3D_PARTIAL_STRIDE 32      // int64 words per block partial of the reduction pass (29 used)
#define M3D_INVALID_KEY 0xFFFFFFFFu
#define M3D_IDX_MASK 0x0FFFFFFFu   // pts[].w = bits of the input index (at most 2^28 - 1 points per cloud)
#define M3D_MAX_TRACE 256
#define M3D_CHUNK 16               // sorted points per chunk box
#define M3D_LONG_ROW 24            // a row part with more candidates than this is walked chunk by chunk, boxes first

struct M3dGrid {           // geometry of one voxel grid (host computes it from the exact AABB)
    float mn[3];
    float inv_leaf;
    float center[3];
    float leaf;
    int32_t dims[3];       // extent in voxels
    int32_t cb[3];         // bit widths of the bucket coordinates (bucket = voxel >> 1)
    int32_t hshift;        // hash slot = (bucket key * 0x9E3779B1u) >> hshift
    uint32_t hmask;
    int32_t n_valid;
    float prune_slack;     // absolute slack [m] subtracted from voxel-box gaps before pruning
};

// What the bucketing pipeline derives ON THE DEVICE for one grid of a cloud (k_grid_params from the exact AABB, k_table_params
// after the sort) — the host never waits for any of it: k_patch_jobs hands it to the registrations, the host reads it back
// only when somebody asks (grid_info, export, the synchronous entry points' error check). 144 bytes, in the cloud's block.
#define M3D_ERR_GRID_TOO_LARGE (-4)   // == M3DREG_ERR_GRID_TOO_LARGE
#define M3D_ERR_EMPTY_CLOUD (-5)      // == M3DREG_ERR_EMPTY_CLOUD
struct M3dLevelMeta {
    uint32_t dyn[8];       // {occupied voxels, hmask, hshift, occupied buckets, big buckets, voxel heads, -, -}
    M3dGrid g;             // hmask / hshift live in dyn[1], dyn[2] (written later in the pipeline than the rest)
    float lbound;          // max half extent + 3 leaf: bound of |u - centre| that sizes the fixed-point exponents
    float mx[3];
    int32_t err;           // 0, or the m3dreg_error of the CLOUD (same value in every level's meta)
    int32_t bits[3];
    float sumsq;           // sum over the occupied voxels of population^2 (float, fixed summation order): sumsq / n_valid = the mean population of the voxel a
                           // point lies in — what a query meets in its home voxel, the a-priori cost of a registration (m3dreg_cloud_density; LPT sharding)
    int32_t pad;
};
static_assert(sizeof(M3dLevelMeta) == 144, "M3dLevelMeta layout");

struct M3dBucket {         // 32 bytes, 32-byte aligned
    uint32_t key;          // cx | cy << cb[0] | cz << (cb[0]+cb[1]); M3D_INVALID_KEY = empty slot
    uint32_t start;        // first sorted position of the bucket's points
    uint32_t count;        // points in the bucket
    uint32_t big;          // 0, or 1 + index into the level's bigcum table when count > 65535
    uint16_t cum[8];       // cum[s] = points of the bucket in voxels 0..s (s = (ix&1)|(iy&1)<<1|(iz&1)<<2)
};

// ---- target tiles: the LDS-staged search (k_nn_tiles) ------------------------------------------------------------------
// A TILE owns the buckets whose first sorted position lies in [t * M3D_TILE_PTS, (t + 1) * M3D_TILE_PTS): consecutive buckets of the
// Morton order, i.e. one compact patch of the scanned surface. Its IMAGE, built once per target by the bucketing pipeline
// (bucket.hip: the tile workgroups of k_tiles_normals), is what a workgroup must hold in LDS to answer every query that has an occupied bucket of the tile among the
// 2x2x2 buckets of its 27-voxel neighbourhood: the tile's own buckets plus every occupied bucket within one bucket of them.
//   vlist[n_voxels]        the staged VOXELS, a compact list {voxel key, value}: value = first LDS position (11 bits) | population - 1 (11 bits) << 11 |
//                          staged-bucket number << 22; voxel key = ix | iy << (cb0 + 1) | iz << (cb0 + cb1 + 2): a neighbour's key is the query voxel's
//                          key plus a constant. k_nn_tiles hashes the list into an M3D_TILE_VS-slot open-addressing directory IN LDS while it stages
//                          (round 4; rounds 2-3 stored the 16 KB hashed directory itself: a third of an image's bytes, most of them empty slots)
//   pts[n_points]          the staged points themselves {x, y, z, bits(input index)}, in LDS order: staging an image is two coalesced
//                          streams (voxel list, points) — no gather, no index indirection on the critical path
//   delta[n_buckets]       (the tile's FIRST image only, one table for all of its images) sorted position - LDS position of the points of staged bucket b:
//                          a staged bucket is one contiguous run in both orders, so the winner's sorted position is its LDS position + delta[bucket of its
//                          voxel] — 2 KB per tile instead of one 4-byte sorted position per staged point (8 KB per image, read by a random 4-byte load per answer)
#define M3D_TILE_PTS 512
#define M3D_TILE_ECAP 512       // staged buckets per tile at most
#define M3D_TILE_VS 2048        // slots of the voxel directory
#define M3D_TILE_VCAP 1280      // staged voxels per tile at most
#define M3D_TILE_PCAP 2048      // staged points per tile at most
#define M3D_TILE_QCAP 2048      // query records a tile accepts per iteration (the rest takes the global walk)
#define M3D_TILE_OVERSIZE 1u
#define M3D_OCC_BITS 23         // the occupancy bitmap covers grids of up to 2^23 bucket positions (1 MiB per level)
#define M3D_TILE_MAXIMG 32
#define M3D_TILE_LISTS 8      // work-item lists of k_nn_tiles (one counter per list, each on its own 128-B line; list l is served by the workgroups with blockIdx & 7 == l)
struct M3dTileHdr { uint32_t extra, n_img, flags, meta0; };   // images of the tile: image t, then images extra .. extra + n_img - 2; flags = M3D_TILE_OVERSIZE | staged buckets << 16; meta0 = staged points of image t | crowd level << 30
// crowd level of a tile (0: no voxel of more than M3D_LONG_ROW points; 1 / 2 / 3: largest voxel <= 64 / <= 160 / larger): 2^level lanes share a record of k_nn_tiles, 512 >> level records make a work item
// (a SMALL batch — fewer than 768 plain items, launch_iteration: one or two pairs alone on the GPU — keeps rounds 2-4's rule, 64 records and eight lanes for every crowded tile: there the
//  items are what fills the chip, and a lone crowded pair's k_nn_tiles took 255 instead of 181 us per registration with the graded rule)
__host__ __device__ inline uint32_t m3d_tile_records_per_item(uint32_t meta0, uint32_t plain_chunk) {
    const uint32_t lvl = meta0 >> 30;
    if (lvl == 0u) return plain_chunk;
    return plain_chunk < 512u ? 64u : (512u >> lvl);
}
struct M3dTileImgMeta { uint32_t n_points, n_voxels; };   // n_voxels bit 31: the image holds a voxel of more than M3D_LONG_ROW points
static_assert(sizeof(M3dTileHdr) == 16 && sizeof(M3dTileImgMeta) == 8, "tile image layout");
__host__ __device__ inline int m3d_tile_pool(int n_tiles) { return n_tiles / 2 + 8; }   // extra images per level
#define M3D_TILE_IMG_PTS (M3D_TILE_VCAP * 8)                               // byte offsets inside an image: the voxel list first
#define M3D_TILE_IMG_DELTA (M3D_TILE_VCAP * 8 + M3D_TILE_PCAP * 16)
#define M3D_TILE_IMG_BYTES (M3D_TILE_VCAP * 8 + M3D_TILE_PCAP * 16 + M3D_TILE_ECAP * 4)   // 44 KB (rounds 2-3: 56 KB)
#define M3D_TILE_SV_POS(sv) ((sv) & 0x7FFu)                                // directory value: first LDS position ...
#define M3D_TILE_SV_CNT(sv) ((((sv) >> 11) & 0x7FFu) + 1u)                 // ... population ...
#define M3D_TILE_SV_BKT(sv) ((sv) >> 22)                                   // ... staged-bucket number
static_assert(M3D_TILE_PCAP <= 2048 && M3D_TILE_ECAP <= 512, "directory value fields");
__host__ __device__ inline int m3d_tiles_of(int n) { return (n + M3D_TILE_PTS - 1) / M3D_TILE_PTS; }

struct M3dLevelDev {       // what the NN / ICP kernels need from a target level
    const float4* pts;
    const float4* nrm;
    const M3dTileHdr* thdr;    // [m3d_tiles_of(n)] tile headers, or null (the level then has no tiles: every search walks global memory)
    const uint8_t* timg;       // [tiles + pool][M3D_TILE_IMG_BYTES] tile images
    const M3dTileImgMeta* timeta;   // [tiles + pool] staged points / voxels of every image
    const uint32_t* occ;       // one bit per bucket POSITION (bit index = bucket key): set = occupied. k_nn_iter tests the 2x2x2 buckets around a
                               // query with eight 4-byte loads from this small table instead of eight 16-byte hash probes; null when the grid
                               // has more positions than M3D_OCC_BITS (k_patch_jobs clears the pointer; the search then probes the hash table)
    const M3dBucket* htab;
    const uint32_t* bigcum;   // [n_big][8] 32-bit cumulative populations of buckets with more than 65535 points
    const float4* cbox;       // [2 * ceil(n / 16)] exact AABB {min, max} of every 16 consecutive sorted points: lets the search skip most of a
                              // crowded voxel (a surface a metre from the sensor puts a hundred points into one 10 cm voxel)
    const uint32_t* dyn;      // the level's M3dLevelMeta in the cloud's block (its first 8 words are the dyn counters): derived on the
                              // device by the bucketing pipeline; k_patch_jobs copies the geometry into g (the host never waits for it)
    M3dGrid g;
};

struct M3dPairState {      // per-registration state, lives in HBM for the whole run (no host sync per iteration)
    double T[16];                  // current pose, column-major
    long long sums[M3D_NSUMS];     // fixed-point normal-equation sums of the running iteration
    double th2, tr2;               // |omega|^2, |v|^2 of the last update
    long long n_corr, ssr;         // of the last executed iteration
    int32_t ssr_exp;
    int32_t iters;
    int32_t status;
    int32_t done;                  // final: no further iteration may run
    int32_t level_done;            // current level converged: skip its remaining iterations
    uint32_t pad0;
    uint32_t ctr[2];               // diagnostics: searches answered from LDS tiles / by the global walk of the fallback blocks
    uint32_t gsync[2];             // pair 0 only: {pairs that reported this iteration, of which still active} (k_solve_update)
};

struct M3dJob {            // one pair at one level
    const float* src;      // source points {x, y, z} PACKED (12 bytes each: a registration streams them in four kernels of every iteration and never needs the
                           // index word of the float4 array), in the source cloud's own sorted order (finest level), n_src finite points
    int32_t n_src;
    int32_t metric;
    M3dLevelDev tgt;
    float dmax2;
    float S[6];            // 2^exps
    int32_t exps[6];
    int32_t min_corr;
    int32_t last_level;
    double eps_rot2, eps_trans2, pivot_rel_tol;
    M3dPairState* st;
    double* trace;         // [M3D_MAX_TRACE][16] or null
    int32_t src_nblk;          // entries of src_order: ceil(points of the source, finite or not, / 256)
    const uint32_t* src_order; // the source's 256-point blocks, most crowded first: k_nn_iter starts its long-running blocks first (or null)
    const uint32_t* src_dyn;   // M3dLevelMeta of the source's finest level (n_src, error state): read by k_patch_jobs
    float dmax;                // max_corr_dist of this level (k_patch_jobs derives the fixed-point exponents from it and the target's lbound)
    const float4* prev_pts;    // a level behind the first of a pyramid: the PREVIOUS level's sorted points (the matches the last iteration of that level left index
                               // them): a query's old match, when it lies within one voxel edge of this level, bounds this level's first search (a seed
                               // only bounds: same result); null on the first level
    float* ring;               // [32][12] the pair's pose ring (icp.hip: m3d_cert_state): float poses {R row-major, t} of its last 32 iterations
    int32_t coop_always;       // k_patch_jobs: 1 = the target level is DENSE for this source (m3d_dense_level): every search is walked eight lanes per query,
                               // nearest chunk first (k_nn_coop), none is walked by one lane and none is binned to tiles
    int32_t pad_;
};
// A target level is dense when a query's 27 voxels hold hundreds of candidates — more than M3D_COOP_DENSITY points per occupied voxel (a coarse level of a
// pyramid) — or when it holds more than M3D_COOP_DENSITY_MAP points per voxel AND the target is a map, M3D_COOP_MAP_RATIO times the source and more: a tile
// image is then staged for two dozen queries instead of five hundred, and one lane walks a hundred candidates alone. Measured (profiles/r04_dense_levels.txt):
// config 5's 0.2 m level (35 per voxel) 0.070 -> 0.050 ms per iteration, its 0.1 m level (8 per voxel, 2 M : 100 k) 87 + 183 -> 50 us for the first
// iteration; config 2's 0.8 m level (70 k : 70 k) gains 4 %, its 0.4 m level (10 per voxel) would LOSE 5 %: hence the ratio.
#ifndef M3D_COOP_DENSITY
#define M3D_COOP_DENSITY 24
#endif
#define M3D_COOP_DENSITY_MAP 6
#define M3D_COOP_MAP_RATIO 4
// ... or more than M3D_COOP_DENSITY_LATER points per voxel on a level that starts from a coarser level's result: nearly every query is certified from its second
// iteration on, and the compacted search (k_nn_coop_list) beats classifying 256 queries per workgroup and walking the few searchers of a crowded block one per
// lane (config 2's 0.4 m level, 10 per voxel: 4.75 -> 4.56 ms per registration; as a registration's FIRST level the same density is better served by the tiles:
// with the threshold at 4 config 2's 0.2 m level loses 15 %).
#define M3D_COOP_DENSITY_LATER 8
__host__ __device__ inline bool m3d_dense_level(uint32_t n_tgt, uint32_t occupied_voxels, uint32_t n_src, int level) {
    const unsigned long long n = n_tgt, v = occupied_voxels;
    return n > (unsigned long long)M3D_COOP_DENSITY * v || (level > 0 && n > (unsigned long long)M3D_COOP_DENSITY_LATER * v) ||
           (n > (unsigned long long)M3D_COOP_DENSITY_MAP * v && n >= (unsigned long long)M3D_COOP_MAP_RATIO * (unsigned long long)n_src);
}


// ---- -DM3D_JITTER (diagnosis build, `make jitter` -> libm3dreg_jitter.so) ---------------------------------------------------------------------
// Behind EVERY __syncthreads() about one wave in four pauses for 0.2 - 4 us (wave-uniform, pseudo-random from the clock, the wave and the workgroup): the waves
// of a workgroup leave every barrier far apart instead of a few cycles apart. A word of LDS that one wave still tests while another already re-uses it — the
// race rounds 5-6 shipped in tile_build_role, one pipelined step in ~10 000 — then goes wrong in nearly every workgroup that has the pattern; code that is
// correctly fenced returns the same bits, only later. The GPU suite runs against this build (M3DREG_LIB selects it): profiles/r06_fault_hunt.txt.
#ifdef M3D_JITTER
__device__ __forceinline__ void m3d_sync_jitter() {
    __syncthreads();
    unsigned int t = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)clock64());
    t ^= (threadIdx.x >> 6) * 0x9E3779B1u ^ blockIdx.x * 0x85EBCA6Bu;
    t ^= t >> 15; t *= 0x2C1B3C6Du; t ^= t >> 12; t *= 0x297A2D39u; t ^= t >> 15;
    t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    if ((t & 3u) == 0u) { const int n = 1 + (int)((t >> 4) & 15u); for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(8); }
}
#define __syncthreads() m3d_sync_jitter()
// ... and at the ENTRY of the kernels (called from the block -> work maps every kernel starts with: icp.hip m3d_map_block, bucket.hip m3d_row_block*; k_nn_tiles itself):
// a wave in four starts up to 8 us late, so the workgroups of a launch — and the waves of a workgroup before its first barrier — arrive at every cross-workgroup
// protocol (arrival tickets, partial sums, work-item lists, table inserts) in orders an undisturbed launch hardly ever produces.
__device__ __forceinline__ void m3d_entry_jitter() {
    unsigned int t = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)clock64());
    t ^= (threadIdx.x >> 6) * 0x9E3779B1u ^ blockIdx.x * 0xC2B2AE35u;
    t ^= t >> 16; t *= 0x7FEB352Du; t ^= t >> 15; t *= 0x846CA68Bu; t ^= t >> 16;
    t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    if ((t & 3u) == 0u) { const int n = 1 + (int)((t >> 4) & 31u); for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(8); }
}
#define M3D_ENTRY_JITTER() m3d_entry_jitter()
#else
#define M3D_ENTRY_JITTER() ((void)0)
#endif

// ---- -DM3D_CHECKED (diagnosis build, `make checked` -> libm3dreg_checked.so; DESIGN.md §8 "the unexplained GPU memory fault") --------------------------------
// Every index that a kernel takes out of MEMORY before it addresses global memory with it — match indices, permutation values, tile / image numbers, staged
// counts, work-item fields, block orders — goes through M3D_CHK(site, index, bound): out of range, the FIRST offence is recorded {count, site, index, bound} in
// a per-translation-unit device word (read back by m3dreg_debug_checks) and the index is clamped to 0, so the run goes on and reports instead of faulting.
// In the shipped build the macro is the identity.
#ifdef M3D_CHECKED
static __device__ unsigned int g_m3d_chk[4];
__device__ __forceinline__ unsigned long long m3d_chk_fail(unsigned int site, unsigned long long idx, unsigned long long bound) {
    if (atomicAdd(&g_m3d_chk[0], 1u) == 0u) { g_m3d_chk[1] = site; g_m3d_chk[2] = (unsigned int)(idx > 0xFFFFFFFFull ? 0xFFFFFFFFull : idx); g_m3d_chk[3] = (unsigned int)bound; }
    return 0ull;
}
template <typename T> __device__ __forceinline__ T m3d_chk(unsigned int site, T idx, unsigned long long bound) {
    return (unsigned long long)idx < bound ? idx : (T)m3d_chk_fail(site, (unsigned long long)idx, bound);   // (a negative index converts to a huge one: caught)
}
#define M3D_CHK(site, idx, bound) m3d_chk((site), (idx), (unsigned long long)(bound))
#define M3D_CHK_LE(site, idx, bound) M3D_CHK(site, idx, (unsigned long long)(bound) + 1ull)
#define M3D_CHK_READER(name) extern "C" hipError_t name(unsigned int* out, int reset) { \
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m3d_chk), sizeof(unsigned int) * 4); \
    if (e == hipSuccess && reset) { const unsigned int z[4] = { 0u, 0u, 0u, 0u }; e = hipMemcpyToSymbol(HIP_SYMBOL(g_m3d_chk), z, sizeof(z)); } return e; }
#else
#define M3D_CHK(site, idx, bound) (idx)
#define M3D_CHK_LE(site, idx, bound) (idx)
#define M3D_CHK_READER(name)
#endif

// ---- spec primitives shared by every kernel (operation order is normative) --------
__device__ __forceinline__ bool m3d_finite3(float x, float y, float z) {
    return isfinite(x) && isfinite(y) && isfinite(z);
}
__device__ __forceinline__ float m3d_cell_f(float v, float mn, float inv_leaf) {
    float d = v - mn;          // compiled with -ffp-contract=off: sub, mul, floor stay separate
    float s = d * inv_leaf;
    return floorf(s);
}
// Spec §Grid: voxel sort key = compact Morton code of the bucket coordinates << 3 | position in the bucket
__device__ __forceinline__ uint32_t m3d_voxel_key(const int32_t (&cb)[3], int ix, int iy, int iz) {
    const uint32_t cx = (uint32_t)ix >> 1, cy = (uint32_t)iy >> 1, cz = (uint32_t)iz >> 1;
    uint32_t code = 0; int pos = 0;
#pragma unroll
    for (int b = 0; b < 11; b++) {
        if (b < cb[0]) { code |= ((cx >> b) & 1u) << pos; pos++; }
        if (b < cb[1]) { code |= ((cy >> b) & 1u) << pos; pos++; }
        if (b < cb[2]) { code |= ((cz >> b) & 1u) << pos; pos++; }
    }
    return (code << 3) | ((uint32_t)ix & 1u) | (((uint32_t)iy & 1u) << 1) | (((uint32_t)iz & 1u) << 2);
}
__device__ __forceinline__ uint32_t m3d_bucket_key(const M3dGrid& g, int cx, int cy, int cz) {
    return (uint32_t)cx | ((uint32_t)cy << g.cb[0]) | ((uint32_t)cz << (g.cb[0] + g.cb[1]));
}
__device__ __forceinline__ uint32_t m3d_hash_slot(uint32_t key, int hshift) {
    return (key * 0x9E3779B1u) >> hshift;
}
// ---- Spec §Grid on either side of the bus: everything the kernels need, from the exact AABB. The bucketing pipeline runs it on
// the device (k_grid_params); the arithmetic is float add / sub / mul / div / floor only (correctly rounded on both sides).
__host__ __device__ inline int m3d_bits_for(int32_t d) {
    int b = 1;
    while (b < 31 && (1u << b) < (uint32_t)d) b++;   // d <= 2^30
    return b;
}
__host__ __device__ inline float m3d_cell_hd(float v, float mn, float inv_leaf) {
    const float d = v - mn;
    const float s = d * inv_leaf;
    return floorf(s);
}
// returns 0 or M3D_ERR_GRID_TOO_LARGE; hmask / hshift are left 0 (derived after the sort)
__host__ __device__ inline int m3d_make_grid(const float mn[3], const float mx[3], float leaf, int32_t n_valid, M3dGrid& g, int32_t bits[3],
                                             float& lbound) {
    g.leaf = leaf;
    g.inv_leaf = 1.0f / leaf;
    g.n_valid = n_valid;
    float half_max = 0.0f, amax = 0.0f, ext_max = 0.0f;
    int total_bits = 0;
    for (int a = 0; a < 3; a++) {
        g.mn[a] = mn[a];
        const float fc = m3d_cell_hd(mx[a], mn[a], g.inv_leaf);
        if (!(fc < 1073741824.0f)) return M3D_ERR_GRID_TOO_LARGE;
        g.dims[a] = (int32_t)fc + 1;
        bits[a] = m3d_bits_for((g.dims[a] + 1) >> 1);   // bit width of the BUCKET coordinate
        if (bits[a] > 11) return M3D_ERR_GRID_TOO_LARGE;
        g.cb[a] = bits[a];
        total_bits += bits[a];
        const float ext = mx[a] - mn[a];
        const float half = ext * 0.5f;
        g.center[a] = mn[a] + half;
        if (half > half_max) half_max = half;
        amax = fmaxf(amax, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
        ext_max = fmaxf(ext_max, ext);
    }
    if (total_bits + 3 > 31) return M3D_ERR_GRID_TOO_LARGE;
    lbound = half_max + 3.0f * leaf;
    g.hmask = 0;
    g.hshift = 0;
    // pruning slack (not part of the results: only makes the box test conservative)
    g.prune_slack = 1.0e-6f * (amax + ext_max) + 1.0e-3f * leaf;
    return 0;
}
// ceil(log2(x)) of a positive normal double from its bits (what frexp would say, without libm)
__host__ __device__ inline int m3d_ceil_log2_d(double x) {
    union { double d; unsigned long long u; } c; c.d = x;
    const int ex = (int)((c.u >> 52) & 0x7FFull) - 1022;           // x = m * 2^ex, m in [0.5, 1)
    return ((c.u & 0xFFFFFFFFFFFFFull) == 0ull) ? ex - 1 : ex;     // m == 0.5: an exact power of two
}
// Spec §Linearisation: per-class fixed-point exponents e = 30 - ceil_log2(bound), and their scales 2^e
__host__ __device__ inline void m3d_fixed_exps(float lbound, float max_corr_dist, int32_t e[6], float S[6]) {
    const double lb = (double)lbound, D = (double)max_corr_dist * 1.001;
    e[0] = 30 - m3d_ceil_log2_d(3.0 * lb * lb);
    e[1] = 30 - m3d_ceil_log2_d(1.7320508075688772 * lb);
    e[2] = 30;
    e[3] = 30 - m3d_ceil_log2_d(1.7320508075688772 * lb * D);
    e[4] = 30 - m3d_ceil_log2_d(D);
    e[5] = 30 - m3d_ceil_log2_d(D * D);
    for (int k = 0; k < 6; k++) { union { uint32_t u; float f; } c; c.u = (uint32_t)(e[k] + 127) << 23; S[k] = c.f; }   // 2^e, -126 <= e <= 127
}


// ---- Spec §Trig (round 6): sin / cos of a float angle for the LaserScan path's float-overload reading (m3d_aggregator.cpp:281-282 with ::cos(float) in sight:
// m3dagg_set_scan_trig(1)). The C library's cosf / sinf are not one function — glibc's and the device's differ in the last bit for a few per cent of the arguments —
// so the reading is SPECIFIED: double arithmetic only (+, -, x, rint: correctly rounded on both sides of the bus, this file is built with -ffp-contract=off),
// a two-word Cody-Waite reduction by pi/2 (exact products for |x| < 2^19), the classic degree-13 / degree-12 minimax kernels on [-pi/4, pi/4], ONE rounding to float
// at the end. The result is the correctly rounded sine / cosine for all but ~1 argument in 10^8, i.e. within 1 ulp of any C library's (side check in the tests).
// oracle/m3d_agg_oracle.c restates it: same constants, same order, same bits.
__host__ __device__ inline void m3d_sincosf_spec(float xf, float& s_out, float& c_out) {
    const double x = (double)xf;
    const double fn = rint(x * 6.36619772367581382433e-01);                       // x * 2 / pi, to the nearest integer
    const double r = (x - fn * 1.57079632673412561417e+00) - fn * 6.07710050650619224932e-11;   // pi/2 = its first 33 bits + the next 53
    const double z = r * r;
    const double sp = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sn = r + r * z * (-1.66666666666666324348e-01 + z * sp);
    const double cp = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double cs = 1.0 - (0.5 * z - z * cp);
    const long long q = (long long)fn & 3ll;                                       // quadrant (two's complement: also for negative fn)
    const double sv = (q == 0) ? sn : ((q == 1) ? cs : ((q == 2) ? -sn : -cs));
    const double cv = (q == 0) ? cs : ((q == 1) ? -sn : ((q == 2) ? -cs : sn));
    const bool fin = (xf - xf) == 0.0f;                                            // NaN / infinity in: NaN out
    s_out = fin ? (float)sv : (xf - xf);
    c_out = fin ? (float)cv : (xf - xf);
}

// Pointers that arrive inside descriptors loaded from memory are "generic" to the compiler, which then
// emits flat_load (slower, and every wait drains both counters). They all point to hipMalloc'ed HBM,
// so the kernels re-type them as global (address space 1) before use.
#define M3D_GLOBAL __attribute__((address_space(1)))
typedef uint32_t m3d_u32x4 __attribute__((ext_vector_type(4)));
typedef float m3d_f32x4 __attribute__((ext_vector_type(4)));
typedef const M3D_GLOBAL m3d_u32x4* m3d_gu4;
typedef const M3D_GLOBAL m3d_f32x4* m3d_gf4;
typedef const M3D_GLOBAL uint32_t* m3d_gu32;
__device__ __forceinline__ m3d_gu4 m3d_as_global(const uint4* p) { return (m3d_gu4)(const void M3D_GLOBAL*)p; }
__device__ __forceinline__ m3d_gf4 m3d_as_global(const float4* p) { return (m3d_gf4)(const void M3D_GLOBAL*)p; }
__device__ __forceinline__ m3d_gu32 m3d_as_global(const uint32_t* p) { return (m3d_gu32)(const void M3D_GLOBAL*)p; }
__device__ __forceinline__ uint4 m3d_ld(m3d_gu4 p, size_t i) { const m3d_u32x4 v = p[i]; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 m3d_ld(m3d_gf4 p, size_t i) { const m3d_f32x4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
// streamed-once data (source points, per-query results): non-temporal so it does not evict the gathered
// target points / bucket table from the XCD's 4 MiB L2 (one pair's gather set is ~2.6 MB)
// packed {x, y, z} source points: one 12-byte load per lane (global_load_dwordx3; consecutive lanes: consecutive addresses)
typedef float m3d_f32x3 __attribute__((ext_vector_type(3)));
typedef const M3D_GLOBAL float* m3d_gf3;
__device__ __forceinline__ m3d_gf3 m3d_as_global3(const float* p) { return (m3d_gf3)(const void M3D_GLOBAL*)p; }
__device__ __forceinline__ float4 m3d_ld3(m3d_gf3 p, size_t i) {
    const m3d_f32x3 v = *reinterpret_cast<const M3D_GLOBAL m3d_f32x3*>(p + 3 * i);   // (12-byte records: 4-byte aligned, which is all dwordx3 needs)
    return make_float4(v.x, v.y, v.z, 0.f);
}
__device__ __forceinline__ float4 m3d_ld_stream(m3d_gf4 p, size_t i) { const m3d_f32x4 v = __builtin_nontemporal_load(&p[i]); return make_float4(v.x, v.y, v.z, v.w); }

// slot of the bucket `key` or -1; `lo` receives the first half of its entry {key, start, count, big}
__device__ __forceinline__ int m3d_find_bucket(const M3dBucket* __restrict__ htab, uint32_t hmask, int hshift, uint32_t key, uint4& lo) {
    uint32_t h = m3d_hash_slot(key, hshift);
    for (;;) {
        lo = reinterpret_cast<const uint4*>(htab)[2 * (size_t)h];
        if (lo.x == key) return (int)h;
        if (lo.x == M3D_INVALID_KEY) return -1;
        h = (h + 1) & hmask;
    }
}
// [begin, end) sorted positions of voxel `sub` of a found bucket (lo = first half, hi = second half of the entry)
__device__ __forceinline__ uint2 m3d_sub_range(const uint4& lo, const uint4& hi, const uint32_t* __restrict__ bigcum, int sub) {
    uint32_t c0, c1;
    if (lo.w == 0) {
        const unsigned long long a = ((unsigned long long)hi.y << 32) | hi.x, b = ((unsigned long long)hi.w << 32) | hi.z;
        const unsigned long long w1 = (sub < 4) ? a : b;
        c1 = (uint32_t)(w1 >> (16 * (sub & 3))) & 0xFFFFu;
        const int sm = sub - 1;
        const unsigned long long w0 = (sm < 4) ? a : b;
        c0 = sub ? ((uint32_t)(w0 >> (16 * (sm & 3))) & 0xFFFFu) : 0u;
    } else {
        const uint32_t* bc = bigcum + 8 * (size_t)(lo.w - 1);
        c1 = bc[sub];
        c0 = sub ? bc[sub - 1] : 0u;
    }
    return make_uint2(lo.y + c0, lo.y + c1);
}
// [begin, end) of an arbitrary voxel (ix,iy,iz inside the grid); empty range when the voxel is unoccupied
__device__ __forceinline__ uint2 m3d_find_voxel(const M3dLevelDev& L, int ix, int iy, int iz) {
    uint4 lo;
    const int h = m3d_find_bucket(L.htab, L.g.hmask, L.g.hshift, m3d_bucket_key(L.g, ix >> 1, iy >> 1, iz >> 1), lo);
    if (h < 0) return make_uint2(0u, 0u);
    const uint4 hi = reinterpret_cast<const uint4*>(L.htab)[2 * (size_t)h + 1];
    return m3d_sub_range(lo, hi, L.bigcum, (ix & 1) | ((iy & 1) << 1) | ((iz & 1) << 2));
}
