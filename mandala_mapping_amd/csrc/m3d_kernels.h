// m3d_kernels.h — internal launcher interface between the C-ABI layer (m3dreg_api.cpp) and the HIP
// kernels (bucket.hip, icp.hip). Nothing here is exported from libm3dreg.so.
#pragma once
#include "m3d_device.h"
#include <cstdio>
#include <cstdlib>

// M3DREG_DEBUG_SYNC=1: synchronise after every launch and name it on stderr (locates a faulting kernel)
static inline bool m3d_dbg_on() { static const bool on = getenv("M3DREG_DEBUG_SYNC") != nullptr; return on; }
#define M3D_DBG(s, name)                                                                       \
    do {                                                                                       \
        if (m3d_dbg_on()) {                                                                    \
            fprintf(stderr, "[m3dreg] %s ...", name); fflush(stderr);                          \
            const hipError_t _de = hipStreamSynchronize(s);                                    \
            fprintf(stderr, " %s\n", hipGetErrorString(_de)); fflush(stderr);                  \
        }                                                                                      \
    } while (0)

struct M3dDecode {               // one cloud of a decode batch (a2)
    const uint8_t* raw;          // PointCloud2 payload on the device
    int n, step, ox, oy, oz;
    int generic;                 // 0: little-endian FLOAT32 fields at 4-byte aligned addresses, rows without padding (what m3d_aggregator sends);
                                 // 1: the general sensor_msgs/PointCloud2 case below, assembled byte by byte
    int width, row_step;         // generic: point i sits at (i / width) * row_step + (i % width) * step
    int f64[3];                  // generic: field a is FLOAT64 (rounded to float) instead of FLOAT32
    int bigendian;               // generic: fields are stored big-endian
    float4* xyz;                 // [n] out: coordinates in input order, one 16-B element per point (finalize gathers them by permutation)
    uint32_t* aabb;              // [8] out (zeroed): ~ordered min[3] (as max), ordered max[3], finite count
};

struct M3dBuild {                // one voxel grid of a bucketing batch (a3, a4, a9)
    int n;
    int sort_passes;             // 8-bit LSD passes needed to order this grid's keys
    int ntiles;
    int fine;                    // -1, or (a coarser level of a pyramid) the index of the build of the SAME cloud's finest level: this grid is then sorted
                                 // from the finest level's order, so that inside one of its voxels the points lie in the finest level's Morton order
                                 // instead of input order — chunks of 16 consecutive points become compact boxes, and a coarse level's search (hundreds
                                 // of points per voxel) skips nearly all of them by their box. The spec's order (stable: input order inside a voxel) is
                                 // what m3dreg_cloud_export returns (restored on the host); results do not depend on the order inside a voxel.
    const float4* xyz;           // input-order coordinates of the cloud
    M3dGrid grid;                // in: grid.leaf; everything else is derived ON THE DEVICE from the cloud's exact AABB (k_grid_params), like sort_passes
    const uint32_t* aabb;        // [8] the cloud's k_decode_aabb words
    uint32_t* keys;              // [n] out: key per input point
    uint32_t *ka, *va, *kb, *vb; // [n] sort ping-pong workspace
    uint32_t* hist;              // [256 * ntiles] workspace
    uint32_t* skey_out;          // [n] out: sorted keys
    uint32_t* perm_out;          // [n] out: permutation
    float4* pts;                 // [n] out
    float* src3;                 // [3 n] out (a cloud's FINEST level, else null): the sorted points packed {x, y, z}: what a registration streams as its source
    M3dBucket* htab;             // [hcap] out (worst-case allocation; the used size is derived on the device)
    uint32_t hcap;
    uint32_t* bigcum;            // [bigcap][8] out
    uint32_t bigcap;
    float4* cbox;                // [2 * ceil(n / 16)] out: chunk boxes of the sorted points (levels only; null for normal grids)
    uint32_t* blkw;              // [ceil(n / 256)] workspace: occupied voxels per 256 sorted positions (k_count_cells)
    uint32_t* order;             // [ceil(n / 256)] out (levels only, else null): the 256-point blocks of the sorted cloud, most crowded first (k_table_params)
    uint32_t* dyn;               // out: the grid's M3dLevelMeta (144 B; its first 8 words are the dyn counters {occupied voxels, hmask, hshift, ...})
    // a9 (round 5): the normal-estimation grid is never sorted. Its build (a cloud's first) carries the grid's GEOMETRY (k_grid_params) and an open-addressing
    // table of its occupied voxels, filled from the cloud's finest level in that level's sorted order (a normal-grid voxel is a short run there): host sets n = 0,
    // so every stage of the sort / table pipeline skips the build.
    uint32_t* nkeys;             // [ncap] normal grids only (else null): voxel key ix | iy << (cb0 + 1) | iz << (cb0 + cb1 + 2), or M3D_INVALID_KEY (k_clear_table)
    long long* mom;              // [ncap][10] exact moments {n, S[3], P[6]} of the slot's voxel (zeroed by the thread that inserted the key)
    float4* nnrm;                // [ncap] the voxel's normal ({0,0,0,0}: none)
    uint32_t* nlist;             // [256 * ceil(n / 256)] the slots every 256-position block of the finest level inserted (k_finalize_level), nvcnt[blk] of them
    uint32_t* nvcnt;             // [ceil(n / 256)]
    uint32_t ncap;               // slots: a power of two > 1.25 n
    int32_t nshift;              // 32 - log2(ncap)
    int32_t nrm_build;           // level builds of a cloud with normals: index of the cloud's normal-grid build (else -1)
    int32_t nrm_feed;            // 1 on the level (a cloud's finest) whose sorted order fills the normal grid's table
    float4* nrm_sorted;          // [n] out: the normals in this level's sorted order (k_nrm_handout; level builds of point-to-plane clouds, else null)
    M3dTileHdr* thdr;            // [m3d_tiles_of(n)] out: tile headers (levels of clouds that can be targets, else null)
    uint8_t* timg;               // [tiles + pool][M3D_TILE_IMG_BYTES] out: tile images (k_tiles_normals), tiles = m3d_tiles_of(n), pool = m3d_tile_pool(tiles)
    M3dTileImgMeta* timeta;      // [tiles + pool] out
    uint32_t* occ;               // [2^M3D_OCC_BITS / 32] out: occupancy bitmap of the bucket positions (levels of clouds that can be targets, else null); dyn[7] = 1 when valid
};

float m3d_unord_f32(uint32_t u);
int m3d_sort_tiles(int n);
hipError_t m3d_launch_decode_aabb(hipStream_t s, const M3dDecode* d_descs, int n_clouds, int max_n);
// n_builds = n_clouds * grids_per_cloud, the builds of a cloud are consecutive. No host input beyond sizes: the grid geometry, the
// number of sort passes and the error state of every cloud are derived on the device.
hipError_t m3d_launch_bucket_batch(hipStream_t s, M3dBuild* d_builds, int n_clouds, int grids_per_cloud, int max_n, bool any_normals,
                                   int any_tiles, float plane_ratio, int min_pts, float min_spread, bool pyramid);   // any_tiles: 0 none, 1 only clouds' LAST builds (finest levels) have tiles, 2 any build may; pyramid: some build has fine >= 0
hipError_t m3d_launch_export_sorted(hipStream_t s, const float4* pts, const float4* nrm, int n, float* xyz, float* nxyz);

// aggregate.hip
struct M3dAggArgs {
    int mode;                    // 0 = PointCloud2 payload, 1 = LaserScan ranges (cos / sin in double, product rounded once), 2 = LaserScan ranges (cosf / sinf)
    int n;
    const uint8_t* raw; int step, ox, oy, oz;
    const float* ranges; float angle_min, angle_inc;
    double m[9], o[3];           // tf::Transform: row-major basis + origin
    double bb[6];                // x_up, x_down, y_up, y_down, z_up, z_down
    float4* out;                 // aggregate buffer, pcl::PointXYZ layout
    uint32_t* count;             // [2] device: points aggregated so far, overflow flag
    uint32_t capacity;
    uint32_t* block_counts;      // [blocks] workspace
};
hipError_t m3d_launch_aggregate(hipStream_t s, const M3dAggArgs& A);

// calibrate.hip
struct M3dCalArgs {
    const float4* pts;           // [n] raw points of every segment {x, y, z, bits(segment index)}
    int n, n_seg, axis;          // axis = laserUpAxis (the RAW coordinate whose sign splits the sweep)
    const float* mm;             // [candidates][n_seg][12]: original_Transform * laserOffsetMatrix, row-major linear (9) + translation (3)
    unsigned long long* keys;    // [candidates][tsize] (half << 63 | voxel) or all-ones = empty
    uint32_t* cnt;               // [candidates][tsize] points per voxel
    long long* sums;             // [candidates][tsize][3] 2^-16 m fixed-point coordinate sums
    uint32_t tsize; int tshift;  // table size (power of two >= 2 n), 64 - log2(tsize)
    unsigned int* result;        // [candidates][3]: {cost, voxels of the first half, voxels of the second half}
    int* status;                 // [candidates]: 1 = a voxel coordinate left the +-2^20 range
};
hipError_t m3d_launch_calibration(hipStream_t s, const M3dCalArgs& A, int n_candidates);

// map.hip
struct M3dMapArgs {
    const float4* src;           // [n] the scan's points in input order (m3dreg_cloud::xyz)
    int n;
    float R[9], t[3];            // pose of the scan in the map frame, rounded to float (row-major R)
    float inv_leaf;              // 1.0f / dedup leaf
    unsigned long long* keys;    // [tsize] occupancy table: voxel key or all-ones
    uint32_t* epoch;             // [tsize] insert that created the slot
    uint32_t* owner;             // [tsize] lowest input index of the creating insert that fell into the voxel (~0 initially)
    uint32_t tsize; int tshift;
    uint32_t cur_epoch;
    uint32_t* slot_of;           // [n] workspace: table slot of every point of this insert (~0 = dropped)
    uint32_t* block_counts;      // [blocks] workspace
    float4* out;                 // map points, pcl::PointXYZ layout
    uint32_t* count;             // [1] points in the map
    uint32_t capacity;
    uint32_t* flags;             // [4]: {table full, voxel out of range, -, point buffer full}
};
hipError_t m3d_launch_map_insert(hipStream_t s, const M3dMapArgs& A);

// loop.hip (second half: loop-closure candidate generation)
struct M3dLoopSignArgs {
    const float4* src;           // [n] the keyframe cloud's points in input order (m3dreg_cloud::xyz)
    int n;
    float R[9], t[3];            // pose of the keyframe in the map frame, rounded to float (row-major R)
    float inv_leaf;              // 1.0f / sig_leaf
    uint32_t* sig;               // [2^log2_bits / 32] the keyframe's signature (zeroed by the launcher)
    int log2_bits;               // 10 .. 18
};
struct M3dLoopScoreArgs {
    const uint32_t* sig;         // [keyframes][W] signatures
    const float4* pos;           // [keyframes] {t.x, t.y, t.z, bits(popcount of the signature)}
    int W;                       // words per signature (a multiple of 4)
    int row0, n_rows;            // query rows (keyframes) row0 .. row0 + n_rows - 1
    int min_gap;                 // columns of row i: j <= i - min_gap
    float r2;                    // radius^2
    uint32_t* ov;                // [n_rows][ov_stride] out: overlap of (row, j), or 0xFFFFFFFF where gap / distance rule the pair out
    int ov_stride;
    int j_per_wg, tr;            // set by the launcher: columns per workgroup, rows per tile
};
hipError_t m3d_launch_loop_sign(hipStream_t s, const M3dLoopSignArgs& A, float4* pos_k);
hipError_t m3d_launch_loop_score(hipStream_t s, M3dLoopScoreArgs A, uint2* d_out /* [n_rows][top_k] {j or ~0, overlap} */, int top_k, uint32_t thr_q16);

// debug.hip (diagnosis only: M3DREG_POISON)
hipError_t m3d_launch_poison(hipStream_t s, void* p, size_t bytes, uint32_t seed);

// icp.hip: one Gauss-Newton iteration = k_nn_iter (classify + search / bin), k_nn_tiles (binned searches from LDS), k_accumulate_matches
// (residuals + reduction; its last block per pair also solves and updates the pose)
struct M3dNnWork {               // workspace of the batch, all per pair with the same stride (a whole number of 256-query blocks)
    int2* match;                 // [n_pairs * stride] {match, certificate word} of every query, kept between iterations (certified / seeds the next search)
    long long* cache;            // [n_pairs * stride] voxel of each cached "no point in the neighbourhood" verdict
    const float* ring;           // [n_pairs][32][12] pose rings (== jobs[pair].ring)
    int certify;                 // A/B switch of the certificates (M3DREG_CERTIFY)
    int lane_min;                // a 256-query block with >= lane_min queries to search bins them / walks one query per lane, else 8 lanes per query
    long long* partials;         // [n_pairs][m3d_acc_blocks(max_n_src, n_pairs)][M3D_PARTIAL_STRIDE] block partial sums of the reduction pass
    M3dPairState* states;        // [n_pairs] the batch's pair states (== jobs[pair].st)
    unsigned int* tickets;       // [m3d_ticket_words(n_pairs, max_n_src)] arrival counters of the reduction pass (zero between launches)
    int stride;
    float seed_reach;            // seeds farther than this many voxel edges are not used (<= 0.99)
    int rot;                     // XCD rotation of the block -> pair map: differs between handles, so concurrent batches do not stack their k-th pairs on one XCD
    int tiles;                   // 1 = dense blocks bin their searches by target tile and k_nn_tiles answers them from LDS (M3DREG_TILES)
    int coop_kernel;             // 1 = k_nn_coop is launched behind k_nn_iter<false> and answers the pairs whose target level is dense; 2 = it is the only search kernel (icp.hip: M3dNnArgs)
    int coop_list;               // 1 / 2 = an iteration in which most / nearly all queries of a dense level are expected to be certified: k_nn_coop_list (searchers compacted per 64 / 128 queries) instead of k_nn_coop
    int lean;                    // 1 = the tile iterations run k_nn_iter<true> (classify + bin only; the reduction pass walks what it cannot bin); needs every target of the batch to have tiles (M3DREG_LEAN)
    int ntile_max;               // tiles per pair the arrays below are laid out for
    float4* rec;                 // [n_pairs][ntile_max * M3D_TILE_QCAP] query records
    float* recd;                 // same layout: squared distance to the seed
    unsigned long long rec_stride;
    unsigned int* tcnt;          // [n_pairs][cnt_stride] records per tile; zero between iterations
    int cnt_stride;
    uint2* witems;               // [wcap] work items of k_nn_tiles
    unsigned int* wcount;        // items published this iteration; zero between iterations
    int wcap;
    int alone;                   // m3dreg_set_latency_mode: the batch has the GPU to itself (m3d_acc_blocks)
};
// k0/k1 (optional): events recorded immediately before / after the dominant kernel of the iteration (k_nn_iter)
// seq / progress: the solve step stores {seq, pairs still active at this level} to *progress (device view of
// host-mapped memory, may be null) so that the host can stop enqueuing iterations without synchronising
hipError_t m3d_launch_icp_iteration(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric, int first_of_level,
                                    const M3dNnWork& w, unsigned int seq, unsigned long long* progress, hipEvent_t k0, hipEvent_t k1);
hipError_t m3d_launch_patch_jobs(hipStream_t s, M3dJob* d_jobs, int n_pairs, int cap_pairs, int n_levels);
int m3d_ticket_words(int n_pairs, int max_n_src, int alone);
int m3d_acc_blocks(int max_n_src, int n_pairs, int alone);   // workgroups per pair of the reduction pass (alone: m3dreg_set_latency_mode)
hipError_t m3d_launch_accumulate_only(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric, const M3dNnWork& w);
hipError_t m3d_launch_debug_candidates(hipStream_t s, const M3dLevelDev& L, const float* q_xyz, int nq, int32_t* out_cnt);
hipError_t m3d_launch_debug_nn(hipStream_t s, const M3dLevelDev& L, const float* q_xyz, int nq, float dmax2, int32_t* out_idx,
                               float* out_d2);
