// m3d_device.h — device-visible data layout of libm3dreg (gfx950 only; no CUDA/dual path).
//
// HBM layout of one bucketed cloud level (DESIGN.md §Data layout):
//   pts  float4[n]  cell-sorted points {x, y, z, bits(input_index | last_in_cell << 31)} — one 16-B
//                   gather per candidate, a cell is a contiguous run terminated by the flag bit
//   nrm  float4[n]  unit normals in the same order ({0,0,0,0} = no usable normal), point-to-plane only
//   htab uint2[T]   open-addressing hash of occupied voxels {key, first sorted position}, T = 2^k >= 2n
// The source side of a registration is plain SoA x[n], y[n], z[n] in input order (coalesced stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define M3D_NSUMS 29
#define M3D_INVALID_KEY 0xFFFFFFFFu
#define M3D_LAST_FLAG 0x80000000u
#define M3D_MAX_TRACE 256

struct M3dGrid {           // geometry of one voxel grid (host computes it from the exact AABB)
    float mn[3];
    float inv_leaf;
    float center[3];
    float leaf;
    int32_t dims[3];
    int32_t sy, sz;        // key = ix | iy << sy | iz << sz
    int32_t hshift;        // hash slot = (key * 0x9E3779B1u) >> hshift
    uint32_t hmask;
    int32_t n_valid;
    float prune_slack;     // absolute slack [m] subtracted from cell-box gaps before pruning
};

struct M3dLevelDev {       // what the NN / ICP kernels need from a target level
    const float4* pts;
    const float4* nrm;
    const uint2* htab;
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
    uint32_t ticket;
    int32_t pad[2];
};

struct M3dJob {            // one pair at one level
    const float* sx;
    const float* sy;
    const float* sz;
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
};

// ---- spec primitives shared by every kernel (operation order is normative) --------
__device__ __forceinline__ bool m3d_finite3(float x, float y, float z) {
    return isfinite(x) && isfinite(y) && isfinite(z);
}
__device__ __forceinline__ float m3d_cell_f(float v, float mn, float inv_leaf) {
    float d = v - mn;          // compiled with -ffp-contract=off: sub, mul, floor stay separate
    float s = d * inv_leaf;
    return floorf(s);
}
__device__ __forceinline__ uint32_t m3d_hash_slot(uint32_t key, int hshift) {
    return (key * 0x9E3779B1u) >> hshift;
}
// first sorted position of the voxel `key`, or -1
__device__ __forceinline__ int m3d_find_cell(const uint2* __restrict__ htab, uint32_t hmask, int hshift, uint32_t key) {
    uint32_t h = m3d_hash_slot(key, hshift);
    for (;;) {
        uint2 e = htab[h];
        if (e.x == key) return (int)e.y;
        if (e.x == M3D_INVALID_KEY) return -1;
        h = (h + 1) & hmask;
    }
}
