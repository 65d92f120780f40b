// loop.hip — SURVEY.md §8 row f4 (second half): loop-closure candidate generation feeding the batch path (BASELINE config 4's
// "batch of 64 loop-closure scan pairs" has to come from somewhere). No reference source exists for this step — the node the reference
// launches is gpu_6dslam_node (/root/reference/m3d/m3d_husky_launch/launch/m3d_husky_bringup.launch:13), its repository an empty
// submodule (/root/reference/.gitmodules:1-3) —; the normative behaviour is DESIGN.md §10, restated by oracle/m3d_loop_oracle.c:
//   signature of keyframe k: bit hash(floor((R_k p + t_k) * inv_leaf)) for every finite point p of its cloud (u = R p + t: the fma chain of spec row a5);
//   score(i, j) = popcount(sig_i & sig_j) for j <= i - min_gap with |t_i - t_j|^2 <= r^2;
//   candidates of i = the top_k scores that reach min_overlap * min(pop_i, pop_j), ties towards the older keyframe.
// Hardware mapping: byte / integer work, HBM- and LDS-bound, no MFMA.
//   k_loop_sign   one pass over the cloud's resident float4 points; a workgroup ORs its 1024 points into an LDS bitmap (<= 32 KB) and then
//                 only its non-zero words into the keyframe's signature (a sweep sets a few thousand of 65536 bits: ~100 atomics per workgroup)
//   k_loop_pop    popcount of the finished signature, stored in the keyframe's position record
//   k_loop_score  the streaming pass: a workgroup keeps the signatures of a tile of <= 4 query rows in LDS; each of its four waves takes
//                 every fourth older keyframe j, tests gap + distance for the tile's rows (wave-uniform), and only then reads sig_j — 16 B per
//                 lane, 1 KiB per wave instruction, fully coalesced — against the rows in LDS: AND + v_bcnt, one butterfly per (row tile, j)
//   k_loop_topk   one workgroup per row: top_k rounds of a 64-bit argmax over its scores (key = overlap << 32 | ~j: unique, so the order of
//                 selection is the order of the keys — no dependence on launch geometry)
#include "m3d_kernels.h"

#define LOOP_OFF 1048576
#define LOOP_NONE 0xFFFFFFFFu

__device__ __forceinline__ bool loop_bit(const M3dLoopSignArgs& A, int i, uint32_t& bit) {
    const float4 p = A.src[i];
    if (!m3d_finite3(p.x, p.y, p.z)) return false;
    const float ux = fmaf(A.R[0], p.x, fmaf(A.R[1], p.y, fmaf(A.R[2], p.z, A.t[0])));
    const float uy = fmaf(A.R[3], p.x, fmaf(A.R[4], p.y, fmaf(A.R[5], p.z, A.t[1])));
    const float uz = fmaf(A.R[6], p.x, fmaf(A.R[7], p.y, fmaf(A.R[8], p.z, A.t[2])));
    if (!m3d_finite3(ux, uy, uz)) return false;
    const float fx = floorf(ux * A.inv_leaf), fy = floorf(uy * A.inv_leaf), fz = floorf(uz * A.inv_leaf);
    const float lim = (float)(LOOP_OFF - 1);
    if (!(fx > -lim && fx < lim && fy > -lim && fy < lim && fz > -lim && fz < lim)) return false;   // beyond +-2^20 signature voxels: not part of the signature
    const unsigned long long key = ((unsigned long long)(uint32_t)((int)fx + LOOP_OFF) << 42) | ((unsigned long long)(uint32_t)((int)fy + LOOP_OFF) << 21) |
                                   (unsigned long long)(uint32_t)((int)fz + LOOP_OFF);
    bit = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - A.log2_bits));
    return true;
}

#define LOOP_SIGN_PTS 1024   // points per workgroup of k_loop_sign
__global__ __launch_bounds__(256) void k_loop_sign(M3dLoopSignArgs A) {
    extern __shared__ uint32_t lbits[];   // 2^log2_bits / 32 words
    const int W = 1 << (A.log2_bits - 5);
    for (int w = threadIdx.x; w < W; w += 256) lbits[w] = 0u;
    __syncthreads();
    const int base = blockIdx.x * LOOP_SIGN_PTS;
#pragma unroll
    for (int k = 0; k < LOOP_SIGN_PTS / 256; k++) {
        const int i = base + k * 256 + threadIdx.x;
        uint32_t bit;
        if (i < A.n && loop_bit(A, i, bit)) atomicOr(&lbits[bit >> 5], 1u << (bit & 31u));
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += 256) {
        const uint32_t v = lbits[w];
        if (v) atomicOr(&A.sig[w], v);
    }
}

// one workgroup: popcount of a finished signature -> pos[k].w (as bits), beside the keyframe's position
__global__ __launch_bounds__(256) void k_loop_pop(const uint32_t* __restrict__ sig, int W, float4* __restrict__ pos_k, float tx, float ty, float tz) {
    __shared__ uint32_t part[4];
    uint32_t c = 0;
    for (int w = threadIdx.x; w < W; w += 256) c += (uint32_t)__popc(sig[w]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) *pos_k = make_float4(tx, ty, tz, __uint_as_float(part[0] + part[1] + part[2] + part[3]));
}

__device__ __forceinline__ float loop_dist2(const float4& a, const float4& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

#define LOOP_TR 4   // query rows per workgroup of k_loop_score at most (their signatures sit in LDS)
__global__ __launch_bounds__(256) void k_loop_score(M3dLoopScoreArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t rows[];   // [tr][W]
    const int W = A.W, tr = A.tr;
    const int r0 = A.row0 + (int)blockIdx.y * tr;                  // first keyframe of this tile
    const int nr = min(tr, A.row0 + A.n_rows - r0);                // rows of the tile that exist (>= 1 by the grid)
    const int jmax = (r0 + nr - 1) - A.min_gap + 1;                // columns any row of the tile may pair with: j < jmax
    const int j0 = (int)blockIdx.x * A.j_per_wg, j1 = min(j0 + A.j_per_wg, jmax);
    if (j0 >= j1) return;                                          // (workgroup-uniform)
    const m3d_gu4 gs = m3d_as_global(reinterpret_cast<const uint4*>(A.sig));
    const int W4 = W >> 2;                                         // uint4 words per signature
    for (int r = 0; r < nr; r++)
        for (int c = threadIdx.x; c < W4; c += 256) reinterpret_cast<uint4*>(rows)[r * W4 + c] = m3d_ld(gs, (size_t)(r0 + r) * W4 + c);
    __shared__ float4 rpos[LOOP_TR];
    if (threadIdx.x < nr) rpos[threadIdx.x] = A.pos[r0 + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = j0 + wave; j < j1; j += 4) {                      // (j is wave-uniform)
        const float4 pj = A.pos[j];
        uint32_t mask = 0;
#pragma unroll
        for (int r = 0; r < LOOP_TR; r++)
            if (r < nr && j <= (r0 + r) - A.min_gap && loop_dist2(rpos[r], pj) <= A.r2) mask |= 1u << r;
        uint32_t cnt[LOOP_TR] = { 0u, 0u, 0u, 0u };
        if (mask) {
            for (int c = lane; c < W4; c += 64) {
                const uint4 v = m3d_ld(gs, (size_t)j * W4 + c);
#pragma unroll
                for (int r = 0; r < LOOP_TR; r++) {
                    if (r < nr) {   // (uniform; rows outside the mask cost an LDS read, not a branch per lane)
                        const uint4 a = reinterpret_cast<const uint4*>(rows)[r * W4 + c];
                        cnt[r] += (uint32_t)(__popc(v.x & a.x) + __popc(v.y & a.y) + __popc(v.z & a.z) + __popc(v.w & a.w));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < LOOP_TR; r++) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cnt[r] += __shfl_xor(cnt[r], o);
            }
        }
        if (lane < nr && j <= (r0 + lane) - A.min_gap) {
            uint32_t mine = cnt[0];
            if (lane == 1) mine = cnt[1];
            if (lane == 2) mine = cnt[2];
            if (lane == 3) mine = cnt[3];
            A.ov[(size_t)(r0 + lane - A.row0) * A.ov_stride + j] = ((mask >> lane) & 1u) ? mine : LOOP_NONE;
        }
    }
}

// one workgroup per row: the top_k scores that pass the overlap threshold, best first
__global__ __launch_bounds__(256) void k_loop_topk(M3dLoopScoreArgs A, uint2* __restrict__ out, int top_k, uint32_t thr_q16) {
    const int i = A.row0 + (int)blockIdx.x;
    const int jn = i - A.min_gap + 1;
    const uint32_t pop_i = __float_as_uint(A.pos[i].w);
    const uint32_t* __restrict__ ov = A.ov + (size_t)blockIdx.x * A.ov_stride;
    __shared__ unsigned long long best[4];
    unsigned long long below = ~0ull;   // keys selected so far are >= below
    for (int k = 0; k < top_k; k++) {
        unsigned long long key = 0ull;
        for (int j = threadIdx.x; j < jn; j += 256) {
            const uint32_t o = ov[j];
            if (o == LOOP_NONE) continue;
            const uint32_t pop_j = __float_as_uint(A.pos[j].w);
            const uint32_t pm = pop_i < pop_j ? pop_i : pop_j;
            if (((unsigned long long)o << 16) < (unsigned long long)thr_q16 * (unsigned long long)pm) continue;
            const unsigned long long c = ((unsigned long long)o << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)j);
            if (c < below && c > key) key = c;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other > key ? other : key; }
        if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6] = key;
        __syncthreads();
        unsigned long long b = best[0];
        b = best[1] > b ? best[1] : b; b = best[2] > b ? best[2] : b; b = best[3] > b ? best[3] : b;
        __syncthreads();
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * top_k + k] = b ? make_uint2(0xFFFFFFFFu - (uint32_t)b, (uint32_t)(b >> 32)) : make_uint2(LOOP_NONE, 0u);
        if (!b) { for (int kk = k + 1 + (int)threadIdx.x; kk < top_k; kk += 256) out[(size_t)blockIdx.x * top_k + kk] = make_uint2(LOOP_NONE, 0u); break; }
        below = b;
    }
}

hipError_t m3d_launch_loop_sign(hipStream_t s, const M3dLoopSignArgs& A, float4* pos_k) {
    const int W = 1 << (A.log2_bits - 5);
    hipError_t e = hipMemsetAsync(A.sig, 0, sizeof(uint32_t) * (size_t)W, s);
    if (e != hipSuccess) return e;
    const int nblocks = (A.n + LOOP_SIGN_PTS - 1) / LOOP_SIGN_PTS;
    if (nblocks > 0) hipLaunchKernelGGL(k_loop_sign, dim3(nblocks), dim3(256), sizeof(uint32_t) * (size_t)W, s, A);
    M3D_DBG(s, "k_loop_sign");
    hipLaunchKernelGGL(k_loop_pop, dim3(1), dim3(256), 0, s, A.sig, W, pos_k, A.t[0], A.t[1], A.t[2]);
    M3D_DBG(s, "k_loop_pop");
    return hipGetLastError();
}

int m3d_loop_tile_rows(int W) { const int fit = (32768 / 4) / W; return fit >= LOOP_TR ? LOOP_TR : (fit < 1 ? 1 : fit); }

hipError_t m3d_launch_loop_score(hipStream_t s, M3dLoopScoreArgs A, uint2* d_out, int top_k, uint32_t thr_q16) {
    if (A.n_rows <= 0) return hipSuccess;
    A.tr = m3d_loop_tile_rows(A.W);
    const int tiles = (A.n_rows + A.tr - 1) / A.tr;
    const int jmax = (A.row0 + A.n_rows - 1) - A.min_gap + 1;   // columns of the last row
    if (jmax > 0) {
        // enough workgroups to fill 256 CUs a few times over, but at least 16 columns each (the row tile is staged once per workgroup)
        int splits = (2048 + tiles - 1) / tiles;
        const int max_splits = (jmax + 15) / 16;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        A.j_per_wg = (jmax + splits - 1) / splits;
        A.j_per_wg = (A.j_per_wg + 3) & ~3;
        splits = (jmax + A.j_per_wg - 1) / A.j_per_wg;
        hipLaunchKernelGGL(k_loop_score, dim3(splits, tiles), dim3(256), sizeof(uint32_t) * (size_t)A.W * (size_t)A.tr, s, A);
        M3D_DBG(s, "k_loop_score");
    }
    hipLaunchKernelGGL(k_loop_topk, dim3(A.n_rows), dim3(256), 0, s, A, d_out, top_k, thr_q16);
    M3D_DBG(s, "k_loop_topk");
    return hipGetLastError();
}
