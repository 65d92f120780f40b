// icp.hip — SURVEY.md §8 rows a5-a8: source transform, 27-voxel nearest-neighbour correspondence,
// point-to-point / point-to-plane residuals, the 29-term normal-equation reduction, the 6x6 solve and
// the SE(3) update, hand-written for gfx950. No reference source exists for this path (gpu_6dslam is
// an empty submodule); the nearest in-tree analogue of the neighbour query is the per-point
// KdTreeFLANN::radiusSearch loop at /root/reference/m3d/m3d_calibration/src/m3d_calibration_twiddle.cpp:292-304.
// The normative arithmetic is DESIGN.md §Spec; oracle/m3d_oracle.c restates it on the CPU and the
// parity tests compare every output bit for bit.
//
// Hardware mapping: source points stream coalesced as 16-B elements of the source cloud's own sorted array (a wave's
// queries are neighbours in space); candidates are 16-B gathers from the target's voxel-sorted float4 array
// (L2/MALL-resident); the 29 sums are int64 fixed point, reduced with a transposed wave64 butterfly, then LDS across
// the 4 waves, then one block partial that the pair's last block adds up — integer addition is associative, so the
// result does not depend on launch geometry. Two launches per Gauss-Newton iteration (k_nn_iter,
// k_accumulate_matches). No MFMA: there is no contraction.
#include "m3d_kernels.h"
M3D_CHK_READER(m3d_chk_read_icp)

// -DM3D_STATS: instrumented build (scripts/walk_stats.py): counts, per Gauss-Newton iteration of pair 0's clock, what the
// search does — one wave-aggregated atomic per event. Never defined in the shipped library.
#ifdef M3D_STATS
__device__ unsigned long long g_m3d_stats[64][24];
__device__ __forceinline__ void m3d_stat(int it, int what, unsigned int v = 1u) {
    const unsigned long long m = __ballot(1);
    unsigned int tot = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);   // inactive lanes contribute garbage-free zeros only if masked below
    (void)tot;
    // sum over ACTIVE lanes only: use ballot-popcount for unit events, atomics per lane otherwise
    if (v == 1u) { if ((int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&g_m3d_stats[it & 63][what], (unsigned long long)__popcll(m)); }
    else atomicAdd(&g_m3d_stats[it & 63][what], (unsigned long long)v);
}
__device__ __forceinline__ void m3d_stat_wave(int it, int what) {   // one count per WAVE that executes this point (SIMT trip counts)
    const unsigned long long m = __ballot(1);
    if ((int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&g_m3d_stats[it & 63][what], 1ull);
}
#define M3D_STAT(it, what) m3d_stat(it, what)
#define M3D_STATV(it, what, v) m3d_stat(it, what, v)
#define M3D_STATW(it, what) m3d_stat_wave(it, what)
extern "C" hipError_t m3d_debug_read_stats(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m3d_stats), sizeof(unsigned long long) * 64 * 24);
    if (e == hipSuccess && reset) { static unsigned long long z[64 * 24]; e = hipMemcpyToSymbol(HIP_SYMBOL(g_m3d_stats), z, sizeof(z)); }
    return e;
}
#else
#define M3D_STAT(it, what) ((void)0)
#define M3D_STATV(it, what, v) ((void)0)
#define M3D_STATW(it, what) ((void)0)
#endif

// -DM3D_BLOCKTIME: every block of k_nn_iter records {start, end} (100 MHz wall clock), where it ran and how many of its queries it
// searched, for the Gauss-Newton iteration g_m3d_blk_iter (scripts/block_times.py). Never defined in the shipped library.
#ifdef M3D_BLOCKTIME
__device__ unsigned long long g_m3d_blk[8192][8];   // {start, end, where, searched, sum of gather trips, max trips of a lane, sum of chunk tests, probes}
__device__ int g_m3d_blk_iter = 0;
extern "C" hipError_t m3d_debug_read_blocks(unsigned long long* out, int iter) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m3d_blk), sizeof(unsigned long long) * 8192 * 8);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_m3d_blk_iter), &iter, sizeof(int));
    return e;
}
__device__ int g_bt_on[8192];
__device__ unsigned long long g_m3d_tblk[16384][8];   // k_nn_tiles: {start, end, where, kind (0 tile / 1 global walk), records, staged points, staged at, pair}
extern "C" hipError_t m3d_debug_read_tile_blocks(unsigned long long* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m3d_tblk), sizeof(unsigned long long) * 16384 * 8);
    static unsigned long long z[16384 * 8];
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_m3d_tblk), z, sizeof(z));
    return e;
}
#define M3D_TBT_BEGIN() const unsigned long long tb0 = wall_clock64(); unsigned long long tb_st = 0, tb_t0 = 0; const unsigned int tb_i = blockIdx.x + gridDim.x * blockIdx.y; const bool tb_on = st->iters == g_m3d_blk_iter && tb_i < 16384
#define M3D_TBT_STAGED() do { if (tb_on) tb_t0 = wall_clock64(); } while (0)
#define M3D_TBT_SEARCHED() do { if (tb_on) { __syncthreads(); tb_st += wall_clock64() - tb_t0; } } while (0)
#define M3D_TBT_END(kind, nrec, npts) do { if (tb_on) { __syncthreads(); if (threadIdx.x == 0) { unsigned long long* b = g_m3d_tblk[tb_i]; b[0] = tb0; b[1] = wall_clock64(); \
    b[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
    b[3] = (kind); b[4] = (nrec); b[5] = (npts); b[6] = tb_st; b[7] = (unsigned long long)pair; } } } while (0)
#define M3D_BT_COUNT(W, f) ((W).f++)
#define M3D_BT_FLUSH(W) do { if (blockIdx.x < 8192 && g_bt_on[blockIdx.x]) { atomicAdd(&g_m3d_blk[blockIdx.x][4], (unsigned long long)(W).bt_trips); atomicMax(&g_m3d_blk[blockIdx.x][5], (unsigned long long)(W).bt_trips); \
    atomicAdd(&g_m3d_blk[blockIdx.x][6], (unsigned long long)(W).bt_chunks); atomicAdd(&g_m3d_blk[blockIdx.x][7], (unsigned long long)(W).bt_probes); } } while (0)
#define M3D_BT_BEGIN() const unsigned long long bt0 = wall_clock64(); const bool bt_on = st->iters == g_m3d_blk_iter && blockIdx.x < 8192; \
    if (blockIdx.x < 8192 && threadIdx.x == 0) g_bt_on[blockIdx.x] = bt_on ? 1 : 0; \
    if (bt_on && threadIdx.x < 4) g_m3d_blk[blockIdx.x][4 + threadIdx.x] = 0ull
#define M3D_BT_END(nsearch) do { if (bt_on && threadIdx.x == 0) { g_m3d_blk[blockIdx.x][0] = bt0; g_m3d_blk[blockIdx.x][1] = wall_clock64(); \
    g_m3d_blk[blockIdx.x][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
    g_m3d_blk[blockIdx.x][3] = (unsigned long long)(nsearch); } } while (0)
#else
#define M3D_BT_BEGIN() ((void)0)
#define M3D_BT_END(nsearch) ((void)0)
#define M3D_BT_COUNT(W, f) ((void)0)
#define M3D_BT_FLUSH(W) ((void)0)
#define M3D_TBT_BEGIN() ((void)0)
#define M3D_TBT_STAGED() ((void)0)
#define M3D_TBT_SEARCHED() ((void)0)
#define M3D_TBT_END(kind, nrec, npts) ((void)0)
#endif

#define ICP_THREADS 256
#define ICP_WAVES (ICP_THREADS / 64)

// ---- a6: exact NN over the 27 voxels around u --------------------------------------------------
// The 27 voxels live in at most 2x2x2 buckets: up to 8 independent hash probes are issued first
// (memory-level parallelism instead of a 27-step dependent chain), then only the voxels of the hit
// buckets that belong to the neighbourhood are walked. Pruning only ever skips a voxel whose box is
// provably farther than the current best (or than d_max), so the result is identical to the
// exhaustive walk of the oracle. Returns 1 (match) or -1; out_q = matched point (w = input index bits).
struct M3dQuery {          // per-query geometry shared by the global and the LDS search
    int ic[3];             // voxel of the query (may be -1 or dims: one voxel outside the grid)
    int lo[3], hi[3];      // neighbourhood clamped to the grid (voxel coordinates)
    float gl[3], gh[3];    // conservative distance to the lower / upper faces of the query's own voxel
};

__device__ __forceinline__ bool m3d_query_setup(const M3dGrid& g, float ux, float uy, float uz, M3dQuery& Q) {
    const float u[3] = { ux, uy, uz };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float f = m3d_cell_f(u[a], g.mn[a], g.inv_leaf);
        if (!(f >= -1.0f && f <= (float)g.dims[a])) return false;
        Q.ic[a] = (int)f;
        Q.lo[a] = max(Q.ic[a] - 1, 0);
        Q.hi[a] = min(Q.ic[a] + 1, g.dims[a] - 1);
        const float r = (u[a] - g.mn[a]) - f * g.leaf;
        Q.gl[a] = fmaxf(r - g.prune_slack, 0.f);
        Q.gh[a] = fmaxf((g.leaf - r) - g.prune_slack, 0.f);
    }
    return true;
}
// squared lower bound of the distance from the query to voxel (vx,vy,vz) of its neighbourhood
__device__ __forceinline__ float m3d_voxel_lb2(const M3dQuery& Q, int vx, int vy, int vz) {
    const int dx = vx - Q.ic[0], dy = vy - Q.ic[1], dz = vz - Q.ic[2];
    const float gx = dx < 0 ? Q.gl[0] : (dx > 0 ? Q.gh[0] : 0.f);
    const float gy = dy < 0 ? Q.gl[1] : (dy > 0 ? Q.gh[1] : 0.f);
    const float gz = dz < 0 ? Q.gl[2] : (dz > 0 ? Q.gh[2] : 0.f);
    return gx * gx + gy * gy + gz * gz;
}

struct M3dBest { int found; float d2; uint32_t oi; float4 q; };

// Branch-free argmin step: (d2, input index) packed into one 64-bit key — d2 >= 0, so its float bits order
// like the value, and the low word breaks ties towards the lowest input index exactly as the spec says.
// One v_cmp_lt_u64 + three v_cndmask per candidate instead of a saveexec/branch ladder.
__device__ __forceinline__ void m3d_argmin_step(unsigned long long& bestkey, int& best, float dd, uint32_t oi, int t) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | oi;
    const bool better = key < bestkey;
    bestkey = better ? key : bestkey;
    best = better ? t : best;
}
__device__ __forceinline__ float m3d_key_d2(unsigned long long key) { return __uint_as_float((uint32_t)(key >> 32)); }
// Same, additionally maintaining `sec`: a lower bound of d2 over every candidate that is NOT the current best
// (needed for the NN certificates). A candidate equal to the best key is the best point itself (clamped
// re-reads, or the seed met again during the walk) and does not count. The empty key decodes to NaN, which
// fminf ignores.
__device__ __forceinline__ void m3d_argmin_step2(unsigned long long& bestkey, int& best, float& sec, float dd, uint32_t oi, int t) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | oi;
    const bool better = key < bestkey;
    const float loser = m3d_key_d2(better ? bestkey : key);
    sec = (key != bestkey) ? fminf(sec, loser) : sec;
    bestkey = better ? key : bestkey;
    best = better ? t : best;
}


__device__ __forceinline__ void m3d_consider(M3dBest& B, const float4& c4, float ux, float uy, float uz) {
    const float ex = ux - c4.x, ey = uy - c4.y, ez = uz - c4.z;
    const float dd = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
    const uint32_t oi = __float_as_uint(c4.w) & M3D_IDX_MASK;
    if (B.found < 0 || dd < B.d2 || (dd == B.d2 && oi < B.oi)) { B.found = 1; B.d2 = dd; B.oi = oi; B.q = c4; }
}

__device__ __forceinline__ int m3d_nn27(const M3dLevelDev& L, float ux, float uy, float uz, float dmax2, float& out_d2, float4& out_q) {
    const M3dGrid& g = L.g;
    M3dQuery Q;
    if (!m3d_query_setup(g, ux, uy, uz, Q)) return -1;
    const int b0x = Q.lo[0] >> 1, b0y = Q.lo[1] >> 1, b0z = Q.lo[2] >> 1;
    const int nbx = (Q.hi[0] >> 1) - b0x, nby = (Q.hi[1] >> 1) - b0y, nbz = (Q.hi[2] >> 1) - b0z;   // 0 or 1 each
    const uint4* tab = reinterpret_cast<const uint4*>(L.htab);
    // phase 1: issue every probe (independent loads)
    uint4 lo[8]; uint32_t slot[8], key[8]; bool act[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const int ox = b & 1, oy = (b >> 1) & 1, oz = b >> 2;
        act[b] = (ox <= nbx) && (oy <= nby) && (oz <= nbz);
        key[b] = m3d_bucket_key(g, b0x + ox, b0y + oy, b0z + oz);
        slot[b] = m3d_hash_slot(key[b], g.hshift);
        lo[b] = make_uint4(M3D_INVALID_KEY, 0u, 0u, 0u);
        if (act[b]) lo[b] = tab[2 * (size_t)slot[b]];
    }
    // phase 2: linear probing for the rare collisions
#pragma unroll
    for (int b = 0; b < 8; b++) {
        if (act[b]) {
            while (lo[b].x != key[b] && lo[b].x != M3D_INVALID_KEY) { slot[b] = (slot[b] + 1) & g.hmask; lo[b] = tab[2 * (size_t)slot[b]]; }
        }
    }
    // phase 3: walk the neighbourhood voxels of the hit buckets
    M3dBest B; B.found = -1; B.d2 = 3.0e38f; B.oi = 0; B.q = make_float4(0.f, 0.f, 0.f, 0.f);
    float bound = dmax2 * 1.0001f;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        if (!act[b] || lo[b].x != key[b]) continue;
        const uint4 hi = tab[2 * (size_t)slot[b] + 1];
        const int vx0 = 2 * (b0x + (b & 1)), vy0 = 2 * (b0y + ((b >> 1) & 1)), vz0 = 2 * (b0z + (b >> 2));
#pragma unroll
        for (int sub = 0; sub < 8; sub++) {
            const int vx = vx0 + (sub & 1), vy = vy0 + ((sub >> 1) & 1), vz = vz0 + (sub >> 2);
            if (vx < Q.lo[0] || vx > Q.hi[0] || vy < Q.lo[1] || vy > Q.hi[1] || vz < Q.lo[2] || vz > Q.hi[2]) continue;
            if (m3d_voxel_lb2(Q, vx, vy, vz) > bound) continue;
            const uint2 rg = m3d_sub_range(lo[b], hi, L.bigcum, sub);
            for (uint32_t t = rg.x; t < rg.y; t++) m3d_consider(B, L.pts[t], ux, uy, uz);
            bound = fminf(bound, B.d2 * 1.0001f);
        }
    }
    if (B.found < 0 || !(B.d2 <= dmax2)) return -1;
    out_d2 = B.d2;
    out_q = B.q;
    return 1;
}

__device__ __forceinline__ int m3d_quant(float term, float scale) {
    return (int)rintf(term * scale);
}
#include "m3d_acc.h"   // M3dAcc32 / M3dAcc64: a thread's running sums
#ifdef M3D_ACC64   // A/B build: 64-bit running sums per thread (rounds 1-2)
#define M3D_ACC M3dAcc64
#else
#define M3D_ACC M3dAcc32
#endif

// slot of H(k,l), k <= l, in the row-major upper triangle
__host__ __device__ constexpr int hslot21(int k, int l) { return k * 6 - (k * (k - 1)) / 2 + (l - k); }

// (to_lds: the block's sums go to that LDS array and nowhere else; add_lds: sums parked that way are added)
template <int NACC, typename ACC>
__device__ __forceinline__ void block_reduce_to_global(const ACC& acc, long long* __restrict__ sums, long long* __restrict__ partial = nullptr,
                                                       long long* to_lds = nullptr, const long long* add_lds = nullptr) {
    __shared__ long long red[ICP_WAVES][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Transposed butterfly: at every step a lane gives away half of the values it still holds and adds the
    // partner's copy of the half it keeps, so the wave reduces all (up to 32) sums with 16+8+4+2+1+1 = 32
    // 64-bit shuffles instead of 6 per sum (174 for 29 sums). The shuffles go through the CU's single LDS
    // pipeline, which all its waves share: they were half of the reduction pass.
    long long v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = (i < NACC) ? acc.wide(i) : 0ll;
#pragma unroll
    for (int h = 16; h >= 1; h >>= 1) {
        const bool up = (lane & (2 * h)) != 0;     // h = 16 pairs with lane ^ 32, ... h = 1 with lane ^ 2
#pragma unroll
        for (int k = 0; k < h; k++) {
            const long long send = up ? v[k] : v[k + h];
            const long long keep = up ? v[k + h] : v[k];
            v[k] = keep + __shfl_xor(send, 2 * h);
        }
    }
    v[0] += __shfl_xor(v[0], 1);
    const int slot = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 1) == 0 && slot < NACC) red[wave][slot] = v[0];
    __syncthreads();
    if (threadIdx.x < NACC) {
        long long v = 0;
#pragma unroll
        for (int w = 0; w < ICP_WAVES; w++) v += red[w][threadIdx.x];
        if (add_lds) v += add_lds[threadIdx.x];
        if (to_lds) { to_lds[threadIdx.x] = v; return; }
        // `partial`: this block's own slot, summed by the solve kernel (integer sums: any order gives the same bits).
        // Atomics on the pair's 29 shared words serialise at ~50-100 ns each; with ~50 blocks per pair they were
        // a quarter of the reduction pass.
        if (partial) __hip_atomic_store(&partial[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through: read by another block (maybe another XCD) in the same launch
        else if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&sums[threadIdx.x]), (unsigned long long)v);
    }
}

// a7: contribution of one correspondence (u matched to q, normal nq) to the running sums
template <int METRIC, int NACC, typename ACC>
__device__ __forceinline__ void m3d_accumulate_match(ACC& acc, float ux, float uy, float uz, const float4& q, float d2,
                                                     const float4& nq, float cx, float cy, float cz, const float (&S)[6]) {
    const float ex = ux - q.x, ey = uy - q.y, ez = uz - q.z;
    const float wx = ux - cx, wy = uy - cy, wz = uz - cz;
    if constexpr (METRIC == 1) {
        const float nx = nq.x, ny = nq.y, nz = nq.z;
        if (nx == 0.0f && ny == 0.0f && nz == 0.0f) return;   // no usable normal: match rejected
        float Jv[6];
        Jv[0] = wy * nz - wz * ny; Jv[1] = wz * nx - wx * nz; Jv[2] = wx * ny - wy * nx;
        Jv[3] = nx; Jv[4] = ny; Jv[5] = nz;
        const float r = nx * ex + ny * ey + nz * ez;
        // H(k,l), k <= l, at hslot21(k, l) (row-major upper triangle; scales: rot-rot S[0], rot-trans S[1], trans-trans S[2]); then g, ssr, count.
        // The scale of a term is folded into ONE of its factors: S = 2^e, so (a * b) * S and (a * S) * b are the same float wherever neither is subnormal
        // or overflows — and a product that is subnormal before the scaling is below 2^-96 after it: both forms quantise to 0. 13 multiplies for 29.
        const float a0 = Jv[0] * S[0], a1 = Jv[1] * S[0], a2 = Jv[2] * S[0];     // rot-rot
        const float b0 = Jv[0] * S[1], b1 = Jv[1] * S[1], b2 = Jv[2] * S[1];     // rot-trans
        const float c3 = Jv[3] * S[2], c4 = Jv[4] * S[2], c5 = Jv[5] * S[2];     // trans-trans
        const float r3 = r * S[3], r4 = r * S[4];
        acc.template add<0>((int)rintf(a0 * Jv[0]));
        acc.template add<1>((int)rintf(a0 * Jv[1]));
        acc.template add<2>((int)rintf(a0 * Jv[2]));
        acc.template add<3>((int)rintf(b0 * Jv[3]));
        acc.template add<4>((int)rintf(b0 * Jv[4]));
        acc.template add<5>((int)rintf(b0 * Jv[5]));
        acc.template add<6>((int)rintf(a1 * Jv[1]));
        acc.template add<7>((int)rintf(a1 * Jv[2]));
        acc.template add<8>((int)rintf(b1 * Jv[3]));
        acc.template add<9>((int)rintf(b1 * Jv[4]));
        acc.template add<10>((int)rintf(b1 * Jv[5]));
        acc.template add<11>((int)rintf(a2 * Jv[2]));
        acc.template add<12>((int)rintf(b2 * Jv[3]));
        acc.template add<13>((int)rintf(b2 * Jv[4]));
        acc.template add<14>((int)rintf(b2 * Jv[5]));
        acc.template add<15>((int)rintf(c3 * Jv[3]));
        acc.template add<16>((int)rintf(c3 * Jv[4]));
        acc.template add<17>((int)rintf(c3 * Jv[5]));
        acc.template add<18>((int)rintf(c4 * Jv[4]));
        acc.template add<19>((int)rintf(c4 * Jv[5]));
        acc.template add<20>((int)rintf(c5 * Jv[5]));
        acc.template add<21>((int)rintf(Jv[0] * r3));
        acc.template add<22>((int)rintf(Jv[1] * r3));
        acc.template add<23>((int)rintf(Jv[2] * r3));
        acc.template add<24>((int)rintf(Jv[3] * r4));
        acc.template add<25>((int)rintf(Jv[4] * r4));
        acc.template add<26>((int)rintf(Jv[5] * r4));
        acc.template add<27>(m3d_quant(r * r, S[5]));
        acc.template add<28>(1);
    } else {
        // 17 running sums: Hrr(6) | sum w (3) | g(6) | ssr | count
        acc.template add<0>(m3d_quant(wy * wy + wz * wz, S[0]));
        acc.template add<1>(m3d_quant(-(wx * wy), S[0]));
        acc.template add<2>(m3d_quant(-(wx * wz), S[0]));
        acc.template add<3>(m3d_quant(wx * wx + wz * wz, S[0]));
        acc.template add<4>(m3d_quant(-(wy * wz), S[0]));
        acc.template add<5>(m3d_quant(wx * wx + wy * wy, S[0]));
        acc.template add<6>(m3d_quant(wx, S[1]));
        acc.template add<7>(m3d_quant(wy, S[1]));
        acc.template add<8>(m3d_quant(wz, S[1]));
        acc.template add<9>(m3d_quant(wy * ez - wz * ey, S[3]));
        acc.template add<10>(m3d_quant(wz * ex - wx * ez, S[3]));
        acc.template add<11>(m3d_quant(wx * ey - wy * ex, S[3]));
        acc.template add<12>(m3d_quant(ex, S[4]));
        acc.template add<13>(m3d_quant(ey, S[4]));
        acc.template add<14>(m3d_quant(ez, S[4]));
        acc.template add<15>(m3d_quant(d2, S[5]));
        acc.template add<16>(1);
    }
}

// the pose is the same for every lane of a pair's workgroups: keep it in scalar registers (12 VGPRs less per kernel;
// the search kernels are latency-bound, so their throughput is their occupancy)
__device__ __forceinline__ float m3d_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ void m3d_load_pose(const M3dPairState* st, float (&R)[9], float (&tt)[3]) {
    // current pose rounded to float (spec: R row-major from the column-major double pose)
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) R[3 * r + c] = m3d_uniform((float)st->T[c * 4 + r]);
        tt[r] = m3d_uniform((float)st->T[12 + r]);
    }
}

__device__ __forceinline__ void m3d_map_block(int n_pairs, int bpp, int& pair, int& blk, int rot = 0) {
    // Workgroups are dealt round-robin over the 8 XCDs by their linear id. Two maps (round 6, profiles/r06_map_spread.txt):
    //   rot >= 0 (the caller's latency mode, m3dreg_set_latency_mode: this batch has the GPU to itself) and a batch of eight pairs: the blocks of one pair are kept on
    //     ONE XCD — its L2 then holds one pair's clouds, not all eight: serial 8-pair steps 5500 registrations/s against 5200 with the other map. rot (per handle)
    //     rotates which XCD the k-th pair lands on.
    //   rot < 0 (the default: other batches share the GPU): every pair's blocks go over ALL eight XCDs. A pair that is slower than the rest of its batch (a surface
    //     close to the sensor) then loads the whole chip instead of being the one XCD every launch of the batch waits for: the eight LPT shards of config 4 run
    //     3 - 9 % faster each, steps with rotating inputs +3.8 %, one 64-pair call +1 %, the headline's own eight pairs +0.4 % (same box, interleaved).
    M3D_ENTRY_JITTER();
    const int id = blockIdx.x;
    if (rot >= 0 && n_pairs == 8) { const int slot = id >> 3; pair = (id + rot) & 7; blk = slot; }   // (one 64-pair call is 1.3 % faster with the other map)
    else { pair = id / bpp; blk = id % bpp; }
}


// ---- variant 2 (default): the two stages as two kernels -------------------------------------------
// k_nn_search: a6 only. One thread per query, nothing but the search state in registers, so the kernel
// runs at high occupancy and its dependent gathers overlap across many waves; it writes one int32 per
// query (sorted position of the match, -1 = none). k_accumulate_matches: a7 as a pure streaming
// reduction over (source point, match) pairs. The extra traffic is 8 B per query and iteration.
// per-axis conservative gap from the query to the voxel range [v0, v1] of its neighbourhood
__device__ __forceinline__ float m3d_axis_gap(int ic, int v0, int v1, float gl, float gh) {
    return (v1 < ic) ? gl : ((v0 > ic) ? gh : 0.f);
}

// Search state kept deliberately small (high occupancy). `g` is a by-value copy of the grid (SGPRs),
// tab/pts/bigcum are global-address-space pointers (global_load, not flat_load).
// Per-query state kept between the iterations of a level:
//   match[i] (int32)  >= 0: sorted position of the match; -1: no match; -2: no match AND the 27 voxels
//                     around the query's voxel held no point at all — cache[i] then holds that voxel
//   cache[i] (int64)  the voxel (ic.x | ic.y << 16 | ic.z << 32, each + 1) the "-2" verdict was made for.
// While a query stays in the same voxel its neighbourhood is the same set of (static) target voxels, so
// "-2" is answered again without a single probe. Exact: a changed voxel simply re-runs the full search.
#define M3D_NN_NONE_CACHED (-2)
#define M3D_NN_PENDING (-3)   // a query k_nn_iter<lean> could not hand to a tile: walked by the reduction pass's workgroup that owns it
#define M3D_LATE_QPT 7      // queries per thread at most in k_icp_late / k_accumulate_matches<.., true> (launch_iteration checks; m3d_acc_blocks sizes the grid for it)
#define M3D_LATE_CAP (M3D_LATE_QPT * 256)   // their worklist: every query a workgroup owns, if it must (k_icp_late: 7 x 256 x 20 B + the reductions' arrays = 39 KB of LDS)
#define M3D_TILE_CHUNK 512            // records per work item of k_nn_tiles (one per thread) when the batch has enough of them to fill the GPU; else 256 (M3dNnArgs::tile_chunk: 2 lanes per record; M3DREG_TILE_CHUNK forces 512 / 256 / 128)
#define M3D_TILE_CHUNK_CROWDED 64     // ... at least, of a tile with crowded voxels (m3d_tile_records_per_item: 256 / 128 / 64 by the tile's largest voxel)
#define M3D_NN_HEAVY (-2147483647 - 1)   // internal: the light path hands this query to the compacted full search
__device__ __forceinline__ long long m3d_voxel_code(const M3dQuery& Q) {
    return (long long)(Q.ic[0] + 1) | ((long long)(Q.ic[1] + 1) << 16) | ((long long)(Q.ic[2] + 1) << 32);
}

// ---- the search of variant 2 -----------------------------------------------------------------------------------
// Per-query state kept between the iterations of a level (see above): match[i], cache[i], state[i] = {u0, sec}.
//
// One walk routine serves every case. Order of the walk (the RESULT is order-independent: exact argmin, ties to the lowest
// input index; only the amount of work depends on it):
//   * buckets: the one holding the query's own voxel first (XOR enumeration of the 2x2x2), then its neighbours;
//   * rows of a bucket: the (y,z) row nearest to the query first.
// In dense regions the home voxel row already yields a bound of a few centimetres, after which nearly every other row
// is discarded by its box distance without a single gather. Bucket entries are fetched one ahead (software pipeline)
// instead of all eight up front: 16 VGPRs instead of 64, so the kernel fits 8 waves per SIMD — the walk is a chain of
// dependent gathers, its throughput is the number of waves in flight.
// bkey = (d2 bits << 32 | input index) of the best candidate so far, initially (+inf, ~0): every real candidate is smaller.
// sec  = BITS of a lower bound of d2 over every candidate that is not the winner (second-best seen, boxes of pruned rows);
//        non-negative floats order like their bit patterns, so it is maintained with integer min/max (no canonicalisation
//        instructions, no NaN cases).
#ifdef M3D_BLOCKTIME
struct M3dWalk { unsigned long long bkey; int best; float bound; uint32_t sec; bool any_point; uint32_t bt_trips = 0, bt_chunks = 0, bt_probes = 0; };
#else
struct M3dWalk { unsigned long long bkey; int best; float bound; uint32_t sec; bool any_point; };
#endif
#define M3D_INF_BITS 0x7F800000u
__device__ __forceinline__ void m3d_walk_init(M3dWalk& W, float dmax2) {
    W.bkey = ((unsigned long long)M3D_INF_BITS << 32) | 0xFFFFFFFFull; W.best = -1; W.bound = dmax2 * 1.0001f; W.sec = M3D_INF_BITS; W.any_point = false;
}

// LDS-resident target points (k_nn_tiles): address space 3, so the loads are ds_read_b128 and never flat
typedef const __attribute__((address_space(3))) m3d_f32x4* m3d_lf4;
__device__ __forceinline__ float4 m3d_ld(m3d_lf4 p, size_t i) { const m3d_f32x4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }

// candidates [t, t1) of the sorted target points against the query: exact argmin on the packed (d2, input index) key.
template <int G = 4>
__device__ __forceinline__ void m3d_scan_range(m3d_gf4 pts, uint32_t t, const uint32_t t1, float ux, float uy, float uz, M3dWalk& W, int sit) {
    for (; t < t1; t += G) {
        M3D_BT_COUNT(W, bt_trips);
        M3D_STAT(sit, 12);
        M3D_STATW(sit, 14);
        // G (four; the dense cooperative walk: eight) independent 16-B gathers per wait; slots past the end of the run re-read its last point and count as +inf
        const uint32_t last = t1 - 1u - t;   // >= 0
        uint32_t idx[G]; float4 c4[G];
#pragma unroll
        for (int j = 0; j < G; j++) { idx[j] = t + min((uint32_t)j, last); c4[j] = m3d_ld(pts, idx[j]); }
#pragma unroll
        for (int j = 0; j < G; j++) {
            const float ex = ux - c4[j].x, ey = uy - c4[j].y, ez = uz - c4[j].z;
            const float dd = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            const uint32_t db = (j == 0 || (uint32_t)j <= last) ? __float_as_uint(dd) : M3D_INF_BITS;
            const unsigned long long key = ((unsigned long long)db << 32) | __float_as_uint(c4[j].w);   // .w = input index
            W.sec = min(W.sec, max(db, (uint32_t)(W.bkey >> 32)));   // the loser of (candidate, best so far) is a non-winner
            const bool better = key < W.bkey;
            W.bkey = better ? key : W.bkey;
            W.best = better ? (int)idx[j] : W.best;
        }
    }
}

// rows k in [k0, k1) of one bucket, k enumerating the 4 (y,z) rows nearest-first.
// DEFER (the cooperative walk, 8 lanes per query): a crowded row is not walked by the one lane that owns its bucket — the (at most four) crowded rows of
// a bucket are noted in D and shared by the eight lanes of the group afterwards (m3d_coop_query).
struct M3dDefer { uint32_t b0, e0, b1, e1, b2, e2, b3, e3; int n; };
template <bool DEFER = false>
__device__ __forceinline__ void m3d_walk_rows(const M3dQuery& Q, const uint4& lo, const uint4& hi, m3d_gu32 bigcum, m3d_gf4 pts, m3d_gf4 cbox, int vx0, int vy0,
                                              int vz0, float ux, float uy, float uz, M3dWalk& W, int k0, int k1, int sit = 0, M3dDefer* D = nullptr) {
    const int sx0 = max(Q.lo[0] - vx0, 0), sx1 = min(Q.hi[0] - vx0, 1);
    const int sy0 = max(Q.lo[1] - vy0, 0), sy1 = min(Q.hi[1] - vy0, 1);
    const int sz0 = max(Q.lo[2] - vz0, 0), sz1 = min(Q.hi[2] - vz0, 1);
    if (sx0 > sx1) return;
    const int ny = min(max(Q.ic[1] - vy0, 0), 1), nz = min(max(Q.ic[2] - vz0, 0), 1);   // nearest row of this bucket
    const float gx = m3d_axis_gap(Q.ic[0], vx0 + sx0, vx0 + sx1, Q.gl[0], Q.gh[0]);
    const float gx2 = gx * gx;
    const uint32_t base = lo.y;
    const unsigned long long cumA = ((unsigned long long)hi.y << 32) | hi.x, cumB = ((unsigned long long)hi.w << 32) | hi.z;
    for (int k = k0; k < k1; k++) {
        M3D_STATW(sit, 21);
        const int sy = (k & 1) ^ ny, sz = (k >> 1) ^ nz;
        if (sy < sy0 || sy > sy1 || sz < sz0 || sz > sz1) continue;
        const int s_first = sx0 | (sy << 1) | (sz << 2), s_last = sx1 | (sy << 1) | (sz << 2);
        uint32_t c0, c1;
        if (lo.w == 0) {
            c1 = (uint32_t)(((s_last < 4) ? cumA : cumB) >> (16 * (s_last & 3))) & 0xFFFFu;
            const int sm = s_first - 1;
            c0 = s_first ? ((uint32_t)(((sm < 4) ? cumA : cumB) >> (16 * (sm & 3))) & 0xFFFFu) : 0u;
        } else {
            const M3D_GLOBAL uint32_t* bc = bigcum + 8 * (size_t)(lo.w - 1);
            c1 = bc[s_last];
            c0 = s_first ? bc[s_first - 1] : 0u;
        }
        W.any_point = W.any_point || (c1 > c0);
        M3D_STATW(sit, 18);
        M3D_STAT(sit, 9);
        if (c1 > c0) M3D_STAT(sit, 10);
        const float gy = m3d_axis_gap(Q.ic[1], vy0 + sy, vy0 + sy, Q.gl[1], Q.gh[1]);
        const float gz = m3d_axis_gap(Q.ic[2], vz0 + sz, vz0 + sz, Q.gl[2], Q.gh[2]);
        const float lb2 = gx2 + gy * gy + gz * gz;
        if (lb2 > W.bound) { W.sec = min(W.sec, __float_as_uint(lb2)); if (c1 > c0) M3D_STAT(sit, 11); continue; }
        if (c1 > c0) M3D_STATV(sit, 13, c1 - c0);
        M3D_STATW(sit, 19);
        if (c1 - c0 <= (uint32_t)M3D_LONG_ROW) {
            m3d_scan_range(pts, base + c0, base + c1, ux, uy, uz, W, sit);
        } else if (DEFER && D->n < 4) {
            const uint32_t rb = base + c0, re = base + c1;
            if (D->n == 0) { D->b0 = rb; D->e0 = re; } else if (D->n == 1) { D->b1 = rb; D->e1 = re; } else if (D->n == 2) { D->b2 = rb; D->e2 = re; } else { D->b3 = rb; D->e3 = re; }
            D->n++;
        } else {
            // A crowded row (a surface close to the sensor): chunk by chunk, each chunk's exact box first — the points of a voxel keep
            // their input (firing) order, which sweeps the surface strip by strip, so all but the one or two chunks around the query
            // are provably farther than the best so far and are never gathered. A skipped chunk's box distance bounds its points in `sec`.
            const uint32_t tb = base + c0, te = base + c1;
            const uint32_t cl = (te - 1u) / M3D_CHUNK;
            uint32_t c = tb / M3D_CHUNK;
            float4 bmn = m3d_ld(cbox, 2 * (size_t)c), bmx = m3d_ld(cbox, 2 * (size_t)c + 1);
            for (; c <= cl; c++) {
                const float4 mn = bmn, mx = bmx;
                M3D_BT_COUNT(W, bt_chunks);
                if (c < cl) { bmn = m3d_ld(cbox, 2 * (size_t)c + 2); bmx = m3d_ld(cbox, 2 * (size_t)c + 3); }   // the next box is in flight while this chunk is looked at
                const float dx = fmaxf(fmaxf(mn.x - ux, ux - mx.x), 0.f), dy = fmaxf(fmaxf(mn.y - uy, uy - mx.y), 0.f), dz = fmaxf(fmaxf(mn.z - uz, uz - mx.z), 0.f);
                const float bd = dx * dx + dy * dy + dz * dz;
                if (bd > W.bound) { W.sec = min(W.sec, __float_as_uint(bd)); M3D_STAT(sit, 15); continue; }
                m3d_scan_range(pts, max(tb, c * M3D_CHUNK), min(te, (c + 1u) * M3D_CHUNK), ux, uy, uz, W, sit);
                W.bound = fminf(W.bound, m3d_key_d2(W.bkey) * 1.0001f);
            }
        }
        W.bound = fminf(W.bound, m3d_key_d2(W.bkey) * 1.0001f);
    }
}

// Start of a seeded walk: the previous match is known to lie inside the neighbourhood (closer than one voxel edge), so its
// distance (measured by the query's owner while classifying) bounds the search and shrinks the neighbourhood before any probe. It is NOT entered as a candidate: its own row
// can only be discarded after a strictly closer point was found (box distance of its row <= its distance <= bound), so the
// walk meets it again as an ordinary candidate whenever it can still win — and `sec` never sees the winner twice.
__device__ __forceinline__ void m3d_walk_seed_dd(M3dQuery& Q, float dd, M3dWalk& W) {
    W.bound = fminf(W.bound, dd * 1.0001f);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (Q.gl[a] * Q.gl[a] > W.bound) { Q.lo[a] = max(Q.lo[a], Q.ic[a]); W.sec = min(W.sec, __float_as_uint(Q.gl[a] * Q.gl[a])); }
        if (Q.gh[a] * Q.gh[a] > W.bound) { Q.hi[a] = min(Q.hi[a], Q.ic[a]); W.sec = min(W.sec, __float_as_uint(Q.gh[a] * Q.gh[a])); }
    }
}

__device__ __forceinline__ int m3d_walk_result(const M3dWalk& W, float dmax2, bool seeded) {
    if (W.best >= 0 && m3d_key_d2(W.bkey) <= dmax2) return W.best;
    return (seeded || W.any_point) ? -1 : M3D_NN_NONE_CACHED;
}

// ONE QUERY PER LANE (long worklists, first iteration of a level): throughput-shaped
__device__ __forceinline__ int m3d_nn27_walk(const M3dGrid& g, m3d_gu4 tab, m3d_gf4 pts, m3d_gf4 cbox, m3d_gu32 bigcum, float ux, float uy, float uz, float dmax2,
                                             bool seeded, float dseed, long long& code_out, float& sec, int sit = 0) {
    M3dQuery Q;
    if (!m3d_query_setup(g, ux, uy, uz, Q)) return -1;
    code_out = m3d_voxel_code(Q);
    M3dWalk W; m3d_walk_init(W, dmax2);
    if (seeded) m3d_walk_seed_dd(Q, dseed, W);
    if (Q.lo[0] <= Q.hi[0] && Q.lo[1] <= Q.hi[1] && Q.lo[2] <= Q.hi[2]) {
        const int b0x = Q.lo[0] >> 1, b0y = Q.lo[1] >> 1, b0z = Q.lo[2] >> 1;
        const int nbx = (Q.hi[0] >> 1) - b0x, nby = (Q.hi[1] >> 1) - b0y, nbz = (Q.hi[2] >> 1) - b0z;   // 0 or 1 each
        const int hx = min(max((Q.ic[0] >> 1) - b0x, 0), nbx), hy = min(max((Q.ic[1] >> 1) - b0y, 0), nby), hz = min(max((Q.ic[2] >> 1) - b0z, 0), nbz);
        const int nb = (nbx + 1) * (nby + 1) * (nbz + 1);
        const int shy = nbx, shz = nbx + nby;                 // bit positions of the y / z choice inside the bucket counter
        // bucket bi of the walk: offsets (bi's bits) XOR (home bucket's offsets): bi = 0 is the home bucket
        uint32_t key_n = m3d_bucket_key(g, b0x + hx, b0y + hy, b0z + hz);
        uint32_t slot_n = m3d_hash_slot(key_n, g.hshift);
        uint4 lo_n = m3d_ld(tab, 2 * (size_t)slot_n), hi_n = m3d_ld(tab, 2 * (size_t)slot_n + 1);
        for (int bi = 0; bi < nb; bi++) {
            M3D_STATW(sit, 16);
            uint4 lo = lo_n, hi = hi_n; const uint32_t key = key_n; uint32_t slot = slot_n;
            const int ox = (bi & nbx) ^ hx, oy = ((bi >> shy) & nby) ^ hy, oz = ((bi >> shz) & nbz) ^ hz;
            if (bi + 1 < nb) {   // software pipelining: the next bucket's entry is in flight while this one is walked
                const int b2 = bi + 1;
                key_n = m3d_bucket_key(g, b0x + ((b2 & nbx) ^ hx), b0y + (((b2 >> shy) & nby) ^ hy), b0z + (((b2 >> shz) & nbz) ^ hz));
                slot_n = m3d_hash_slot(key_n, g.hshift);
                lo_n = m3d_ld(tab, 2 * (size_t)slot_n); hi_n = m3d_ld(tab, 2 * (size_t)slot_n + 1);
            }
            if (lo.x != key && lo.x != M3D_INVALID_KEY) {   // rare: linear probing past a collision
                do { slot = (slot + 1) & g.hmask; lo = m3d_ld(tab, 2 * (size_t)slot); } while (lo.x != key && lo.x != M3D_INVALID_KEY);
                hi = m3d_ld(tab, 2 * (size_t)slot + 1);
            }
            M3D_STAT(sit, 7);
            M3D_BT_COUNT(W, bt_probes);
            if (lo.x != key) continue;
            M3D_STAT(sit, 8);
            M3D_STATW(sit, 17);
            m3d_walk_rows<false>(Q, lo, hi, bigcum, pts, cbox, 2 * (b0x + ox), 2 * (b0y + oy), 2 * (b0z + oz), ux, uy, uz, W, 0, 4, sit);
        }
    }
    sec = __uint_as_float(W.sec);
    M3D_BT_FLUSH(W);
    return m3d_walk_result(W, dmax2, seeded);
}

// ONE QUERY PER LANE, EVERYTHING IN LDS (k_nn_tiles). The staged tile is addressed by VOXEL: an LDS hash {voxel key, first LDS
// position | population << 16}; the key is linear in the voxel coordinates, so a neighbour's key is the query voxel's key plus a
// constant. The 27 voxels are visited home first, then faces, edges, corners (compile-time order, fully unrolled): a voxel whose box
// is provably farther than the best so far costs two adds and a compare — after the home voxel that is nearly all of them; the rest
// cost one LDS probe and one ds_read_b128 per candidate. None of the per-bucket / per-row bookkeeping of the global walk
// (variable 64-bit shifts of the cumulative counts, row clipping, 4-wide gather batches with masked slots) exists here: the walk of
// the hash-of-buckets layout cost ~5000 instructions per wave of 64 queries and was VALU-bound even with every operand in LDS.
// Exactly the candidates the spec names are compared (a voxel is skipped only when its box distance exceeds 1.0001 x the best
// squared distance so far, the same test as the global walk's rows), with the same arithmetic: same exact argmin.
// Returns the LDS position of the match, -1, or M3D_NN_NONE_CACHED; sec = squared lower bound of every non-winning candidate.
typedef uint32_t m3d_u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) m3d_u32x2* m3d_lu2;
// candidates t, t + step, ... < t1 of the staged points, two per trip (a repeated last one counts as +inf)
__device__ __forceinline__ void m3d_tile_scan(m3d_lf4 sp, uint32_t t, const uint32_t t1, float ux, float uy, float uz, unsigned long long& bkey, int& best, uint32_t& sec,
                                              const uint32_t step = 1u) {
    for (; t < t1; t += 2u * step) {
        const uint32_t tb = (t + step < t1) ? t + step : t;
        const float4 ca = m3d_ld(sp, t), cb = m3d_ld(sp, tb);
        {
            const float ex = ux - ca.x, ey = uy - ca.y, ez = uz - ca.z;
            const uint32_t db = __float_as_uint(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
            const unsigned long long k = ((unsigned long long)db << 32) | __float_as_uint(ca.w);
            sec = min(sec, max(db, (uint32_t)(bkey >> 32)));
            const bool better = k < bkey;
            bkey = better ? k : bkey; best = better ? (int)t : best;
        }
        {
            const float ex = ux - cb.x, ey = uy - cb.y, ez = uz - cb.z;
            const uint32_t db = (tb != t) ? __float_as_uint(fmaf(ez, ez, fmaf(ey, ey, ex * ex))) : M3D_INF_BITS;
            const unsigned long long k = ((unsigned long long)db << 32) | __float_as_uint(cb.w);
            sec = min(sec, max(db, (uint32_t)(bkey >> 32)));
            const bool better = k < bkey;
            bkey = better ? k : bkey; best = better ? (int)tb : best;
        }
    }
}
// slot of voxel `key` in the directory (linear probing past the rare collision), or an empty slot
__device__ __forceinline__ m3d_u32x2 m3d_tile_find(m3d_lu2 vs, uint32_t key) {
    uint32_t h = (key * 0x9E3779B1u) >> (32 - 11);
    m3d_u32x2 s = vs[h];
    while (s.x != key && s.x != M3D_INVALID_KEY) { h = (h + 1u) & (M3D_TILE_VS - 1u); s = vs[h]; }
    return s;
}
// one staged voxel (directory value sv: first LDS position | population - 1 | staged bucket, m3d_device.h) against the query: lane `sub` of the
// `step` lanes that share the query scans every step-th point of it; bsv = the directory value of the voxel the best candidate came from
__device__ __forceinline__ void m3d_tile_voxel(m3d_lf4 sp, uint32_t sv, float ux, float uy, float uz, unsigned long long& bkey, int& best, uint32_t& sec, uint32_t& bsv, const uint32_t sub, const uint32_t step) {
    const uint32_t t = M3D_TILE_SV_POS(sv), n = M3D_TILE_SV_CNT(sv);
    const unsigned long long k0 = bkey;
    m3d_tile_scan(sp, t + sub, t + n, ux, uy, uz, bkey, best, sec, step);
    bsv = (bkey != k0) ? sv : bsv;
}
// phase 1 of the search, voxel B of the compile-time visiting order at offset (DX, DY, DZ): one unconditional directory probe (no
// branch: 26 of them are in flight together); the voxel enters the lane's work mask when it exists, is not provably farther than
// the bound the home voxel left, and its first slot holds it (or something else: phase 2 then probes on)
// (the slot of voxel key0 + delta: (key0 + delta) * C = key0 * C + delta * C mod 2^32 — h0 = key0 * C once per query, delta * C is wave-uniform
// scalar arithmetic (c1 = C << sh1, c2 = C << sh2): one add instead of a 32-bit multiply, which issues at a quarter of the rate, per probe)
template <int B, int DX, int DY, int DZ>
__device__ __forceinline__ void m3d_tile_probe(m3d_lu2 vs, const float (&G)[3][3], uint32_t h0, uint32_t c1, uint32_t c2, float bound, uint32_t& sec, uint32_t& mask) {
    const float lb = G[0][DX + 1] + G[1][DY + 1] + G[2][DZ + 1];
    const uint32_t hd = h0 + ((uint32_t)DX * 0x9E3779B1u + (uint32_t)DY * c1 + (uint32_t)DZ * c2);
    const bool near = !(lb > bound);
    const m3d_u32x2 s = vs[hd >> (32 - 11)];   // (round 6: pruned lanes reading one common slot instead — a broadcast — −0.5 %; not reading at all — a branch per probe: 23 spilled registers — −13 %)
    sec = near ? sec : min(sec, __float_as_uint(lb));          // a pruned voxel bounds its points (+inf = outside the grid: no-op)
    mask |= (near && s.x != M3D_INVALID_KEY) ? (1u << B) : 0u;
}
// per-query search state carried across the images of a tile
struct M3dTileQ {
    float G[3][3];              // squared conservative gaps to the voxels at offset -1 / 0 / +1 per axis; +inf = outside the grid
    uint32_t key0;              // key of the query's voxel (modular arithmetic: exact for every voxel inside the grid)
    unsigned long long bkey;    // (d2 bits << 32 | input index) of the best candidate so far
    uint32_t sec;               // bits of a squared lower bound of every non-winning candidate
    float bound;                // prune voxels whose box is farther than this
};
// voxel code of a binned query (m3d_voxel_code's encoding; k_nn_iter filed it: -1 <= f <= dims)
__device__ __forceinline__ long long m3d_tile_code(const M3dGrid& g, float ux, float uy, float uz) {
    const int i0 = (int)m3d_cell_f(ux, g.mn[0], g.inv_leaf), i1 = (int)m3d_cell_f(uy, g.mn[1], g.inv_leaf), i2 = (int)m3d_cell_f(uz, g.mn[2], g.inv_leaf);
    return (long long)(i0 + 1) | ((long long)(i1 + 1) << 16) | ((long long)(i2 + 1) << 32);
}
__device__ __forceinline__ void m3d_tile_query(const M3dGrid& g, float ux, float uy, float uz, float dmax2, bool seeded, float dseed, M3dTileQ& Q) {
    const float u[3] = { ux, uy, uz };
    int ic[3];
    const float inf = __uint_as_float(M3D_INF_BITS);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float f = m3d_cell_f(u[a], g.mn[a], g.inv_leaf);     // k_nn_iter filed the query: -1 <= f <= dims
        ic[a] = (int)f;
        const float r = (u[a] - g.mn[a]) - f * g.leaf;
        const float gl = fmaxf(r - g.prune_slack, 0.f), gh = fmaxf((g.leaf - r) - g.prune_slack, 0.f);
        Q.G[a][0] = (ic[a] >= 1) ? gl * gl : inf;                       // (ic - 1 <= dims - 1 always)
        Q.G[a][1] = (ic[a] >= 0 && ic[a] < g.dims[a]) ? 0.f : inf;
        Q.G[a][2] = (ic[a] + 1 < g.dims[a]) ? gh * gh : inf;            // (ic + 1 >= 0 always)
    }
    const int sh1 = g.cb[0] + 1, sh2 = g.cb[0] + g.cb[1] + 2;
    Q.key0 = (uint32_t)ic[0] + ((uint32_t)ic[1] << sh1) + ((uint32_t)ic[2] << sh2);
    Q.bkey = ((unsigned long long)M3D_INF_BITS << 32) | 0xFFFFFFFFull;
    Q.sec = M3D_INF_BITS;
    Q.bound = dmax2 * 1.0001f;
    if (seeded) Q.bound = fminf(Q.bound, dseed * 1.0001f);   // the previous match lies within these 27 voxels: it bounds the search before the first probe
}
// one staged image: returns the SORTED position (in the level's pts array) of a NEW best candidate, or -1 when the best so far stands — a staged bucket is
// one contiguous run in LDS and in the sorted order, delta[bucket] = sorted position - LDS position of its points (bucket.hip: tile_build_role).
// step > 1: `step` consecutive lanes (a power of two, at most 8) answer ONE query together — a query in a crowded stretch compares
// against a thousand and more candidates while the scans are still centimetres apart (no box is provably farther than a
// neighbour that far away): every lane scans its share of each voxel, the group agrees on the bound after the home voxel and on
// the result at the end (xor-shuffles). All lanes of a group enter with the same Q and leave with the same Q.
__device__ __forceinline__ int m3d_tile_search(const M3dGrid& g, m3d_lu2 vs, m3d_lf4 sp, const int* kdelta, const int* delta, float ux, float uy, float uz, M3dTileQ& Q,
                                               const uint32_t sub = 0u, const uint32_t step = 1u) {
    const int sh1 = g.cb[0] + 1, sh2 = g.cb[0] + g.cb[1] + 2;
    int best = -1;
    uint32_t bsv = 0u;
    const unsigned long long bkey0 = Q.bkey;
    // home voxel: every lane, no divergence; leaves the bound that prunes most of the other 26
    if (Q.G[0][1] + Q.G[1][1] + Q.G[2][1] == 0.f) {
        const m3d_u32x2 s = m3d_tile_find(vs, Q.key0);
        if (s.x == Q.key0) m3d_tile_voxel(sp, s.y, ux, uy, uz, Q.bkey, best, Q.sec, bsv, sub, step);
        Q.bound = fminf(Q.bound, m3d_key_d2(Q.bkey) * 1.0001f);
    }
    if (step > 1u) for (uint32_t o = 1u; o < step; o <<= 1) Q.bound = fminf(Q.bound, __shfl_xor(Q.bound, (int)o));
    uint32_t mask = 0u;
    const uint32_t h0 = Q.key0 * 0x9E3779B1u;
    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0x9E3779B1u << sh1)), c2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0x9E3779B1u << sh2));   // (uniform: scalar registers)
#define M3D_P(b, dx, dy, dz) m3d_tile_probe<b, dx, dy, dz>(vs, Q.G, h0, c1, c2, Q.bound, Q.sec, mask)
#ifndef M3D_EXP_NOPROBE   // (timing experiment only — WRONG results: what would the search cost without the 26 blind probes?)
    M3D_P(1, -1, 0, 0); M3D_P(2, 1, 0, 0); M3D_P(3, 0, -1, 0); M3D_P(4, 0, 1, 0); M3D_P(5, 0, 0, -1); M3D_P(6, 0, 0, 1);
    M3D_P(7, -1, -1, 0); M3D_P(8, 1, -1, 0); M3D_P(9, -1, 1, 0); M3D_P(10, 1, 1, 0);
    M3D_P(11, -1, 0, -1); M3D_P(12, 1, 0, -1); M3D_P(13, -1, 0, 1); M3D_P(14, 1, 0, 1);
    M3D_P(15, 0, -1, -1); M3D_P(16, 0, 1, -1); M3D_P(17, 0, -1, 1); M3D_P(18, 0, 1, 1);
    M3D_P(19, -1, -1, -1); M3D_P(20, 1, -1, -1); M3D_P(21, -1, 1, -1); M3D_P(22, 1, 1, -1);
    M3D_P(23, -1, -1, 1); M3D_P(24, 1, -1, 1); M3D_P(25, -1, 1, 1); M3D_P(26, 1, 1, 1);
#endif
#undef M3D_P
#ifdef M3D_EXP_NOPHASE2   // (timing experiment only — WRONG results: the 26 probes are made, the surviving voxels are not visited)
    asm volatile("" :: "v"(mask));
    mask = 0u;
#endif
    // phase 2: the voxels that survived, nearest kinds first (faces, edges, corners), each scanned in full. (Re-testing every voxel against the
    // bound its predecessors left — the offsets from an LDS table, the gaps selected out of G — was tried in round 3: bit-identical and 7 %
    // SLOWER; the re-test costs every visit ~15 instructions and a dependent LDS read, and prunes little: most of what survives phase 1 is
    // genuinely close. Throw-away builds, iteration 0 of the bench batch: 10 us of the launch are records + staging + results, 11 the home
    // voxel, 7 the 26 probes, 23 this loop.)
    // (software-pipelined: the NEXT voxel's key offset and first directory slot are requested before the current voxel's points are compared — the
    // kernel waits for LDS round trips, not for the VALU, and a visit was three dependent ones: offset table, directory slot, points)
    if (mask) {
        int b = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        uint32_t key = Q.key0 + (uint32_t)kdelta[b];
        uint32_t h = (key * 0x9E3779B1u) >> (32 - 11);
        m3d_u32x2 s = vs[h];
        for (;;) {
            const bool more = mask != 0u;
            uint32_t key_n = 0u, h_n = 0u; m3d_u32x2 s_n = (m3d_u32x2){ M3D_INVALID_KEY, 0u };
            if (more) {
                const int b2 = __ffs((int)mask) - 1;
                mask &= mask - 1u;
                key_n = Q.key0 + (uint32_t)kdelta[b2];
                h_n = (key_n * 0x9E3779B1u) >> (32 - 11);
                s_n = vs[h_n];
            }
            while (s.x != key && s.x != M3D_INVALID_KEY) { h = (h + 1u) & (M3D_TILE_VS - 1u); s = vs[h]; }   // (linear probing past the rare collision)
#ifdef M3D_EXP_NOSCAN2   // (timing experiment only — WRONG results: the surviving voxels are looked up but their points are not compared)
            if (s.x == key) Q.sec = min(Q.sec, s.y);
#else
            if (s.x == key) m3d_tile_voxel(sp, s.y, ux, uy, uz, Q.bkey, best, Q.sec, bsv, sub, step);
#endif
            if (!more) break;
            key = key_n; h = h_n; s = s_n;
        }
    }
    if (best >= 0) best += delta[M3D_TILE_SV_BKT(bsv)];   // LDS position -> sorted position
    if (step > 1u) {
        // the group's result: the smallest key; every other lane's NEW key lost to it (the old best, where a lane kept it, was
        // already counted as a loser by the lane that beat it)
        unsigned long long kmin = Q.bkey;
        for (uint32_t o = 1u; o < step; o <<= 1) { const unsigned long long ok = __shfl_xor(kmin, (int)o); kmin = ok < kmin ? ok : kmin; }
        uint32_t secg = (Q.bkey != kmin && Q.bkey != bkey0) ? min(Q.sec, (uint32_t)(Q.bkey >> 32)) : Q.sec;
        int bestg = (Q.bkey == kmin && Q.bkey != bkey0) ? best : -1;
        for (uint32_t o = 1u; o < step; o <<= 1) { secg = min(secg, (uint32_t)__shfl_xor((int)secg, (int)o)); bestg = max(bestg, __shfl_xor(bestg, (int)o)); }
        Q.bkey = kmin; Q.sec = secg; best = bestg;
    }
    Q.bound = fminf(Q.bound, m3d_key_d2(Q.bkey) * 1.0001f);
    return best;
}

// ---- the per-query record kept between the iterations of a level: 8 bytes {match, certificate word} (rounds 1-3: a 4-byte match and a 16-byte
// state {u0.xyz, sec} in two arrays — 20 bytes read per query and iteration by the classifying kernels, two scattered sectors written per answered search).
// certificate word = bits of `sec` (a non-negative float: 31 bits) >> 4, << 5 | iteration of the query's last real search & 31. Dropping the four low
// mantissa bits rounds the lower bound DOWN: valid, 2^-19 weaker. Where the query was at that search, u0, is not stored: it is R0 p + t0 with the
// float pose of that iteration, which the pair's pose RING keeps for 32 iterations (written by k_patch_jobs for the first iteration of a batch and by
// the solve for every following one) — the same fma chain on the same floats: the same bits as the stored u0 of rounds 1-3. A record whose slot
// is the running iteration's own is 32 iterations old (the slot has been overwritten): it certifies nothing and the query searches again.
typedef int m3d_i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int m3d_cert_pack(float sec, int itq) { return (int)(((__float_as_uint(sec) >> 4) << 5) | (uint32_t)itq); }
__device__ __forceinline__ int m3d_cert_pack_bits(uint32_t sec_bits, int itq) { return (int)(((sec_bits >> 4) << 5) | (uint32_t)itq); }
// {u0.xyz, sec} of a record (ring: the pair's pose ring, LDS or global; p: the query's source point); a stale record gets sec = 0 (certifies nothing)
template <typename RING>
__device__ __forceinline__ m3d_f32x4 m3d_cert_state(int cert, int itq, RING ring, const float4& p) {
    const int k0 = cert & 31;
    const float* r = &ring[12 * k0];
    const float x = fmaf(r[0], p.x, fmaf(r[1], p.y, fmaf(r[2], p.z, r[9])));
    const float y = fmaf(r[3], p.x, fmaf(r[4], p.y, fmaf(r[5], p.z, r[10])));
    const float z = fmaf(r[6], p.x, fmaf(r[7], p.y, fmaf(r[8], p.z, r[11])));
    const float sec = (k0 == itq) ? 0.f : __uint_as_float(((uint32_t)cert >> 5) << 4);
    return (m3d_f32x4){ x, y, z, sec };
}
#define M3D_RING_FLOATS (32 * 12)
// the pair's pose ring into LDS (every thread calls it; the caller's next barrier publishes it)
__device__ __forceinline__ void m3d_ring_to_lds(float* s_ring, const float* ring_global) {
    if (threadIdx.x < M3D_RING_FLOATS / 4) reinterpret_cast<float4*>(s_ring)[threadIdx.x] = reinterpret_cast<const float4*>(ring_global)[threadIdx.x];
}

struct M3dNnArgs {
    int2* match; int match_stride;     // per pair: {match, certificate word} of every query (see the encoding above; m3d_cert_pack)
    long long* cache;                  // per pair: voxel code of the "-2" verdicts (same stride as match)
    const float* ring;                 // per pair [32][12]: the float poses {R row-major, t} of the last 32 iterations of the pair (slot = iteration & 31): where a
                                       //           query WAS at its last real search is recomputed from the pose of that iteration (NN certificate)
    int certify;                       // 1 = use the NN certificates (default); 0 = always search (A/B, M3DREG_CERTIFY)
    float seed_reach;                  // seeds farther than this many voxel edges are not used (<= 0.99)
    M3dPairState* states;              // [n_pairs] == jobs[pair].st: addressed from the kernel argument, so the pose loads do not wait for the job's
    int lane_min;                      // a block with at least this many queries to search walks one query per lane, else 8 lanes per query
    int rot;                           // XCD rotation of the block -> pair map (m3d_map_block)
    int coop_kernel;                   // 1 = k_nn_coop follows this launch: the pairs whose target level is dense (M3dJob::coop_always) are ITS work, k_nn_iter leaves them alone;
                                       // 2 = k_nn_coop is the only search kernel of the iteration (every pair of the handle's last batch was dense at this level)
    // the LDS-staged search: a block with many queries to search BINS them by the target tile that owns their home bucket
    // (k_nn_iter), k_nn_tiles then answers every tile's queries from LDS
    int tiles;                         // 1 = on
    int ntile_max;                     // tiles per pair the workspace is laid out for (>= tiles of every target of the batch)
    int tile_chunk;                    // records per work item of a tile without crowded voxels: 512, or 256 when the batch is small (launch_iteration)
    float4* rec;                       // per pair [ntile_max][M3D_TILE_QCAP]: query records {u.xyz, bits(query | seeded << 31)} of the tiles (a query that cannot be
                                       //   filed — tile flagged, slab full — is walked in global memory by k_nn_iter<false> itself or left M3D_NN_PENDING by k_nn_iter<true>)
    float* recd;                       // same layout: squared distance to the seed (the previous match)
    unsigned long long rec_stride;     // records per pair
    unsigned int* tcnt;                // per pair [ntile_max]: records per tile (zero between iterations)
    int cnt_stride;
    uint2* witems;                     // [wcap] work items of k_nn_tiles {pair, tile | record chunk << 20}, published by k_nn_iter
    unsigned int* wcount;              // items published (zero between iterations)
    int wcap;
};

#define NN_SETUP()                                                                                          \
    int pair, blk;                                                                                          \
    m3d_map_block(n_pairs, bpp, pair, blk, A.rot);                                                          \
    const M3dJob& J = jobs[pair];                                                                           \
    const M3dPairState* st = A.states + pair;   /* == J.st, without waiting for the job descriptor */      \
    if (st->done || (!first_of_level && st->level_done)) return;                                            \
    float R[9], tt[3];                                                                                      \
    m3d_load_pose(st, R, tt);                                                                               \
    const M3dGrid g = J.tgt.g;                                                                              \
    const m3d_gu4 tab = m3d_as_global(reinterpret_cast<const uint4*>(J.tgt.htab));                          \
    const m3d_gf4 pts = m3d_as_global(J.tgt.pts);                                                           \
    const m3d_gf4 cbox = m3d_as_global(J.tgt.cbox);                                                         \
    const m3d_gu32 bigcum = m3d_as_global(J.tgt.bigcum);                                                    \
    const m3d_gf3 src = m3d_as_global3(J.src);                                                              \
    const float dmax2 = J.dmax2;                                                                            \
    const int n = J.n_src;                                                                                  \
    const int itq = st->iters & 31;             /* this iteration's slot of the pose ring */                \
    M3D_GLOBAL m3d_i32x2* out = (M3D_GLOBAL m3d_i32x2*)(void M3D_GLOBAL*)(A.match + (size_t)pair * A.match_stride);     \
    M3D_GLOBAL long long* cache = (M3D_GLOBAL long long*)(void M3D_GLOBAL*)(A.cache + (size_t)pair * A.match_stride);

// k_nn_iter: the whole correspondence step of one iteration in ONE launch. A block owns 256 consecutive queries of a pair.
//   1. classify (no search): a query whose previous match is still PROVABLY the exact argmin (NN certificate) is done, so is
//      a cached "nothing within the 27 voxels" verdict while the query stays in its voxel; the rest needs a search — seeded
//      with the previous match when that is still closer than one voxel edge (then it lies inside the new neighbourhood).
//   2. search: a block with many queries left (>= lane_min: the first iterations) walks one query per lane, each thread
//      its own query; a block with few compacts them into an LDS worklist (wave64 ballots) and walks them 8 LANES PER
//      QUERY — one bucket of the 2x2x2 per lane, the nearest row of every bucket first, then a bound exchange, then the
//      other rows — so the block's latency is a handful of dependent waits however sparse its list is.
// No global worklists, no atomics, no scan: what the earlier three-kernel chain (classify / scan / search) exchanged through
// HBM stays inside the block.
// Classification of ONE query (no search). mp = its result of the previous iteration.
// Returns 0 = answered (certified, cached "none", or outside the grid: `out` is corrected when it has to change),
// 1 = needs a seeded search, 2 = needs a full search; dseed = squared distance to the previous match (cls 1, certified);
// certified = the previous match stands (q1 = that point, for the caller's residual).
// (s0 = state[i] and q1 = pts[mp] are passed in: callers that handle several queries per thread load them for all of them first)
__device__ __forceinline__ int m3d_classify_loaded(const M3dGrid& g, M3D_GLOBAL m3d_i32x2* out, const M3D_GLOBAL long long* cache, int i, int mp, float ux,
                                                   float uy, float uz, float dmax2, int certify, float seed_reach, const m3d_f32x4& s0,
                                                   const float4& q1, float& dseed, bool& certified, int sit) {
    certified = false;
    if (!m3d_finite3(ux, uy, uz)) { if (mp != -1) out[i].x = -1; return 0; }
    const float f1x = m3d_cell_f(ux, g.mn[0], g.inv_leaf), f1y = m3d_cell_f(uy, g.mn[1], g.inv_leaf), f1z = m3d_cell_f(uz, g.mn[2], g.inv_leaf);
    const bool in_range = (f1x >= -1.0f && f1x <= (float)g.dims[0]) && (f1y >= -1.0f && f1y <= (float)g.dims[1]) &&
                          (f1z >= -1.0f && f1z <= (float)g.dims[2]);
    if (!in_range) { if (mp != -1) out[i].x = -1; return 0; }
    if (mp == M3D_NN_NONE_CACHED) {
        const long long code = (long long)((int)f1x + 1) | ((long long)((int)f1y + 1) << 16) | ((long long)((int)f1z + 1) << 32);
        const int cls = (code == cache[i]) ? 0 : 2;
        if (cls == 0) M3D_STAT(sit, 2);
        return cls;
    }
    if (mp < 0) return 2;
    // NN certificate: at its last real search (position u0) every candidate other than the match was at
    // least sqrt(sec) away [inside the 27 voxels] / dout away [outside them]. The query has moved by
    // delta since, so those are still farther than (bound - delta); if the match's CURRENT distance is
    // below that, with margins far above float rounding, it is provably still the exact argmin.
    const float ex = ux - q1.x, ey = uy - q1.y, ez = uz - q1.z;
    const float dd1 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
    const float mx = ux - s0.x, my = uy - s0.y, mz = uz - s0.z;
    const float delta = sqrtf(mx * mx + my * my + mz * mz);
    const float f0x = m3d_cell_f(s0.x, g.mn[0], g.inv_leaf), f0y = m3d_cell_f(s0.y, g.mn[1], g.inv_leaf), f0z = m3d_cell_f(s0.z, g.mn[2], g.inv_leaf);
    const bool same_voxel = (f0x == f1x) && (f0y == f1y) && (f0z == f1z);
    const float r0x = (s0.x - g.mn[0]) - f0x * g.leaf, r0y = (s0.y - g.mn[1]) - f0y * g.leaf, r0z = (s0.z - g.mn[2]) - f0z * g.leaf;
    const float gm = fmaxf(fminf(fminf(fminf(r0x, g.leaf - r0x), fminf(r0y, g.leaf - r0y)), fminf(r0z, g.leaf - r0z)) - g.prune_slack, 0.f);
    const float dout = 0.999f * g.leaf + gm;
    const float reach = seed_reach * g.leaf;
    const bool seedable = dd1 < reach * reach;   // closer than one voxel edge => inside the (new) neighbourhood
    const float others = same_voxel ? sqrtf(s0.w) : fminf(sqrtf(s0.w), dout);   // same voxel => same 27 voxels => only `sec` matters
    certified = certify && (same_voxel || seedable) && (dd1 <= dmax2) &&
                (others * 0.9999f > sqrtf(dd1) * 1.0001f + delta * 1.0001f + 1.0e-6f * g.leaf);
    dseed = dd1;
    if (certified) M3D_STAT(sit, 1);
    return certified ? 0 : (seedable ? 1 : 2);
}
// one query per thread: e = its record {match, certificate word} (loaded by the caller beside the source point p), s_ring = the pair's pose ring in LDS
__device__ __forceinline__ int m3d_classify(const M3dGrid& g, m3d_gf4 pts, M3D_GLOBAL m3d_i32x2* out, const M3D_GLOBAL long long* cache,
                                            const float* s_ring, int itq, const float4& p, int i, const m3d_i32x2 e, float ux, float uy, float uz, float dmax2,
                                            int certify, float seed_reach, float& dseed, bool& certified, float4& q1, int sit) {
    const int mp = e.x;
    m3d_f32x4 s0 = (m3d_f32x4){ 0.f, 0.f, 0.f, 0.f };
    q1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mp >= 0) { q1 = m3d_ld(pts, (size_t)M3D_CHK(104, mp, g.n_valid)); s0 = m3d_cert_state(e.y, itq, s_ring, p); }
    return m3d_classify_loaded(g, out, cache, i, mp, ux, uy, uz, dmax2, certify, seed_reach, s0, q1, dseed, certified, sit);
}

// ---- the crowded rows of a query on a DENSE level (k_nn_coop: a map's coarse levels, > 48 points per voxel), by the 8 lanes of its group ------------------
// The rows the lanes set aside (at most 4 each) go into a table in LDS and are treated as ONE list of chunks, 64 at a time:
//   A  lane j loads the boxes of chunks j, j + 8, ... of the block (eight independent 32-B loads, two waits) and files the ones that are not
//      provably farther than the bound in the group's survivor list (ballot + popcount: same slots in every run);
//   B  the group picks the survivor whose box is nearest and compares its 16 points together, two per lane — one gather trip leaves a
//      bound that is close to the final one;
//   C  lane j takes survivors j, j + 8, ...: those still inside the bound are gathered eight points per wait, the others only leave their
//      box distance in `sec`; the group agrees on the bound after every round.
// Rounds 1-3 walked row after row, every lane its chunks j, j + 8, ... one after the other (box, wait, four gather trips of four, wait ...):
// 40-60 dependent round trips per query at config 5's 0.4 m level, where this takes ~10. The candidates compared are the same kind of
// set as before — a chunk is skipped only against a bound that some compared candidate met, every point is compared by exactly one lane —
// so the argmin (and its tie rule) is the spec's; `sec` stays a lower bound of every non-winner.
#define M3D_DENSE_SURV 64
struct M3dCoopLds { uint2 rows[32]; uint2 surv[M3D_DENSE_SURV]; };   // per group: {first, end} sorted position of a row; {chunk | row << 27, bits of the box distance}
__device__ __forceinline__ float m3d_group_min(float v) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float m3d_box_d2(const float4& mn, const float4& mx, float vx, float vy, float vz) {
    const float dx = fmaxf(fmaxf(mn.x - vx, vx - mx.x), 0.f), dy = fmaxf(fmaxf(mn.y - vy, vy - mx.y), 0.f), dz = fmaxf(fmaxf(mn.z - vz, vz - mx.z), 0.f);
    return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ void m3d_coop_dense_rows(m3d_gf4 pts, m3d_gf4 cbox, const M3dDefer& D, int hsub, float vx, float vy, float vz, M3dWalk& W, int sub, int g0,
                                                    M3dCoopLds* gl, int sit) {
    static_assert(M3D_CHUNK == 16, "two points per lane in step B, two gather trips of eight in step C");
    const float inf = __uint_as_float(M3D_INF_BITS);
    // the table of rows, the home bucket's lane first (its rows hold the answer most of the time): deterministic slots
    const int rot = (sub - hsub) & 7;                  // this lane's place in the order
    int rbase = 0;
    uint32_t T = 0;                                    // chunks of all rows together
    {
        const int nd = D.n;
        uint32_t nch = 0;
        if (nd > 0) nch += (D.e0 - 1u) / M3D_CHUNK - D.b0 / M3D_CHUNK + 1u;
        if (nd > 1) nch += (D.e1 - 1u) / M3D_CHUNK - D.b1 / M3D_CHUNK + 1u;
        if (nd > 2) nch += (D.e2 - 1u) / M3D_CHUNK - D.b2 / M3D_CHUNK + 1u;
        if (nd > 3) nch += (D.e3 - 1u) / M3D_CHUNK - D.b3 / M3D_CHUNK + 1u;
#pragma unroll
        for (int k = 0; k < 8; k++) {                  // (8 shuffles each: lane order -> rotated order)
            const int ndk = __shfl(nd, g0 + ((hsub + k) & 7));
            const uint32_t nck = (uint32_t)__shfl((int)nch, g0 + ((hsub + k) & 7));
            if (k < rot) rbase += ndk;
            T += nck;
        }
        if (nd > 0) gl->rows[rbase] = make_uint2(D.b0, D.e0);
        if (nd > 1) gl->rows[rbase + 1] = make_uint2(D.b1, D.e1);
        if (nd > 2) gl->rows[rbase + 2] = make_uint2(D.b2, D.e2);
        if (nd > 3) gl->rows[rbase + 3] = make_uint2(D.b3, D.e3);
    }
    if (T == 0u) return;                               // (uniform inside a group)
    W.any_point = true;
    __builtin_amdgcn_wave_barrier();
    // this lane's cursor into the list of chunks: row, its first chunk, the list index of that chunk, its chunk count
    int row = 0;
    uint2 rr = gl->rows[0];
    uint32_t rc0 = rr.x / M3D_CHUNK, rn = (rr.y - 1u) / M3D_CHUNK - rc0 + 1u, rp = 0u;
#pragma unroll 1
    for (uint32_t blk0 = 0; blk0 < T; blk0 += 64u) {   // (uniform inside a group)
        uint32_t ns = 0;                               // survivors of this block (uniform inside a group)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t code[4]; float4 mn[4], mx[4]; bool val[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t i = blk0 + 8u * (uint32_t)(4 * half + j) + (uint32_t)sub;
                val[j] = i < T;
                uint32_t c = rc0;
                if (val[j]) {
                    while (i >= rp + rn) { rp += rn; row++; rr = gl->rows[row]; rc0 = rr.x / M3D_CHUNK; rn = (rr.y - 1u) / M3D_CHUNK - rc0 + 1u; }
                    c = rc0 + (i - rp);
                }
                code[j] = c | ((uint32_t)row << 27);
                mn[j] = m3d_ld(cbox, 2 * (size_t)c); mx[j] = m3d_ld(cbox, 2 * (size_t)c + 1);   // (an idle slot re-reads a box of the lane's row)
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                M3D_BT_COUNT(W, bt_chunks);
                const float bd = m3d_box_d2(mn[j], mx[j], vx, vy, vz);
                const bool live = val[j] && !(bd > W.bound);
                if (val[j] && !live) W.sec = min(W.sec, __float_as_uint(bd));
                const uint32_t bits = (uint32_t)(__ballot(live) >> g0) & 0xFFu;
                if (live) gl->surv[ns + (uint32_t)__popc(bits & ((1u << sub) - 1u))] = make_uint2(code[j], __float_as_uint(bd));
                ns += (uint32_t)__popc(bits);
            }
        }
        if (ns == 0u) continue;
        __builtin_amdgcn_wave_barrier();
        // B: the nearest survivor, by all eight lanes
        uint32_t e_done = 0xFFFFFFFFu;
        {
            float mb = inf; uint32_t me = 0u;
            for (uint32_t e = (uint32_t)sub; e < ns; e += 8u) { const float bd = __uint_as_float(gl->surv[e].y); if (bd < mb) { mb = bd; me = e; } }
            const float gm = m3d_group_min(mb);
            const uint32_t who = (uint32_t)(__ballot(mb == gm && mb < inf) >> g0) & 0xFFu;
            if (who != 0u && !(gm > W.bound)) {
                e_done = (uint32_t)__shfl((int)me, g0 + (__ffs((int)who) - 1));
                const uint32_t cd = gl->surv[e_done].x;
                const uint2 r2 = gl->rows[cd >> 27];
                const uint32_t c = cd & 0x7FFFFFFu;
                const uint32_t s0 = max(r2.x, c * M3D_CHUNK) + 2u * (uint32_t)sub, e0 = min(r2.y, (c + 1u) * M3D_CHUNK);
                if (s0 < e0) m3d_scan_range<4>(pts, s0, min(s0 + 2u, e0), vx, vy, vz, W, sit);
                W.bound = m3d_group_min(fminf(W.bound, m3d_key_d2(W.bkey) * 1.0001f));
            }
        }
        // C: the other survivors, one per lane and round
#pragma unroll 1
        for (uint32_t e00 = 0; e00 < ns; e00 += 8u) {
            const uint32_t e = e00 + (uint32_t)sub;
            if (e < ns && e != e_done) {
                const uint2 sv = gl->surv[e];
                const float bd = __uint_as_float(sv.y);
                if (bd > W.bound) W.sec = min(W.sec, sv.y);
                else {
                    const uint2 r2 = gl->rows[sv.x >> 27];
                    const uint32_t c = sv.x & 0x7FFFFFFu;
                    m3d_scan_range<8>(pts, max(r2.x, c * M3D_CHUNK), min(r2.y, (c + 1u) * M3D_CHUNK), vx, vy, vz, W, sit);
                }
            }
            W.bound = m3d_group_min(fminf(W.bound, m3d_key_d2(W.bkey) * 1.0001f));
        }
        __builtin_amdgcn_wave_barrier();   // (the list is rewritten by the next block)
    }
}

// One query walked by the 8 lanes of a group (sub = lane & 7 holds one bucket of the 2x2x2): nearest row of every bucket,
// bound exchange, the other rows, shuffle merge. Every lane of the wave must call it (act = this group has a query);
// returns the match (>= 0), -1 or M3D_NN_NONE_CACHED on every lane of the group; sec / code for the state arrays.
template <bool DENSE = false>
__device__ __forceinline__ int m3d_coop_query(const M3dGrid& g, m3d_gu4 tab, m3d_gf4 pts, m3d_gf4 cbox, m3d_gu32 bigcum, float dmax2, bool act, bool seeded,
                                              float vx, float vy, float vz, float dseed, int sub, long long& code, float& sec, int sit, M3dCoopLds* gl = nullptr) {
    M3dQuery Q;
    M3dWalk W; m3d_walk_init(W, dmax2);
    M3dDefer D; D.b0 = D.e0 = D.b1 = D.e1 = D.b2 = D.e2 = D.b3 = D.e3 = 0u; D.n = 0;
    bool ok = false, found = false;
    code = 0;
    uint4 lo = make_uint4(M3D_INVALID_KEY, 0u, 0u, 0u), hi = make_uint4(0u, 0u, 0u, 0u);
    int vx0 = 0, vy0 = 0, vz0 = 0;
    if (act) {
        ok = m3d_query_setup(g, vx, vy, vz, Q);   // finite by classification
        if (ok) {
            code = m3d_voxel_code(Q);
            if (seeded) m3d_walk_seed_dd(Q, dseed, W);   // every lane of the group starts from the seed's bound
            if (Q.lo[0] <= Q.hi[0] && Q.lo[1] <= Q.hi[1] && Q.lo[2] <= Q.hi[2]) {
                const int b0x = Q.lo[0] >> 1, b0y = Q.lo[1] >> 1, b0z = Q.lo[2] >> 1;
                const int nbx = (Q.hi[0] >> 1) - b0x, nby = (Q.hi[1] >> 1) - b0y, nbz = (Q.hi[2] >> 1) - b0z;
                const int nb = (nbx + 1) * (nby + 1) * (nbz + 1);
                if (sub < nb) {   // this lane's bucket
                    const int shy = nbx, shz = nbx + nby;
                    const int ox = sub & nbx, oy = (sub >> shy) & nby, oz = (sub >> shz) & nbz;
                    const uint32_t key = m3d_bucket_key(g, b0x + ox, b0y + oy, b0z + oz);
                    uint32_t slot = m3d_hash_slot(key, g.hshift);
                    lo = m3d_ld(tab, 2 * (size_t)slot); hi = m3d_ld(tab, 2 * (size_t)slot + 1);   // both halves in one round trip
                    if (lo.x != key && lo.x != M3D_INVALID_KEY) {   // rare: linear probing past a collision
                        do { slot = (slot + 1) & g.hmask; lo = m3d_ld(tab, 2 * (size_t)slot); } while (lo.x != key && lo.x != M3D_INVALID_KEY);
                        hi = m3d_ld(tab, 2 * (size_t)slot + 1);
                    }
                    if (lo.x == key) {
                        found = true;
                        vx0 = 2 * (b0x + ox); vy0 = 2 * (b0y + oy); vz0 = 2 * (b0z + oz);
                    }
                }
            }
        }
    }
    // nearest row of every bucket (the home voxel's row among them), then the group agrees on the bound ...
    if (act) { M3D_STAT(sit, 7); if (found) M3D_STAT(sit, 8); }
#pragma unroll 1
    for (int phase = 0; phase < 2; phase++) {   // (one copy of the row walk in the code: a rolled loop, not two inlined calls)
        if (found) m3d_walk_rows<true>(Q, lo, hi, bigcum, pts, cbox, vx0, vy0, vz0, vx, vy, vz, W, phase, phase ? 4 : 1, sit, &D);
        // ... the group agrees on the bound; the other rows are then mostly discarded by their box distance
        float bnd = W.bound;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) bnd = fminf(bnd, __shfl_xor(bnd, o));
        W.bound = bnd;
    }
    // The crowded rows the lanes set aside: every one is shared by the eight lanes of its group — lane j tests the boxes of chunks j, j + 8, ...
    // and scans the ones that can still win. (One lane walking such a row alone made 17 to 35 dependent gather trips: a launch that
    // classifies in 9 us ended after 23 because of a handful of such queries.) Exact in any order: a chunk is skipped only against a
    // bound that some candidate already met, a skipped chunk's box distance and every loser go into `sec`.
    {
        const int lane = (int)(threadIdx.x & 63u), g0 = lane & ~7;
        // the lane that holds the query's HOME bucket first (its nearest row was set aside first): on a crowded level everything is set aside, and
        // the first chunks scanned are scanned against d_max — let them be the ones around the query. The bound is agreed on after every row.
        int hsub = 0;
        if (ok && Q.lo[0] <= Q.hi[0] && Q.lo[1] <= Q.hi[1] && Q.lo[2] <= Q.hi[2]) {
            const int b0x = Q.lo[0] >> 1, b0y = Q.lo[1] >> 1, b0z = Q.lo[2] >> 1;
            const int nbx = (Q.hi[0] >> 1) - b0x, nby = (Q.hi[1] >> 1) - b0y, nbz = (Q.hi[2] >> 1) - b0z;
            const int hx = min(max((Q.ic[0] >> 1) - b0x, 0), nbx), hy = min(max((Q.ic[1] >> 1) - b0y, 0), nby), hz = min(max((Q.ic[2] >> 1) - b0z, 0), nbz);
            hsub = hx | (hy << nbx) | (hz << (nbx + nby));
        }
        hsub = __shfl(hsub, g0);   // (group-uniform by construction; taken from the group's first lane so that an idle group agrees with itself too)
        if (DENSE) m3d_coop_dense_rows(pts, cbox, D, hsub, vx, vy, vz, W, sub, g0, gl, sit);
        else
#pragma unroll 1
        for (int L0 = 0; L0 < 8; L0++) {
            const int L = (hsub + L0) & 7;
            const int ndL = __shfl(D.n, g0 + L);
#pragma unroll 1
            for (int r = 0; r < ndL; r++) {   // (uniform inside a group)
                const uint32_t rb = r == 0 ? D.b0 : (r == 1 ? D.b1 : (r == 2 ? D.b2 : D.b3)), re = r == 0 ? D.e0 : (r == 1 ? D.e1 : (r == 2 ? D.e2 : D.e3));
                const uint32_t tb = (uint32_t)__shfl((int)rb, g0 + L), te = (uint32_t)__shfl((int)re, g0 + L);
                W.any_point = true;
                const uint32_t cl = (te - 1u) / M3D_CHUNK;
                for (uint32_t c = tb / M3D_CHUNK + (uint32_t)sub; c <= cl; c += 8u) {
                    M3D_BT_COUNT(W, bt_chunks);
                    const float4 mn = m3d_ld(cbox, 2 * (size_t)c), mx = m3d_ld(cbox, 2 * (size_t)c + 1);
                    const float dx = fmaxf(fmaxf(mn.x - vx, vx - mx.x), 0.f), dy = fmaxf(fmaxf(mn.y - vy, vy - mx.y), 0.f), dz = fmaxf(fmaxf(mn.z - vz, vz - mx.z), 0.f);
                    const float bd = dx * dx + dy * dy + dz * dz;
                    if (bd > W.bound) { W.sec = min(W.sec, __float_as_uint(bd)); continue; }
                    m3d_scan_range(pts, max(tb, c * M3D_CHUNK), min(te, (c + 1u) * M3D_CHUNK), vx, vy, vz, W, sit);
                    W.bound = fminf(W.bound, m3d_key_d2(W.bkey) * 1.0001f);
                }
                {   // (every lane of the group is here: ndL and r are uniform inside a group)
                    float bnd = W.bound;
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) bnd = fminf(bnd, __shfl_xor(bnd, o));
                    W.bound = bnd;
                }
            }
        }
    }
    M3D_BT_FLUSH(W);
    // merge the 8 lanes of the query: argmin of the keys; `sec` = min of everything that is not the winner
    // (every point lives in exactly one bucket, so two lanes never hold the same candidate)
    unsigned long long bkey = W.bkey; int best = W.best; uint32_t secb = W.sec; bool any_point = W.any_point;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned long long ok2 = __shfl_xor(bkey, o);
        const int ob = __shfl_xor(best, o);
        const uint32_t os = (uint32_t)__shfl_xor((int)secb, o);
        const bool oa = __shfl_xor((int)any_point, o) != 0;
        const bool better = ok2 < bkey;
        secb = min(min(secb, os), (uint32_t)((better ? bkey : ok2) >> 32));
        bkey = better ? ok2 : bkey;
        best = better ? ob : best;
        any_point = any_point || oa;
    }
    sec = __uint_as_float(secb);
    int m = -1;
    if (ok) {
        if (best >= 0 && m3d_key_d2(bkey) <= dmax2) m = best;
        else m = (seeded || any_point) ? -1 : M3D_NN_NONE_CACHED;
    }
    return m;
}

// Occupancy of k_nn_iter<true>: its 41 VGPRs allow 8 waves per SIMD, but left to itself the compiler takes 105 SGPRs (800 per SIMD, granules of 16: 7 waves).
// Asked for 8 it makes do with 77 (more of them parked in VGPR lanes): the kernel is a chain of dependent round trips (record, previous match, occupancy
// words, table entry, two returning atomics) whose speed is the number of waves in flight — 21.9 -> 19.8 us alone, 139.9 -> 121.2 us per 64-pair launch,
// headline +0.8 %, serial steps +1.3 % (round 5, same box). k_nn_iter<false> (128 VGPRs: the walks) is left alone.
#ifndef M3D_NN_WAVES
#define M3D_NN_WAVES 8   // (A/B builds: make CXXFLAGS+=-DM3D_NN_WAVES=7)
#endif
#define M3D_NN_OCC __attribute__((amdgpu_waves_per_eu(LEAN ? M3D_NN_WAVES : 1, 8)))
// LEAN (k_nn_iter<true>, the tile iterations of a level whose target has tiles): classify + bin ONLY — every query that must search goes
// to its tile's slab however few they are, and the rare one that cannot (a tile that could not be staged, a full slab, more than 64
// tiles in one workgroup) is left M3D_NN_PENDING for the reduction pass's workgroup that streams it. Without the two walks compiled in the kernel
// needs 41 VGPRs instead of 124: 7 (round 5: 8) waves per SIMD instead of 4 — worth 4-6 % of the headline, where three chains compete for the
// register file (alone it is only 2-5 us faster per launch).
template <bool LEAN>
__global__ __launch_bounds__(256) M3D_NN_OCC void k_nn_iter(const M3dJob* __restrict__ jobs, int n_pairs, int bpp, int first_of_level, M3dNnArgs A) {
    NN_SETUP();
    if (!LEAN && A.coop_kernel && J.coop_always) return;   // (block-uniform) a crowded level: k_nn_coop, launched right behind, answers this pair
    {   // the source's crowded blocks first (they run several times as long as the rest: started last they were the kernel's tail)
        const uint32_t* ord = J.src_order;
        if (ord && blk < J.src_nblk) blk = (int)M3D_CHK(101, ord[blk], J.src_nblk);
    }
    M3D_BT_BEGIN();
    __shared__ int s_cnt[4];
    __shared__ int s_list[LEAN ? 1 : 256];          // worklist of the cooperative walk: owner thread | seeded << 8 ...
    __shared__ float s_wu[3][LEAN ? 1 : 256];       // ... its transformed query ...
    __shared__ float s_wd[LEAN ? 1 : 256];          // ... and the squared distance to its seed (the previous match), all known to the owner
    const int tid = (int)threadIdx.x;
    const int i = blk * 256 + tid;
#ifdef M3D_STATS
    const int sit = st->iters;
#else
    const int sit = 0;
#endif
    (void)sit;
    int cls = 0;   // 0 = done, 1 = seeded search, 2 = full search
    float ux = 0.f, uy = 0.f, uz = 0.f, dseed = 0.f;
    __shared__ float s_ring[M3D_RING_FLOATS];   // the pair's pose ring (the certificates recompute where a query was at its last search)
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    m3d_i32x2 e = (m3d_i32x2){ -1, 0 };
    const m3d_gf4 prev = m3d_as_global(LEAN ? nullptr : J.prev_pts);   // (a lean registration has one level)
    if (i < n) { p = m3d_ld3(src, (size_t)i); if (!first_of_level || (!LEAN && J.prev_pts)) e = out[i]; }
    if (!first_of_level) { m3d_ring_to_lds(s_ring, A.ring + (size_t)pair * M3D_RING_FLOATS); __syncthreads(); }   // (block-uniform; the loads above are in flight across it)
    if (i < n) {
        ux = fmaf(R[0], p.x, fmaf(R[1], p.y, fmaf(R[2], p.z, tt[0])));
        uy = fmaf(R[3], p.x, fmaf(R[4], p.y, fmaf(R[5], p.z, tt[1])));
        uz = fmaf(R[6], p.x, fmaf(R[7], p.y, fmaf(R[8], p.z, tt[2])));
        if (first_of_level) {
            if (m3d_finite3(ux, uy, uz)) {
                cls = 2;
                if (!LEAN && J.prev_pts && e.x >= 0) {   // the coarser level's match seeds this level's first search when it lies inside the new neighbourhood
                    const float4 q1 = m3d_ld(prev, (size_t)M3D_CHK(105, e.x, g.n_valid));
                    const float ex = ux - q1.x, ey = uy - q1.y, ez = uz - q1.z;
                    const float dd1 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                    const float reach = A.seed_reach * g.leaf;
                    if (dd1 < reach * reach) { cls = 1; dseed = dd1; }
                }
            } else out[i].x = -1;
        } else {
            bool certified; float4 q1;
            cls = m3d_classify(g, pts, out, cache, s_ring, itq, p, i, e, ux, uy, uz, dmax2, A.certify, A.seed_reach, dseed, certified, q1, sit);
        }
    }
    // how many queries of this block need a search?
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long bW = __ballot(cls != 0);
    if (lane == 0) s_cnt[wave] = (int)__popcll(bW);
    __syncthreads();
    int offW = 0, nW = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) offW += s_cnt[w]; nW += s_cnt[w]; }
    if (i < n) { M3D_STAT(sit, 0); if (cls == 1) M3D_STAT(sit, 3); if (cls == 2) M3D_STAT(sit, 4); }
    if (nW == 0) { M3D_BT_END(0); return; }   // block-uniform
    // A crowded level (hundreds of points per voxel: the coarse levels of a dense map) is walked eight lanes per query whatever the number of
    // searchers: one lane alone makes a hundred and more dependent gather trips through its 27 voxels' chunks, eight share them (config 5's
    // 0.4 m level: 0.33 -> 0.20 ms per iteration; its 0.2 m level, 25 points per voxel, is faster one query per lane: 0.051 vs 0.063).
    const int lane_min = (!LEAN && J.coop_always) ? 257 : A.lane_min;
    if (tid == 0) M3D_STAT(sit, nW >= lane_min ? 5 : 6);
    if (LEAN || (nW >= lane_min && A.tiles && J.tgt.thdr)) {
        // ---- many queries to search, the target has tiles: bin them, k_nn_tiles answers them from LDS ----------------
        // A query goes to the tile that owns its home bucket (one probe of the level's table: the bucket's first sorted position
        // names the tile); every other bucket of its 2x2x2 neighbourhood is within one bucket of that one, hence staged with
        // the tile. A query whose home bucket is empty or outside the grid, whose tile is flagged (staged set too large) or
        // whose tile's slab is full goes to the pair's global-walk list instead. One returning atomic per wave and tile.
        const M3dTileHdr* thdr = J.tgt.thdr;
        unsigned int* tcnt = A.tcnt + (size_t)pair * A.cnt_stride;
        int tile = -2;   // -2 = nothing to file, -1 = global-walk list, >= 0 = tile
        if (cls != 0) {
            M3dQuery Q;
            if (!m3d_query_setup(g, ux, uy, uz, Q)) out[i].x = -1;
            else {
                // the 2x2x2 buckets of the (unrestricted) neighbourhood: ANY occupied one names a tile that stages them all; the home
                // bucket is probed first (it is the occupied one for two queries in three), the others together, only when it is empty
                const int b0x = Q.lo[0] >> 1, b0y = Q.lo[1] >> 1, b0z = Q.lo[2] >> 1;
                const int nbx = (Q.hi[0] >> 1) - b0x, nby = (Q.hi[1] >> 1) - b0y, nbz = (Q.hi[2] >> 1) - b0z;   // 0 or 1 each
                const int hx = min(max((Q.ic[0] >> 1) - b0x, 0), nbx), hy = min(max((Q.ic[1] >> 1) - b0y, 0), nby), hz = min(max((Q.ic[2] >> 1) - b0z, 0), nbz);
                const M3D_GLOBAL uint32_t* occ = (const M3D_GLOBAL uint32_t*)(const void M3D_GLOBAL*)J.tgt.occ;
                uint32_t hkey = M3D_INVALID_KEY;   // an occupied bucket of the 2x2x2: the one whose table entry names the tile
                bool none = false;
                if (occ) {
                    // occupancy bitmap: eight 4-byte loads from a table of a few hundred KB (the hash table's entries are 32 B in 1 MB)
                    uint32_t key[8], wbit[8];
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const int ox = b & 1, oy = (b >> 1) & 1, oz = b >> 2;
                        key[b] = m3d_bucket_key(g, b0x + min(ox, nbx), b0y + min(oy, nby), b0z + min(oz, nbz));   // (an inactive offset repeats an active bucket)
                        wbit[b] = occ[key[b] >> 5];
                    }
                    const uint32_t khome = m3d_bucket_key(g, b0x + hx, b0y + hy, b0z + hz);
#pragma unroll
                    for (int b = 7; b >= 0; b--) if ((wbit[b] >> (key[b] & 31u)) & 1u) hkey = (key[b] == khome || hkey != khome) ? key[b] : hkey;   // the home bucket when it is occupied
                    none = hkey == M3D_INVALID_KEY;
                } else {
                    uint32_t key[8], slot[8]; uint4 lo[8]; bool act[8];
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const int ox = b & 1, oy = (b >> 1) & 1, oz = b >> 2;
                        act[b] = ox <= nbx && oy <= nby && oz <= nbz;
                        key[b] = m3d_bucket_key(g, b0x + ox, b0y + oy, b0z + oz);
                        slot[b] = m3d_hash_slot(key[b], g.hshift);
                        lo[b] = make_uint4(M3D_INVALID_KEY, 0u, 0u, 0u);
                        if (act[b]) lo[b] = m3d_ld(tab, 2 * (size_t)slot[b]);
                    }
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        if (!act[b]) continue;
                        while (lo[b].x != key[b] && lo[b].x != M3D_INVALID_KEY) { slot[b] = (slot[b] + 1) & g.hmask; lo[b] = m3d_ld(tab, 2 * (size_t)slot[b]); }
                        if (lo[b].x == key[b] && hkey == M3D_INVALID_KEY) hkey = key[b];
                    }
                    none = hkey == M3D_INVALID_KEY;
                }
                if (none) {   // no occupied bucket around the query: answered here (what the walk would find: nothing, not even a point)
                    if (cls == 1) out[i].x = -1;   // (cannot happen: a seed lies in one of these buckets)
                    else { out[i].x = M3D_NN_NONE_CACHED; cache[i] = m3d_voxel_code(Q); }
                } else {
                    uint32_t slot = m3d_hash_slot(hkey, g.hshift);
                    uint4 lo = m3d_ld(tab, 2 * (size_t)slot);
                    while (lo.x != hkey) { slot = (slot + 1) & g.hmask; lo = m3d_ld(tab, 2 * (size_t)slot); }   // (the bucket exists)
                    tile = (int)M3D_CHK(107, lo.y / (uint32_t)M3D_TILE_PTS, A.ntile_max);
                }
            }
        }
        // A record's slot: ONE returning atomic per (workgroup, tile) — the queries of a workgroup are neighbours, they fall into a
        // handful of tiles. The lanes count themselves per tile in a small LDS table (rank inside the workgroup), one thread per
        // distinct tile then reserves the workgroup's records in the tile's slab (all those atomics are in flight together: a
        // per-wave loop over its distinct tiles waited for each in turn) and publishes the tile's new work items.
        __shared__ int s_tk[64];
        __shared__ unsigned int s_tn[64], s_tb[64];
        if (tid < 64) { s_tk[tid] = -1; s_tn[tid] = 0u; }
        __syncthreads();
        int slot = -1; uint32_t rank = 0;
        if (tile >= 0) {
            uint32_t h = ((uint32_t)tile * 0x9E3779B1u) >> 26;
            for (int tries = 0; tries < 64; tries++) {
                const int old = atomicCAS(&s_tk[h], -1, tile);
                if (old == -1 || old == tile) { slot = (int)h; break; }
                h = (h + 1u) & 63u;
            }
            if (slot >= 0) rank = atomicAdd(&s_tn[slot], 1u);
        }
        __syncthreads();
        if (tid < 64 && s_tk[tid] >= 0) {
            const int t0 = s_tk[tid];
            // the reservation does not wait for the tile's header (both round trips are in flight together): a flagged tile's records
            // are never looked at (no work item names them), its queries are walked below
            const M3D_GLOBAL uint32_t* hw = (const M3D_GLOBAL uint32_t*)(const void M3D_GLOBAL*)(thdr + t0);
            const uint32_t hflags = hw[2], hmeta0 = hw[3];
            const uint32_t cnt = s_tn[tid];
            uint32_t base = atomicAdd(&tcnt[t0], cnt);
            if ((hflags & M3D_TILE_OVERSIZE) != 0u) base = 0xFFFFFFFFu;
            else {
                // work items of k_nn_tiles: one per chunk of the tile's records (512, or 64 for a tile with crowded voxels, whose
                // queries cost ten times as much: they are spread over more workgroups); the append that covers a chunk's first
                // record publishes it — into one of M3D_TILE_LISTS lists (each with its counter on its own 128-B line): 2000
                // returning atomics on ONE word were 8 us of the first iteration's 30. (Lists chosen per workgroup, or no lists at
                // all — k_nn_tiles testing every (tile, chunk) against the counters itself — made k_nn_iter as fast and k_nn_tiles
                // 12-25 us slower: whatever clumps the items of a crowded tile, or hands a workgroup 3 items and its neighbour none,
                // shows up as the tail of that kernel.)
                const uint32_t cs = m3d_tile_records_per_item(hmeta0, (uint32_t)A.tile_chunk);
                const uint32_t end = min(base + cnt, (uint32_t)M3D_TILE_QCAP);
                for (uint32_t c = (base + cs - 1u) / cs; c * cs < end; c++) {
                    const uint32_t wl = ((uint32_t)t0 + c + (uint32_t)pair) & (uint32_t)(M3D_TILE_LISTS - 1);   // (per item, not per workgroup: a crowded tile's chunks spread over all lists)
                    const uint32_t w = atomicAdd(A.wcount + 32u * wl, 1u);
                    if (w < (uint32_t)A.wcap) A.witems[(size_t)wl * (size_t)A.wcap + w] = make_uint2((uint32_t)pair, (uint32_t)t0 | (c << 20));
                    else (void)M3D_CHK(108, w, A.wcap);   // (a dropped work item: its records would never be answered)
                }
            }
            s_tb[tid] = base;
        }
        __syncthreads();
        uint32_t pos = 0;
        if (tile >= 0) {
            const uint32_t base = slot >= 0 ? s_tb[slot] : 0xFFFFFFFFu;   // (more than 64 distinct tiles in one workgroup: the rest is walked below)
            pos = base + rank;
            if (base == 0xFFFFFFFFu || pos >= (uint32_t)M3D_TILE_QCAP) tile = -1;
        }
        if (tile >= 0) {
            const size_t r = (size_t)pair * A.rec_stride + (size_t)tile * M3D_TILE_QCAP + pos;
            A.rec[r] = make_float4(ux, uy, uz, __uint_as_float((uint32_t)i | (cls == 1 ? 0x80000000u : 0u)));
            A.recd[r] = dseed;
        } else if (tile == -1) {   // a flagged tile (one bucket beyond an image) or a full slab (rare)
            if constexpr (LEAN) {   // ... the reduction pass walks it (k_accumulate_matches<.., true>)
                atomicAdd(&tcnt[A.ntile_max], 1u);   // (the counter behind the tiles': "this pair has pending queries" — every workgroup of the reduction pass reads it)
                out[i].x = M3D_NN_PENDING;
            } else {                // ... walked here, in global memory
                long long code = 0; float sec = 0.f;
                const int m = m3d_nn27_walk(g, tab, pts, cbox, bigcum, ux, uy, uz, dmax2, cls == 1, dseed, code, sec, sit);
                out[i] = (m3d_i32x2){ m, m3d_cert_pack(sec, itq) };
                if (m == M3D_NN_NONE_CACHED) cache[i] = code;
                atomicAdd(&A.states[pair].ctr[1], 1u);
            }
        }
        M3D_BT_END(nW);
        return;
    }
    if constexpr (!LEAN) {
    if (nW >= lane_min) {
        // ---- one query per lane: every thread walks its own query -------------------------------------------------
        if (cls != 0) {
            long long code = 0; float sec = 0.f;
            const int m = m3d_nn27_walk(g, tab, pts, cbox, bigcum, ux, uy, uz, dmax2, cls == 1, dseed, code, sec, sit);
            out[i] = (m3d_i32x2){ m, m3d_cert_pack(sec, itq) };
            if (m == M3D_NN_NONE_CACHED) cache[i] = code;
        }
#ifdef M3D_BLOCKTIME
        __syncthreads();
#endif
        M3D_BT_END(nW);
        return;
    }
    // ---- few queries: LDS worklist, 8 lanes per query ------------------------------------------------------------
    if (cls != 0) {   // the group that walks this query starts from what its owner already knows: no reload of the source point,
        // of the previous match index or of the seed point — three dependent round trips less per pass
        const int w = offW + (int)__popcll(bW & ((1ull << lane) - 1ull));
        s_list[w] = tid | (cls == 1 ? 256 : 0);
        s_wu[0][w] = ux; s_wu[1][w] = uy; s_wu[2][w] = uz; s_wd[w] = dseed;
    }
    __syncthreads();
    const int sub = tid & 7;
    for (int base = 0; base < nW; base += 32) {   // uniform trip count: the shuffles inside need every lane
        const int q = base + (tid >> 3);
        const bool act = q < nW;
        const int e = act ? s_list[q] : 0;
        const int qi = blk * 256 + (e & 255);
        const float vx = act ? s_wu[0][q] : 0.f, vy = act ? s_wu[1][q] : 0.f, vz = act ? s_wu[2][q] : 0.f;
        long long code; float sec;
        if (tid == 0) M3D_STAT(sit, 15);
        const int m = m3d_coop_query(g, tab, pts, cbox, bigcum, dmax2, act, (e & 256) != 0, vx, vy, vz, act ? s_wd[q] : 0.f, sub, code, sec, sit);
        if (act && sub == 0) {
            out[qi] = (m3d_i32x2){ m, m3d_cert_pack(sec, itq) };
            if (m == M3D_NN_NONE_CACHED) cache[qi] = code;
        }
    }
#ifdef M3D_BLOCKTIME
    __syncthreads();
#endif
    M3D_BT_END(nW);
    }
}

// k_nn_coop: the correspondence step of a CROWDED level (M3dJob::coop_always: a coarse level of a dense map, a hundred and more points per voxel) —
// eight lanes per query like k_nn_iter's cooperative walk, but a workgroup owns 32 queries, not 256: the walk of such a query is a chain of a dozen
// dependent rounds (boxes of a voxel's chunks, then the chunks that can still win), and k_nn_iter's workgroups went through EIGHT passes of 32 queries
// one after the other while a 100 k-query level filled 1.5 waves per SIMD. Eight times as many workgroups, one pass each: config 5's 0.4 m level
// 0.175 -> see DESIGN ms per iteration. Launched behind k_nn_iter<false> on the coarser levels of a pyramid; its workgroups leave at once where the target
// level is not crowded (and k_nn_iter's where it is). Every lane of a group classifies its group's query (same addresses: one transaction).
__global__ __launch_bounds__(256) void k_nn_coop(const M3dJob* __restrict__ jobs, int n_pairs, int bpp, int first_of_level, M3dNnArgs A) {
    NN_SETUP();
    if (A.coop_kernel != 2 && !J.coop_always) return;   // (block-uniform; 2: the only search kernel of this iteration, every pair is its work)
    {   // the crowded 256-point blocks first, like k_nn_iter: this workgroup is an eighth of one
        const uint32_t* ord = J.src_order;
        const int b256 = blk >> 3;
        if (ord && b256 < J.src_nblk) blk = (int)M3D_CHK(102, ord[b256], J.src_nblk) * 8 + (blk & 7);
    }
    const int tid = (int)threadIdx.x, sub = tid & 7;
    const int i = blk * 32 + (tid >> 3);
    int cls = 0;
    float ux = 0.f, uy = 0.f, uz = 0.f, dseed = 0.f;
    __shared__ float s_ring[M3D_RING_FLOATS];
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    m3d_i32x2 e = (m3d_i32x2){ -1, 0 };
    const m3d_gf4 prev = m3d_as_global(J.prev_pts);
    if (i < n) { p = m3d_ld3(src, (size_t)i); if (!first_of_level || J.prev_pts) e = out[i]; }
    if (!first_of_level) { m3d_ring_to_lds(s_ring, A.ring + (size_t)pair * M3D_RING_FLOATS); __syncthreads(); }   // (block-uniform)
    if (i < n) {
        ux = fmaf(R[0], p.x, fmaf(R[1], p.y, fmaf(R[2], p.z, tt[0])));
        uy = fmaf(R[3], p.x, fmaf(R[4], p.y, fmaf(R[5], p.z, tt[1])));
        uz = fmaf(R[6], p.x, fmaf(R[7], p.y, fmaf(R[8], p.z, tt[2])));
        if (first_of_level) {
            if (m3d_finite3(ux, uy, uz)) {
                cls = 2;
                if (J.prev_pts && e.x >= 0) {   // (as in k_nn_iter: the coarser level's match as a seed)
                    const float4 q1 = m3d_ld(prev, (size_t)M3D_CHK(105, e.x, g.n_valid));
                    const float ex = ux - q1.x, ey = uy - q1.y, ez = uz - q1.z;
                    const float dd1 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                    const float reach = A.seed_reach * g.leaf;
                    if (dd1 < reach * reach) { cls = 1; dseed = dd1; }
                }
            } else if (sub == 0) out[i].x = -1;
        } else {
            bool certified; float4 q1;
            cls = m3d_classify(g, pts, out, cache, s_ring, itq, p, i, e, ux, uy, uz, dmax2, A.certify, A.seed_reach, dseed, certified, q1, 0);   // (the eight lanes of a group store the same correction, if any)
        }
    }
    long long code; float sec;
    __shared__ M3dCoopLds s_coop[32];
#ifdef M3D_STATS   // (scripts/walk_stats_c5.py: this launch's iteration is its statistics row)
    const int sit_c = (int)st->iters;
#else
    const int sit_c = 0;
#endif
    const int m = m3d_coop_query<true>(g, tab, pts, cbox, bigcum, dmax2, cls != 0, cls == 1, ux, uy, uz, dseed, sub, code, sec, sit_c, &s_coop[tid >> 3]);
    if (cls != 0 && sub == 0) {
        out[i] = (m3d_i32x2){ m, m3d_cert_pack(sec, itq) };
        if (m == M3D_NN_NONE_CACHED) cache[i] = code;
    }
}

// k_nn_coop_list: the same search for the iterations in which most queries of a dense level are CERTIFIED (config 5: 74 % at the end of the 0.4 m level,
// 93-98 % at 0.2 / 0.1 m): k_nn_coop gives every query a group of eight lanes whether it searches or not, and a wave with one searching group of eight
// runs the whole chain. Here a workgroup owns QPB queries, its first QPB lanes classify one each (k_nn_iter's way), the searchers are compacted into an
// LDS list (one ballot per wave) and walked 32 at a time by all 256 lanes: QPB / 32 times fewer waves when few search, QPB / 32 passes one after the other
// when all do (why the level's first iterations stay with k_nn_coop).
template <int QPB>
__global__ __launch_bounds__(256, 4) void k_nn_coop_list(const M3dJob* __restrict__ jobs, int n_pairs, int bpp, int first_of_level, M3dNnArgs A) {
    static_assert(QPB == 64 || QPB == 128 || QPB == 256, "whole waves classify");
    NN_SETUP();
    if (A.coop_kernel != 2 && !J.coop_always) return;   // (block-uniform)
    {   // the crowded 256-point blocks first
        const uint32_t* ord = J.src_order;
        constexpr int PER = 256 / QPB;
        const int b256 = blk / PER;
        if (ord && b256 < J.src_nblk) blk = (int)M3D_CHK(103, ord[b256], J.src_nblk) * PER + (blk % PER);
    }
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float s_ring[M3D_RING_FLOATS];
    __shared__ M3dCoopLds s_coop[32];
    __shared__ int s_list[QPB];
    __shared__ float s_wu[3][QPB], s_wd[QPB];
    __shared__ int s_wn[4];
    if (!first_of_level) { m3d_ring_to_lds(s_ring, A.ring + (size_t)pair * M3D_RING_FLOATS); __syncthreads(); }   // (block-uniform)
    int cls = 0;
    float ux = 0.f, uy = 0.f, uz = 0.f, dseed = 0.f;
    const int i = blk * QPB + tid;
    if (tid < QPB && i < n) {
        const float4 p = m3d_ld3(src, (size_t)i);
        m3d_i32x2 e = (m3d_i32x2){ -1, 0 };
        if (!first_of_level || J.prev_pts) e = out[i];
        ux = fmaf(R[0], p.x, fmaf(R[1], p.y, fmaf(R[2], p.z, tt[0])));
        uy = fmaf(R[3], p.x, fmaf(R[4], p.y, fmaf(R[5], p.z, tt[1])));
        uz = fmaf(R[6], p.x, fmaf(R[7], p.y, fmaf(R[8], p.z, tt[2])));
        if (first_of_level) {
            if (m3d_finite3(ux, uy, uz)) {
                cls = 2;
                if (J.prev_pts && e.x >= 0) {
                    const float4 q1 = m3d_ld(m3d_as_global(J.prev_pts), (size_t)M3D_CHK(106, e.x, g.n_valid));
                    const float ex = ux - q1.x, ey = uy - q1.y, ez = uz - q1.z;
                    const float dd1 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                    const float reach = A.seed_reach * g.leaf;
                    if (dd1 < reach * reach) { cls = 1; dseed = dd1; }
                }
            } else out[i].x = -1;
        } else {
            bool certified; float4 q1;
            cls = m3d_classify(g, pts, out, cache, s_ring, itq, p, i, e, ux, uy, uz, dmax2, A.certify, A.seed_reach, dseed, certified, q1, 0);
        }
    }
    const unsigned long long bW = __ballot(cls != 0);
    if (lane == 0) s_wn[wave] = (int)__popcll(bW);
    __syncthreads();
    int offW = 0, nW = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) offW += s_wn[w]; nW += s_wn[w]; }
    if (cls != 0) {
        const int w = offW + (int)__popcll(bW & ((1ull << lane) - 1ull));
        s_list[w] = tid | (cls == 1 ? 256 : 0);
        s_wu[0][w] = ux; s_wu[1][w] = uy; s_wu[2][w] = uz; s_wd[w] = dseed;
    }
    __syncthreads();
    const int sub = tid & 7;
    for (int base = 0; base < nW; base += 32) {   // uniform trip count: the shuffles inside need every lane
        const int q = base + (tid >> 3);
        const bool act = q < nW;
        const int e = act ? s_list[q] : 0;
        const int qi = blk * QPB + (e & 255);
        const float vx = act ? s_wu[0][q] : 0.f, vy = act ? s_wu[1][q] : 0.f, vz = act ? s_wu[2][q] : 0.f;
        long long code; float sec;
        const int m = m3d_coop_query<true>(g, tab, pts, cbox, bigcum, dmax2, act, (e & 256) != 0, vx, vy, vz, act ? s_wd[q] : 0.f, sub, code, sec, 0, &s_coop[tid >> 3]);
        if (act && sub == 0) {
            out[qi] = (m3d_i32x2){ m, m3d_cert_pack(sec, itq) };
            if (m == M3D_NN_NONE_CACHED) cache[qi] = code;
        }
    }
}

// k_nn_tiles: the searches k_nn_iter binned, answered from LDS. One workgroup per (pair, tile): it copies the tile's image into LDS —
// the voxel directory and the tile's own + neighbouring buckets' points (one coalesced index stream, one 16-B gather per staged
// point, all independent) — and answers every query record of the tile against it, one record per thread; a tile with several
// images (a crowded stretch) stages them one after the other, the queries' best-so-far riding in registers. Results go where
// k_nn_iter puts its own (match / cache / state).
// Workgroups are dealt over the XCDs tile by tile, NOT pair by pair like the other kernels of the iteration: nothing here is read
// twice (records, images and points stream through once), and a pair with crowded tiles then loads all XCDs instead of one.
#define M3D_TILE_THREADS 512
#define M3D_TILE_GRID 2048     // workgroups of k_nn_tiles: 256 per list, they stride over the work items k_nn_iter published there (an empty list costs one word read each)
static_assert(M3D_TILE_GRID % M3D_TILE_LISTS == 0, "every list is served by the same number of workgroups");
__global__ __launch_bounds__(M3D_TILE_THREADS, 6) void k_nn_tiles(const M3dJob* __restrict__ jobs, int first_of_level, M3dNnArgs A) {
    __shared__ m3d_f32x4 s_pts[M3D_TILE_PCAP];
    __shared__ m3d_u32x2 s_vs[M3D_TILE_VS];
    __shared__ int s_delta[M3D_TILE_ECAP];   // sorted position - LDS position of the points of every staged bucket of the tile (one table for all of its images)
    __shared__ int s_kd[32];
    M3D_ENTRY_JITTER();
    const unsigned int wl = blockIdx.x & (unsigned int)(M3D_TILE_LISTS - 1);   // this workgroup's list of work items
    const unsigned int n_items = min(A.wcount[32u * wl], (unsigned int)A.wcap);   // (uniform; the reduction pass zeroes the counters)
    const uint2* witems = A.witems + (size_t)wl * (size_t)A.wcap;
    const int tid0 = (int)threadIdx.x;
    const m3d_lu2 vs = (m3d_lu2)s_vs;
    const m3d_lf4 sp = (m3d_lf4)s_pts;
    bool first_item = true;
    for (unsigned int it = blockIdx.x / (unsigned int)M3D_TILE_LISTS; it < n_items; it += gridDim.x / (unsigned int)M3D_TILE_LISTS) {
        // (the thread index is re-made per item behind an opaque barrier for the optimiser: with a loop-invariant tid the compiler hoists a dozen staging
        // addresses out of the item loop and keeps them in VGPRs across the search — at 80 VGPRs that meant 32 bytes of scratch per lane, spilled and
        // reloaded per item: 35 MB of WRITE traffic per launch, more than the results themselves)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const uint2 item = witems[it];
        const int pair = (int)M3D_CHK(109, item.x, 65536u), blk = (int)M3D_CHK(110, item.y & 0xFFFFFu, A.ntile_max);
        const unsigned int chunk = item.y >> 20;
        const M3dJob& J = jobs[pair];
        const M3dPairState* st = A.states + pair;
        const int itq = st->iters & 31;
        M3D_TBT_BEGIN();
        const M3dGrid g = J.tgt.g;
        const float dmax2 = J.dmax2;
        // (workgroup-uniform words, pinned to scalar registers: the compiler kept some of them in VGPRs — and then in scratch — across the images of an item)
        M3dTileHdr H = J.tgt.thdr[blk];
        H.extra = (uint32_t)__builtin_amdgcn_readfirstlane((int)H.extra); H.n_img = (uint32_t)__builtin_amdgcn_readfirstlane((int)H.n_img);
        H.flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)H.flags); H.meta0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)H.meta0);
        const unsigned int qn = (unsigned int)__builtin_amdgcn_readfirstlane((int)min((A.tcnt + (size_t)pair * A.cnt_stride)[blk], (unsigned int)M3D_TILE_QCAP));
        // (a tile with crowded voxels: 256 / 128 / 64 records per item, two / four / eight lanes per record: m3d_tile_search's group mode)
        const unsigned int cs = (unsigned int)__builtin_amdgcn_readfirstlane((int)m3d_tile_records_per_item(H.meta0, (uint32_t)A.tile_chunk));
        const unsigned int lstride = M3D_TILE_THREADS / cs;
        M3D_GLOBAL m3d_i32x2* out = (M3D_GLOBAL m3d_i32x2*)(void M3D_GLOBAL*)(A.match + (size_t)pair * A.match_stride);
        M3D_GLOBAL long long* cache = (M3D_GLOBAL long long*)(void M3D_GLOBAL*)(A.cache + (size_t)pair * A.match_stride);
        if (!first_item) __syncthreads();   // everybody is done with the previous item's LDS
        if (tid < 27) {   // key offset of voxel b of the visiting order (m3d_tile_search): dx + dy * 2^sh1 + dz * 2^sh2
            // the visiting order (home, faces, edges, corners; index = (dx+1) + 3 (dy+1) + 9 (dz+1)), five bits each, in three constants (a table in constant
            // memory cost a 64-bit address held across the item loop)
            const unsigned long long o0 = 13ull | 12ull << 5 | 14ull << 10 | 10ull << 15 | 16ull << 20 | 4ull << 25 | 22ull << 30 | 9ull << 35 | 11ull << 40 | 15ull << 45 | 17ull << 50 | 3ull << 55;
            const unsigned long long o1 = 5ull | 21ull << 5 | 23ull << 10 | 1ull << 15 | 7ull << 20 | 19ull << 25 | 25ull << 30 | 0ull << 35 | 2ull << 40 | 6ull << 45 | 8ull << 50 | 18ull << 55;
            const unsigned long long o2 = 20ull | 24ull << 5 | 26ull << 10;
            const int v = (int)(((tid < 12 ? o0 : (tid < 24 ? o1 : o2)) >> (5 * (tid % 12))) & 31ull), dx = v % 3 - 1, dy = (v / 3) % 3 - 1, dz = v / 9 - 1;
            s_kd[tid] = dx + dy * (1 << (g.cb[0] + 1)) + dz * (1 << (g.cb[0] + g.cb[1] + 2));
        }
        const m3d_gf4 rec = m3d_as_global(A.rec + (size_t)pair * A.rec_stride + (size_t)blk * M3D_TILE_QCAP);
        const float* recd = A.recd + (size_t)pair * A.rec_stride + (size_t)blk * M3D_TILE_QCAP;
        unsigned int n_staged = 0;
        const unsigned int q = chunk * cs + (unsigned int)tid / lstride;
        const bool have = q < qn;
        const uint32_t sub = (uint32_t)tid % lstride;   // (lstride lanes share a record: a whole number of groups per wave, idle groups only at the end of the item)
        // the record first: its loads are in flight while the image is staged
#ifdef M3D_EXP_UNIFORM   // (timing experiment only — WRONG results: every lane of a wave searches lane 0's query: what would the search cost without divergence?)
        float4 r4 = have ? m3d_ld(rec, (size_t)q) : make_float4(0.f, 0.f, 0.f, 0.f);
        r4.x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r4.x))); r4.y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r4.y)));
        r4.z = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r4.z)));
        asm volatile("" : "+v"(r4.x), "+v"(r4.y), "+v"(r4.z));
        float dseed = have ? recd[q] : 0.f;
        dseed = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(dseed)));
        asm volatile("" : "+v"(dseed));
#else
        const float4 r4 = have ? m3d_ld(rec, (size_t)q) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float dseed = have ? recd[q] : 0.f;
#endif
        const uint32_t w = __float_as_uint(r4.w);
        const bool seeded = (w >> 31) != 0u;
        M3dTileQ Q;
        int m = -1;
        for (unsigned int j = 0; j < M3D_CHK_LE(116, H.n_img, M3D_TILE_MAXIMG); j++) {
            const unsigned int image = M3D_CHK(111, j == 0u ? (unsigned int)blk : H.extra + j - 1u, A.ntile_max + m3d_tile_pool(A.ntile_max));
            const uint8_t* img = J.tgt.timg + (size_t)image * M3D_TILE_IMG_BYTES;
            {   // stage: the voxel list is hashed into the LDS directory here (the image stores the list, not 16 KB of mostly empty slots)
                if (j != 0u) __syncthreads();   // everybody is done with the previous image
                const M3dTileImgMeta IM = J.tgt.timeta[image];
                const unsigned int n_points = M3D_CHK_LE(112, IM.n_points, M3D_TILE_PCAP), n_vox = M3D_CHK_LE(113, IM.n_voxels & 0x7FFFFFFFu, M3D_TILE_VCAP);
                const M3D_GLOBAL m3d_u32x2* gl = (const M3D_GLOBAL m3d_u32x2*)(const void M3D_GLOBAL*)img;
                const m3d_gf4 gp = m3d_as_global(reinterpret_cast<const float4*>(img + M3D_TILE_IMG_PTS));
                static_assert(M3D_TILE_VCAP <= 3 * M3D_TILE_THREADS && M3D_TILE_PCAP == 4 * M3D_TILE_THREADS && M3D_TILE_VS == 4 * M3D_TILE_THREADS && M3D_TILE_ECAP <= M3D_TILE_THREADS,
                              "staging: at most three list entries, four points, four directory slots and one bucket delta per thread");
                m3d_u32x2 vl[3];
#pragma unroll
                for (int r = 0; r < 3; r++) { const unsigned int k = (unsigned int)(M3D_TILE_THREADS * r + tid); vl[r] = k < n_vox ? gl[k] : (m3d_u32x2){ M3D_INVALID_KEY, 0u }; }
                float4 pv[4];
#pragma unroll
                for (int r = 0; r < 4; r++) { const unsigned int k = (unsigned int)(M3D_TILE_THREADS * r + tid); pv[r] = k < n_points ? m3d_ld(gp, (size_t)k) : make_float4(0.f, 0.f, 0.f, 0.f); }
                int dl = 0;
                if (j == 0u && (unsigned int)tid < (H.flags >> 16)) dl = reinterpret_cast<const int*>(J.tgt.timg + (size_t)blk * M3D_TILE_IMG_BYTES + M3D_TILE_IMG_DELTA)[tid];
                {   // empty directory (while the loads above are in flight)
                    typedef uint32_t u32x4_lds __attribute__((ext_vector_type(4)));
                    __attribute__((address_space(3))) u32x4_lds* d4 = (__attribute__((address_space(3))) u32x4_lds*)s_vs;
                    const u32x4_lds e4 = { M3D_INVALID_KEY, 0u, M3D_INVALID_KEY, 0u };
                    d4[2 * tid] = e4; d4[2 * tid + 1] = e4;
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    if (vl[r].x == M3D_INVALID_KEY) continue;
                    uint32_t h = (vl[r].x * 0x9E3779B1u) >> (32 - 11);
                    // voxel keys are unique: a successful CAS owns the slot (at most VCAP of the VS slots are ever taken)
                    while (atomicCAS(reinterpret_cast<uint32_t*>(&s_vs[h]), M3D_INVALID_KEY, vl[r].x) != M3D_INVALID_KEY) h = (h + 1u) & (M3D_TILE_VS - 1u);
                    reinterpret_cast<uint32_t*>(&s_vs[h])[1] = vl[r].y;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) { const unsigned int k = (unsigned int)(M3D_TILE_THREADS * r + tid); if (k < n_points) s_pts[k] = (m3d_f32x4){ pv[r].x, pv[r].y, pv[r].z, pv[r].w }; }
                if (j == 0u && (unsigned int)tid < (H.flags >> 16)) s_delta[tid] = dl;
                __syncthreads();
                n_staged += n_points;
                M3D_TBT_STAGED();
            }
            if (have) {
                // (the query's geometry is derived HERE, behind the staging: hoisted above it — it only needs the record — its nine gap words lived across the
                // staging's sixteen point registers and went to scratch)
                float qx = r4.x, qy = r4.y, qz = r4.z;
                asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));
                if (j == 0u) m3d_tile_query(g, qx, qy, qz, dmax2, seeded, dseed, Q);
#ifdef M3D_EXP_NOSEARCH   // (timing experiment only — WRONG results: the floor of an item — records, staging, result writes — without the search)
                const int b = -1;
#else
                const int b = m3d_tile_search(g, vs, sp, s_kd, s_delta, r4.x, r4.y, r4.z, Q, sub, lstride);
#endif
                if (b >= 0) m = b;   // (a sorted position already)
            }
            M3D_TBT_SEARCHED();
        }
        if (have && sub == 0u) {
            // (the record is read AGAIN here — an L2 hit — instead of riding through the search in registers: query index, seeded flag and the voxel code
            // of a "nothing around" verdict were the last values the kernel kept in scratch across the images of an item)
            const float4 re = m3d_ld(rec, (size_t)q);
            const uint32_t we = __float_as_uint(re.w);
            const int qi = (int)M3D_CHK(114, we & 0x7FFFFFFFu, A.match_stride);
            if (m >= 0) m = (int)M3D_CHK(115, m, g.n_valid);
            const float d2 = m3d_key_d2(Q.bkey);
            // "nothing at all in the 27 voxels" may be cached only when every existing voxel was looked up and found empty
            if (!(m >= 0 && d2 <= dmax2)) m = ((we >> 31) != 0u || m >= 0 || Q.sec != M3D_INF_BITS) ? -1 : M3D_NN_NONE_CACHED;
            out[qi] = (m3d_i32x2){ m, m3d_cert_pack_bits(Q.sec, itq) };   // ONE 8-byte store per answer (rounds 2-3: a 4-byte match and a 16-byte state, two sectors)
            if (m == M3D_NN_NONE_CACHED) cache[qi] = m3d_tile_code(g, re.x, re.y, re.z);
        }
        const unsigned int n_done = min(cs, qn - min(qn, chunk * cs));
        if (tid == 0) atomicAdd(&A.states[pair].ctr[0], n_done);
        (void)n_staged;
        M3D_TBT_END(0, n_done, n_staged);
        first_item = false;
    }
}

// point-to-point: expand the 17 transported sums into the spec's 29 slots (exact integer identities:
// quant(-x) == -quant(x), and the translation block is count * 2^30 on the diagonal).
__device__ __forceinline__ void expand_pt2pt(const long long in[17], long long out[M3D_NSUMS]) {
    for (int i = 0; i < M3D_NSUMS; i++) out[i] = 0;
    out[hslot21(0, 0)] = in[0]; out[hslot21(0, 1)] = in[1]; out[hslot21(0, 2)] = in[2];
    out[hslot21(1, 1)] = in[3]; out[hslot21(1, 2)] = in[4]; out[hslot21(2, 2)] = in[5];
    const long long swx = in[6], swy = in[7], swz = in[8];
    out[hslot21(0, 4)] = -swz; out[hslot21(0, 5)] = swy;
    out[hslot21(1, 3)] = swz;  out[hslot21(1, 5)] = -swx;
    out[hslot21(2, 3)] = -swy; out[hslot21(2, 4)] = swx;
    const long long cnt = in[16];
    out[hslot21(3, 3)] = cnt << 30; out[hslot21(4, 4)] = cnt << 30; out[hslot21(5, 5)] = cnt << 30;
    for (int k = 0; k < 6; k++) out[21 + k] = in[9 + k];
    out[27] = in[15];
    out[28] = cnt;
}

// ---- a8: 6x6 LDL^T solve + SE(3) update about the centre (one thread per pair, all double) -------
// returns -1 to keep iterating, else an m3dreg_status
__device__ int m3d_solve_update(const long long sums[M3D_NSUMS], const int exps[6], const float center[3], double pivot_rel_tol,
                                double T[16], double& th2_out, double& tr2_out) {
    double A[6][6], b[6];
    for (int k = 0; k < 6; k++)
        for (int l = k; l < 6; l++) {
            const int cls = (l < 3) ? 0 : (k < 3 ? 1 : 2);
            const double v = ldexp((double)sums[hslot21(k, l)], -exps[cls]);
            A[k][l] = v; A[l][k] = v;
        }
    for (int k = 0; k < 6; k++) b[k] = -ldexp((double)sums[21 + k], -exps[k < 3 ? 3 : 4]);
    double dmax = 0.0;
    for (int k = 0; k < 6; k++) if (A[k][k] > dmax) dmax = A[k][k];
    const double tol = pivot_rel_tol * dmax;
    double Lm[6][6], D[6], Dinv[6];   // spec v2: ONE division per pivot (its reciprocal), every other quotient is a product with it
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
        for (int k = 0; k < j; k++) d = d - Lm[j][k] * Lm[j][k] * D[k];
        if (!(d > tol)) return 3;
        D[j] = d;
        const double inv_d = 1.0 / d;
        Dinv[j] = inv_d;
        for (int i = j + 1; i < 6; i++) {
            double v = A[i][j];
            for (int k = 0; k < j; k++) v = v - Lm[i][k] * Lm[j][k] * D[k];
            Lm[i][j] = v * inv_d;
        }
    }
    double y[6], x[6];
    for (int i = 0; i < 6; i++) { double v = b[i]; for (int k = 0; k < i; k++) v = v - Lm[i][k] * y[k]; y[i] = v; }
    for (int i = 0; i < 6; i++) y[i] = y[i] * Dinv[i];
    for (int i = 5; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 6; k++) v = v - Lm[k][i] * x[k]; x[i] = v; }
    const double w0 = x[0], w1 = x[1], w2 = x[2], v0 = x[3], v1 = x[4], v2 = x[5];
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    const double tr2 = v0 * v0 + v1 * v1 + v2 * v2;
    th2_out = th2; tr2_out = tr2;
    if (!(th2 <= 4.0) || !(tr2 < 1e300)) return 4;
    double sa = 1.0, sb = 1.0, sc = 1.0;
    // spec v2: the constant divisors are multiplied in as their (correctly rounded) reciprocals — 59 double divisions per solve were
    // 4 of the 5 us this step takes on one lane. (Unrolled: the reciprocals fold to constants.)
#pragma unroll
    for (int k = 12; k >= 1; k--) {
        sa = 1.0 - th2 * sa * (1.0 / (double)((2 * k) * (2 * k + 1)));
        sb = 1.0 - th2 * sb * (1.0 / (double)((2 * k + 1) * (2 * k + 2)));
        sc = 1.0 - th2 * sc * (1.0 / (double)((2 * k + 2) * (2 * k + 3)));
    }
    const double Ac = sa, Bc = sb * 0.5, Cc = sc * (1.0 / 6.0);
    const double W[9] = { 0, -w2, w1, w2, 0, -w0, -w1, w0, 0 };
    const double W2[9] = { -(w1 * w1 + w2 * w2), w0 * w1, w0 * w2, w0 * w1, -(w0 * w0 + w2 * w2), w1 * w2, w0 * w2, w1 * w2, -(w0 * w0 + w1 * w1) };
    double Re[9], Ve[9];
    for (int i = 0; i < 9; i++) {
        const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        Re[i] = id + Ac * W[i] + Bc * W2[i];
        Ve[i] = id + Bc * W[i] + Cc * W2[i];
    }
    const double te[3] = { Ve[0] * v0 + Ve[1] * v1 + Ve[2] * v2, Ve[3] * v0 + Ve[4] * v1 + Ve[5] * v2, Ve[6] * v0 + Ve[7] * v1 + Ve[8] * v2 };
    const double c[3] = { (double)center[0], (double)center[1], (double)center[2] };
    double Rn[9], tn[3];
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++)
            Rn[3 * r + cc] = Re[3 * r] * T[cc * 4 + 0] + Re[3 * r + 1] * T[cc * 4 + 1] + Re[3 * r + 2] * T[cc * 4 + 2];
    const double d0 = T[12] - c[0], d1 = T[13] - c[1], d2 = T[14] - c[2];
    for (int r = 0; r < 3; r++) tn[r] = (Re[3 * r] * d0 + Re[3 * r + 1] * d1 + Re[3 * r + 2] * d2) + c[r] + te[r];
    for (int r = 0; r < 3; r++) { for (int cc = 0; cc < 3; cc++) T[cc * 4 + r] = Rn[3 * r + cc]; T[12 + r] = tn[r]; }
    T[3] = 0.0; T[7] = 0.0; T[11] = 0.0; T[15] = 1.0;
    return -1;
}

// One thread per pair: consume the sums of the iteration that just ran, update the pose, decide.
// active_known: the caller's workgroups left at once had the pair been finished (the reduction kernels) — the state's flags are not loaded again: the solve's other loads
// then do not wait behind that round trip
__device__ __forceinline__ void m3d_solve_pair(const M3dJob& J, int first_of_level, const long long* raw = nullptr, const double* T_pre = nullptr, bool active_known = false) {
    M3dPairState* st = J.st;
    if (!active_known && (st->done || (!first_of_level && st->level_done))) return;
    long long sums[M3D_NSUMS];
    const bool own = raw == nullptr;   // sums accumulated by atomics in the state (fused variants) or handed in (block partials)
    if (own) raw = st->sums;
    if (J.metric == 1) { for (int i = 0; i < M3D_NSUMS; i++) sums[i] = raw[i]; }
    else { long long in[17]; for (int i = 0; i < 17; i++) in[i] = raw[i]; expand_pt2pt(in, sums); }
    if (own) for (int i = 0; i < M3D_NSUMS; i++) st->sums[i] = 0;
    int exps[6];
    for (int i = 0; i < 6; i++) exps[i] = J.exps[i];
    double T[16];
    for (int i = 0; i < 16; i++) T[i] = T_pre ? T_pre[i] : st->T[i];   // (T_pre: the pose, fetched while the block was streaming — LDS)
    const int it = st->iters;
    st->iters = it + 1;
    st->n_corr = sums[28];
    st->ssr = sums[27];
    st->ssr_exp = exps[5];
    int done = 0, level_done = 0;
    if (sums[28] < (long long)J.min_corr) { st->status = 2; done = 1; }
    else {
        double th2 = 0.0, tr2 = 0.0;
        const int rc = m3d_solve_update(sums, exps, J.tgt.g.center, J.pivot_rel_tol, T, th2, tr2);
        st->th2 = th2; st->tr2 = tr2;
        if (rc >= 0) { st->status = rc; done = 1; }
        else {
            for (int i = 0; i < 16; i++) st->T[i] = T[i];
            if (J.ring) {   // the float pose the NEXT iteration will use (m3d_load_pose's rounding), for the certificates of the iterations after it
                float* r = J.ring + 12 * ((it + 1) & 31);
                for (int rr = 0; rr < 3; rr++) { for (int c = 0; c < 3; c++) r[3 * rr + c] = (float)T[c * 4 + rr]; r[9 + rr] = (float)T[12 + rr]; }
            }
            if (th2 < J.eps_rot2 && tr2 < J.eps_trans2) {
                if (J.last_level) { st->status = 0; done = 1; }
                else level_done = 1;
            }
        }
    }
    if (J.trace && it < M3D_MAX_TRACE) for (int i = 0; i < 16; i++) J.trace[16 * it + i] = T[i];
    st->done = done;
    st->level_done = level_done;
}

// Batch-wide arrival: the last pair to report publishes {iteration sequence number, pairs that still have work at this
// level} as ONE 8-byte store into host-mapped memory: the host polls it between launches and stops enqueuing a level's
// remaining iterations once nothing is active — early termination without any host-device synchronisation (a stale read
// only costs a few empty launches, never correctness). The two counters live in pair 0's state, agent-scope atomics only.
__device__ __forceinline__ void m3d_report_progress(const M3dJob* __restrict__ jobs, int n_pairs, bool still_active, unsigned int seq,
                                                    unsigned long long* __restrict__ progress) {
    if (!progress) return;   // nobody listens (fixed iteration counts): skip the device-scope atomics
    M3dPairState* g0 = jobs[0].st;
    // ONE read-modify-write per pair carries both the arrival (low half) and "still active" (high half): the last arriver reads
    // the number of active pairs out of the value its own add returned — no second counter whose increment could land late
    // (n_pairs <= 65535).
    const unsigned int add = still_active ? 0x10001u : 1u;
    const unsigned int t = __hip_atomic_fetch_add(&g0->gsync[0], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
    if ((t & 0xFFFFu) != (unsigned int)n_pairs) return;
    const unsigned int active = t >> 16;
    __hip_atomic_store(&g0->gsync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (progress) {
        __threadfence_system();
        *reinterpret_cast<volatile unsigned long long*>(progress) = ((unsigned long long)seq << 32) | active;
    }
}

__host__ __device__ inline int m3d_ticket_group(int bpp);
#ifdef M3D_LATE_STAMPS   // (diagnosis build) the tail of pair 0's LAST workgroup, phase by phase: entry, stores drained, tickets taken, partials added, solved, reported
__device__ unsigned long long g_tail_stamp[8];
extern "C" hipError_t m3d_debug_read_tail(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_stamp), sizeof(unsigned long long) * 8); }
#define TAIL_STAMP(k) do { if (threadIdx.x == 0 && pair == 0) g_tail_stamp[k] = wall_clock64(); } while (0)
#else
#define TAIL_STAMP(k) ((void)0)
#endif
#ifndef M3D_TAIL_LOADS
#define M3D_TAIL_LOADS 32
#endif
// The end of an iteration for one pair, reached by every thread of every block of the pair's reduction: the LAST block
// to arrive adds up the pair's block partials, solves the 6x6 system and updates the pose (a8).
__device__ __forceinline__ void m3d_pair_tail(const M3dJob* __restrict__ jobs, const M3dJob& J, M3dPairState* st, int n_pairs, int pair, int blk, int bpp,
                                              int first_of_level, const long long* __restrict__ partials, unsigned int* __restrict__ tickets,
                                              unsigned int seq, unsigned long long* __restrict__ progress, const double* T_pre = nullptr, unsigned int* zero_word = nullptr) {
    // ---- a8 in the same launch: the LAST block of the pair to finish adds up the pair's block partials and solves.
    // (A separate solve kernel cost its ~10 us plus a dependent-launch gap of ~5 us in every iteration.)
    // The partials are stored and loaded with agent-scope (write-through / coherent) accesses and the writers wait for
    // their stores before the block takes its ticket: no __threadfence(), whose agent-scope release writes back the whole
    // L2 — including the megabytes of match/state lines the search kernel left dirty (measured: +50 us per iteration).
    // (The blocks of a pair sit on one XCD only when the batch is a multiple of 8 pairs; the L2s of different XCDs are
    // not coherent for plain accesses.)
    // "Last block": a device-scope returning atomic is performed at the memory side and same-address ones retire one after
    // the other at ~5 per microsecond — 49 arrivals on one ticket were the 10 us this tail cost. Two levels instead: a block
    // arrives at its GROUP's counter (~sqrt(bpp) blocks each, every counter on its own 128-B line), the last of a group at
    // the pair's counter: at most ~7 + 7 serialised arrivals instead of 49.
    __shared__ int s_last;
    __shared__ long long s_part[8][M3D_PARTIAL_STRIDE];
#ifdef M3D_LATE_STAMPS
    const unsigned long long ts0 = wall_clock64();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef M3D_LATE_STAMPS
    const unsigned long long ts1 = wall_clock64();
#endif
    if (threadIdx.x == 0) {
        const int gs = m3d_ticket_group(bpp), ng = (bpp + gs - 1) / gs, grp = blk / gs;
        unsigned int* tk = tickets + (size_t)pair * (size_t)(ng + 1) * 32u;
        const unsigned int members = (unsigned int)min(gs, bpp - grp * gs);
        int last = 0;
        if (__hip_atomic_fetch_add(&tk[32 * (1 + grp)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
            __hip_atomic_store(&tk[32 * (1 + grp)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&tk[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)ng - 1u) {
                __hip_atomic_store(&tk[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
#ifdef M3D_LATE_STAMPS
    if (threadIdx.x == 0 && pair == 0) { g_tail_stamp[0] = ts0; g_tail_stamp[1] = ts1; }
#endif
    TAIL_STAMP(2);
    if (zero_word && threadIdx.x == 0) *zero_word = 0u;   // (every workgroup of the pair has read it: they all arrived)
    {
        const int slot = threadIdx.x & 31, seg = threadIdx.x >> 5;
        long long v = 0;
        if (slot < M3D_NSUMS)
            for (int b0 = seg; b0 < bpp; b0 += 8 * M3D_TAIL_LOADS) {   // M3D_TAIL_LOADS loads in flight per thread (a `v += load` loop waits for every one). Round 5: 32, was 8 —
                // a LONE pair is reduced by 256 workgroups (m3d_acc_blocks), its last workgroup made five dependent round trips here: 3 us of every launch of configs 2, 3 and 5
                long long t[M3D_TAIL_LOADS];
#pragma unroll
                for (int k = 0; k < M3D_TAIL_LOADS; k++) {
                    const int b = b0 + 8 * k;
                    t[k] = (b < bpp) ? __hip_atomic_load(&partials[((size_t)pair * bpp + b) * M3D_PARTIAL_STRIDE + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ll;
                }
#pragma unroll
                for (int k = 0; k < M3D_TAIL_LOADS; k++) v += t[k];
            }
        s_part[seg][slot] = v;
        __syncthreads();
        if (threadIdx.x < M3D_NSUMS) {
            long long t = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) t += s_part[k][threadIdx.x];
            s_part[0][threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    TAIL_STAMP(3);
    m3d_solve_pair(J, first_of_level, s_part[0], T_pre, true);   // (every caller of this tail returned at its first line for a finished pair)
    TAIL_STAMP(4);
    m3d_report_progress(jobs, n_pairs, !st->done && !st->level_done, seq, progress);
    TAIL_STAMP(5);
}

// WALK (the tile iterations of a lean registration): a query k_nn_iter<lean> left M3D_NN_PENDING is walked HERE, by the workgroup that streams it
// (so nobody else ever reads its match in this launch), its residual then added by the streaming loop like everybody's. Unseeded: a seed only
// bounds the walk, the match is the same. Normally there are none (one counter read per workgroup); a few are walked 8 lanes per query like
// k_icp_late's worklist; a workgroup that finds MANY (a target whose buckets hold more points than a tile image: every query of the pair is
// pending) walks them one query per lane, all 256 lanes busy. (Rounds 2-3 had a launch of their own for this, k_nn_fallback — 5.3 us in every tile
// iteration for a list that is empty on ordinary clouds — and, in round 3, a per-handle memory of whether the last batch had needed it: both gone.)
template <int METRIC, bool WALK>
__global__ __launch_bounds__(ICP_THREADS) void k_accumulate_matches(   // (WALK: 133 VGPRs = 3 waves per SIMD, the walk's; forced to 128 = 4 waves with 5 spilled registers: headline and serial steps -0.5 %)
                                                                    const M3dJob* __restrict__ jobs, int n_pairs, int bpp, int first_of_level,
                                                                    long long* __restrict__ partials, unsigned int* __restrict__ tickets,
                                                                    M3dPairState* __restrict__ states, unsigned int seq, unsigned long long* __restrict__ progress, int fuse_solve, int rot,
                                                                    unsigned int* __restrict__ gw_cnt, int gw_stride, int gw_n, unsigned int* __restrict__ wcount, M3dNnArgs A) {
    int pair, blk;
    m3d_map_block(n_pairs, bpp, pair, blk, rot);
    if (gw_cnt && blk == 0) {   // the record counters of k_nn_tiles: each is read by several of its workgroups, so they are zeroed one launch later
        for (int i = (int)threadIdx.x; i < gw_n; i += ICP_THREADS) gw_cnt[(size_t)pair * gw_stride + i] = 0u;
        if (pair == 0 && threadIdx.x < M3D_TILE_LISTS) wcount[32u * threadIdx.x] = 0u;
    }
    const M3dJob& J = jobs[pair];
    M3dPairState* st = states ? states + pair : J.st;   // == J.st, addressed from the kernel argument when the caller has it

    if (st->done || (!first_of_level && st->level_done)) {
        // a finished pair still reports to the batch-wide progress word (one thread per pair)
        if (fuse_solve && blk == 0 && threadIdx.x == 0) m3d_report_progress(jobs, n_pairs, false, seq, progress);
        return;
    }
    float R[9], tt[3];
    m3d_load_pose(st, R, tt);
    // the pose in double for the solve at the end of the launch: requested now, parked in LDS after the streaming loop (the last
    // block's solving thread otherwise starts with a 1 us round trip)
    __shared__ double s_T[16];
    const M3dLevelDev& L = J.tgt;
    const float cx = L.g.center[0], cy = L.g.center[1], cz = L.g.center[2];
    const float S[6] = { J.S[0], J.S[1], J.S[2], J.S[3], J.S[4], J.S[5] };
    constexpr int NACC = (METRIC == 1) ? 29 : 17;
    const int n = J.n_src;
    const int2* in = A.match + (size_t)pair * A.match_stride;
    const m3d_gf3 src = m3d_as_global3(J.src);
    const m3d_gf4 pts = m3d_as_global(L.pts), nrm = m3d_as_global(L.nrm);
    // NB queries per trip, every load of a stage issued before the first use: the pass is a chain of
    // dependent gathers (match -> point, normal), so its speed is the number of them in flight
    constexpr int NB = 2;   // (4 in flight: 180 VGPRs = 2 waves per SIMD; 2: 154 = 3 waves, same duration alone, +1.7 % with three chains sharing the GPU;
                            //  the next trip's (match, source point) loads issued ahead of this trip's gathers: 170 VGPRs, same duration, -1 %)
    const int stride = bpp * ICP_THREADS;
    unsigned int* pend_n = nullptr;   // WALK: the pair's count of pending queries (the counter behind its tiles'); the pair's LAST workgroup zeroes it (m3d_pair_tail)
    if constexpr (WALK) {
        pend_n = A.tcnt + (size_t)pair * A.cnt_stride + A.ntile_max;
        if (*pend_n != 0u) {   // (uniform; nearly never taken — and BEFORE the streaming loop: the 29 sums are not live yet, the walk's registers are the loop's)
            __shared__ int s_pn;
            __shared__ int s_pend[M3D_LATE_CAP];   // pending queries of this workgroup (it owns at most M3D_LATE_QPT x 256: launch_iteration checks)
            if (threadIdx.x == 0) s_pn = 0;
            __syncthreads();
            for (int i = blk * ICP_THREADS + (int)threadIdx.x; i < n; i += stride)
                if (in[i].x == M3D_NN_PENDING) { const int w = atomicAdd(&s_pn, 1); if (w < M3D_LATE_CAP) s_pend[w] = i; }
            __syncthreads();
            const int nW = min(s_pn, M3D_LATE_CAP);
            const M3dGrid g = L.g;
            const m3d_gu4 tab = m3d_as_global(reinterpret_cast<const uint4*>(L.htab));
            const m3d_gf4 cbox = m3d_as_global(L.cbox);
            const m3d_gu32 bigcum = m3d_as_global(L.bigcum);
            M3D_GLOBAL m3d_i32x2* out = (M3D_GLOBAL m3d_i32x2*)(void M3D_GLOBAL*)(A.match + (size_t)pair * A.match_stride);
            M3D_GLOBAL long long* cache = (M3D_GLOBAL long long*)(void M3D_GLOBAL*)(A.cache + (size_t)pair * A.match_stride);
            const int itq = st->iters & 31;
            if (nW > 2 * ICP_THREADS) {   // (uniform) many: one query per lane
                for (int w = (int)threadIdx.x; w < nW; w += ICP_THREADS) {
                    const int qi = M3D_CHK(119, s_pend[w], n);
                    const float4 ps = m3d_ld3(src, (size_t)qi);
                    const float vx = fmaf(R[0], ps.x, fmaf(R[1], ps.y, fmaf(R[2], ps.z, tt[0])));
                    const float vy = fmaf(R[3], ps.x, fmaf(R[4], ps.y, fmaf(R[5], ps.z, tt[1])));
                    const float vz = fmaf(R[6], ps.x, fmaf(R[7], ps.y, fmaf(R[8], ps.z, tt[2])));
                    long long code = 0; float sec = 0.f;
                    const int mq = m3d_nn27_walk(g, tab, pts, cbox, bigcum, vx, vy, vz, J.dmax2, false, 0.f, code, sec, 0);
                    out[qi] = (m3d_i32x2){ mq, m3d_cert_pack(sec, itq) };
                    if (mq == M3D_NN_NONE_CACHED) cache[qi] = code;
                }
            } else {
                const int sub = (int)threadIdx.x & 7;
                for (int base = 0; base < nW; base += ICP_THREADS / 8) {   // uniform trip count: the shuffles inside need every lane
                    const int w = base + ((int)threadIdx.x >> 3);
                    const bool act = w < nW;
                    const int qi = act ? s_pend[w] : 0;
                    const float4 ps = m3d_ld3(src, (size_t)qi);
                    const float vx = fmaf(R[0], ps.x, fmaf(R[1], ps.y, fmaf(R[2], ps.z, tt[0])));
                    const float vy = fmaf(R[3], ps.x, fmaf(R[4], ps.y, fmaf(R[5], ps.z, tt[1])));
                    const float vz = fmaf(R[6], ps.x, fmaf(R[7], ps.y, fmaf(R[8], ps.z, tt[2])));
                    long long code; float sec;
                    const int mq = m3d_coop_query(g, tab, pts, cbox, bigcum, J.dmax2, act, false, vx, vy, vz, 0.f, sub, code, sec, 0);
                    if (act && sub == 0) {
                        out[qi] = (m3d_i32x2){ mq, m3d_cert_pack(sec, itq) };   // (read back by this workgroup only, below, behind the barrier)
                        if (mq == M3D_NN_NONE_CACHED) cache[qi] = code;
                    }
                }
            }
            if (threadIdx.x == 0 && nW > 0) atomicAdd(&st->ctr[1], (unsigned int)nW);
            __threadfence_block();
            __syncthreads();
        }
    }
    double t_pre = 0.0;   // (requested here, behind the walk: two registers it does not have to carry)
    if (threadIdx.x < 16) t_pre = st->T[threadIdx.x];
    M3D_ACC<NACC> acc;
    acc.clear();
    for (int i0 = blk * ICP_THREADS + (int)threadIdx.x; i0 < n; i0 += NB * stride) {
        int m[NB]; float4 p[NB], q[NB], nq[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) { const int i = i0 + k * stride; m[k] = (i < n) ? in[i].x : -1; }
#pragma unroll
        for (int k = 0; k < NB; k++) { const int i = i0 + k * stride; p[k] = (i < n) ? m3d_ld3(src, (size_t)i) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const size_t mm = (size_t)M3D_CHK(117, max(m[k], 0), max(L.g.n_valid, 1));
            q[k] = m3d_ld(pts, mm);
            nq[k] = (METRIC == 1) ? m3d_ld(nrm, mm) : make_float4(0.f, 0.f, 0.f, 0.f);   // sorted order: neighbouring matches share cache lines
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (m[k] < 0) continue;
            const float ux = fmaf(R[0], p[k].x, fmaf(R[1], p[k].y, fmaf(R[2], p[k].z, tt[0])));
            const float uy = fmaf(R[3], p[k].x, fmaf(R[4], p[k].y, fmaf(R[5], p[k].z, tt[1])));
            const float uz = fmaf(R[6], p[k].x, fmaf(R[7], p[k].y, fmaf(R[8], p[k].z, tt[2])));
            const float ex = ux - q[k].x, ey = uy - q[k].y, ez = uz - q[k].z;
            const float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));   // same chain as the search: same bits
            m3d_accumulate_match<METRIC, NACC>(acc, ux, uy, uz, q[k], d2, nq[k], cx, cy, cz, S);
        }
    }
    if (threadIdx.x < 16) s_T[threadIdx.x] = t_pre;   // (the reduction below has the barriers that publish it)
    block_reduce_to_global<NACC>(acc, st->sums, partials ? partials + ((size_t)pair * bpp + blk) * M3D_PARTIAL_STRIDE : nullptr);
    if (!fuse_solve || !partials) return;
    m3d_pair_tail(jobs, J, st, n_pairs, pair, blk, bpp, first_of_level, partials, tickets, seq, progress, s_T, pend_n);
}


// ---- late iterations of a level in ONE launch -----------------------------------------------------------------------------------
// From the iteration on from which a level's searches no longer go through the tiles nearly every query is certified: k_nn_iter and
// k_accumulate_matches then read the same source point, match and matched point one launch after the other, both at 2.7-3.5 TB/s of
// HBM traffic — bandwidth-bound, each. k_icp_late is the reduction pass's own shape (fat workgroups, 6-8 queries per thread, the 29
// sums in registers) that classifies while it streams: a certified query's residual is accumulated on the spot from what the
// certificate check loaded (one third of the two launches' bytes never moves), the uncertified few go to an LDS worklist and are
// walked 8 lanes per query (m3d_coop_query, the same code as k_nn_iter's), their residuals added by the group's first lane. Integer
// sums: any order, same bits. The last workgroup of the pair solves, as in k_accumulate_matches.
// (Round 1 tried this shape against a per-lane walk and a 27 + 6 + 22 us chain and dropped it; with the cooperative walk sharing its
// crowded rows and the chain at 21 + 18 us of bandwidth-bound launches it pays. Round 3 shipped a second, small-footprint variant for launches
// that share the GPU with other batches — 0.8 % of the headline on one box, profiles/r04_zoo_ab.txt — round 4 removed it.)
#ifdef M3D_LATE_STAMPS   // diagnosis build: wall-clock stamps (100 MHz) at k_icp_late's phase boundaries, every workgroup of the LAST launch (scripts/late_stamps.py)
__device__ unsigned long long g_late_stamp[4096][8];
extern "C" hipError_t m3d_debug_read_late(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_late_stamp), sizeof(unsigned long long) * 4096 * 8); }
#define LATE_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4096u) g_late_stamp[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define LATE_STAMP(k) ((void)0)
#endif
template <int METRIC>
__global__ __launch_bounds__(ICP_THREADS) void k_icp_late(const M3dJob* __restrict__ jobs, int n_pairs, int bpp, M3dNnArgs A, long long* __restrict__ partials,
                                                          unsigned int* __restrict__ tickets, unsigned int seq, unsigned long long* __restrict__ progress) {
    int pair, blk;
    m3d_map_block(n_pairs, bpp, pair, blk, A.rot);
    const M3dJob& J = jobs[pair];
    M3dPairState* st = A.states + pair;
    if (st->done || st->level_done) {
        if (blk == 0 && threadIdx.x == 0) m3d_report_progress(jobs, n_pairs, false, seq, progress);
        return;
    }
    LATE_STAMP(0);
    float R[9], tt[3];
    m3d_load_pose(st, R, tt);
    __shared__ double s_T[16];
    double t_pre = 0.0;
    if (threadIdx.x < 16) t_pre = st->T[threadIdx.x];
    const M3dGrid g = J.tgt.g;
    const m3d_gu4 tab = m3d_as_global(reinterpret_cast<const uint4*>(J.tgt.htab));
    const m3d_gf4 pts = m3d_as_global(J.tgt.pts), nrm = m3d_as_global(J.tgt.nrm), cbox = m3d_as_global(J.tgt.cbox);
    const m3d_gf3 src = m3d_as_global3(J.src);
    const m3d_gu32 bigcum = m3d_as_global(J.tgt.bigcum);
    const float dmax2 = J.dmax2;
    const int n = J.n_src;
    const int itq = st->iters & 31;
    M3D_GLOBAL m3d_i32x2* out = (M3D_GLOBAL m3d_i32x2*)(void M3D_GLOBAL*)(A.match + (size_t)pair * A.match_stride);
    M3D_GLOBAL long long* cache = (M3D_GLOBAL long long*)(void M3D_GLOBAL*)(A.cache + (size_t)pair * A.match_stride);
    __shared__ float s_ring[M3D_RING_FLOATS];
    m3d_ring_to_lds(s_ring, A.ring + (size_t)pair * M3D_RING_FLOATS);   // (published by the barrier below)
    const float cx = g.center[0], cy = g.center[1], cz = g.center[2];
    const float S[6] = { J.S[0], J.S[1], J.S[2], J.S[3], J.S[4], J.S[5] };
    constexpr int NACC = (METRIC == 1) ? 29 : 17;
    M3D_ACC<NACC> acc;
    acc.clear();
    __shared__ int s_cnt;
    constexpr int CAPS = M3D_LATE_CAP;
    __shared__ int s_list[CAPS];                 // query index | seeded << 31
    __shared__ float s_wu[3][CAPS];              // its transformed position
    __shared__ float s_wd[CAPS];                 // squared distance to its seed
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    LATE_STAMP(1);
#ifndef M3D_LATE_NB
#define M3D_LATE_NB 2   // queries per thread and trip (A/B builds)
#endif
    constexpr int NB = M3D_LATE_NB;
    const int stride = bpp * ICP_THREADS;
    for (int i0 = blk * ICP_THREADS + (int)threadIdx.x; i0 < n; i0 += NB * stride) {
        int m[NB], ce[NB]; float4 p[NB], q[NB], nq[NB]; m3d_f32x4 s0[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) { const int i = i0 + k * stride; const m3d_i32x2 e = (i < n) ? out[i] : (m3d_i32x2){ -1, 0 }; m[k] = e.x; ce[k] = e.y; }
#pragma unroll
        for (int k = 0; k < NB; k++) { const int i = i0 + k * stride; p[k] = (i < n) ? m3d_ld3(src, (size_t)i) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = i0 + k * stride;
            const size_t mm = (size_t)M3D_CHK(118, max(m[k], 0), max(g.n_valid, 1));
            s0[k] = (m3d_f32x4){ 0.f, 0.f, 0.f, 0.f }; q[k] = make_float4(0.f, 0.f, 0.f, 0.f); nq[k] = q[k];
            if (i < n && m[k] >= 0) {
                q[k] = m3d_ld(pts, mm);
                if (METRIC == 1) nq[k] = m3d_ld(nrm, mm);
                s0[k] = m3d_cert_state(ce[k], itq, s_ring, p[k]);   // {u0, sec}: where the query was at its last search, from that iteration's pose
            }
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = i0 + k * stride;
            if (i >= n) continue;
            const float ux = fmaf(R[0], p[k].x, fmaf(R[1], p[k].y, fmaf(R[2], p[k].z, tt[0])));
            const float uy = fmaf(R[3], p[k].x, fmaf(R[4], p[k].y, fmaf(R[5], p[k].z, tt[1])));
            const float uz = fmaf(R[6], p[k].x, fmaf(R[7], p[k].y, fmaf(R[8], p[k].z, tt[2])));
            float dseed = 0.f; bool certified = false;
            const int cls = m3d_classify_loaded(g, out, cache, i, m[k], ux, uy, uz, dmax2, A.certify, A.seed_reach, s0[k], q[k], dseed, certified, 0);
            if (certified) m3d_accumulate_match<METRIC, NACC>(acc, ux, uy, uz, q[k], dseed, nq[k], cx, cy, cz, S);   // (dseed: the same fma chain as the reduction pass's d2)
            else if (cls != 0) {
                const int w = atomicAdd(&s_cnt, 1);
                const int e = i | (cls == 1 ? (int)0x80000000u : 0);
                if (w < CAPS) { s_list[w] = e; s_wu[0][w] = ux; s_wu[1][w] = uy; s_wu[2][w] = uz; s_wd[w] = dseed; }
            }
        }
    }
    LATE_STAMP(2);
    __syncthreads();
    const int nW = min(s_cnt, M3D_LATE_CAP);   // (the cap cannot be exceeded: see launch_iteration)
    if (threadIdx.x < 16) s_T[threadIdx.x] = t_pre;
    long long* my_partial = partials + ((size_t)pair * bpp + blk) * M3D_PARTIAL_STRIDE;
    // MANY uncertified queries in this workgroup (config 2's 0.2 m point-to-point level: neighbours on a scan ring are 2 cm apart, half of the certificates fail
    // in every iteration — ~190 of a workgroup's 385 queries, six cooperative passes of 32): one query per lane, one pass. The streaming loop's sums are
    // parked in LDS meanwhile (the walk needs their registers). Not on a dense level, where one lane's walk is hundreds of candidates long.
#ifndef M3D_LATE_LANE_MIN
#define M3D_LATE_LANE_MIN 96
#endif
    __shared__ long long s_park[32];
    // (point-to-point only: with the 29 sums and the normals of point-to-plane the same code takes 193 VGPRs instead of 157 — two waves per SIMD instead of three
    // for every launch of the headline's kernel)
    // (and measured with it on the headline's batch: the two late iterations that have such workgroups took 92 / 82 instead of 75 / 57 us — its uncertified
    // queries sit next to obstacles, in long rows)
    if (METRIC == 0 && nW > M3D_LATE_LANE_MIN && !J.coop_always) {   // (uniform)
        block_reduce_to_global<NACC>(acc, st->sums, nullptr, s_park, nullptr);
        __syncthreads();   // (s_park is complete, the reduction's scratch is free again)
        for (int w = (int)threadIdx.x; w < nW; w += ICP_THREADS) {   // (the walk alone: no sum is live in it)
            const int e = s_list[w];
            const int qi = e & 0x7FFFFFFF;
            long long code = 0; float sec = 0.f;
            const int mq = m3d_nn27_walk(g, tab, pts, cbox, bigcum, s_wu[0][w], s_wu[1][w], s_wu[2][w], dmax2, e < 0, s_wd[w], code, sec, 0);
            out[qi] = (m3d_i32x2){ mq, m3d_cert_pack(sec, itq) };
            if (mq == M3D_NN_NONE_CACHED) cache[qi] = code;
            s_list[w] = mq;   // (this thread's own slot)
        }
        acc.clear();
        for (int w = (int)threadIdx.x; w < nW; w += ICP_THREADS) {
            const int mq = s_list[w];
            const float vx = s_wu[0][w], vy = s_wu[1][w], vz = s_wu[2][w];
            if (mq >= 0) {
                const float4 qm = m3d_ld(pts, (size_t)mq);
                const float4 nm = (METRIC == 1) ? m3d_ld(nrm, (size_t)mq) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float ex = vx - qm.x, ey = vy - qm.y, ez = vz - qm.z;
                const float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                m3d_accumulate_match<METRIC, NACC>(acc, vx, vy, vz, qm, d2, nm, cx, cy, cz, S);
            }
        }
        LATE_STAMP(3);
        block_reduce_to_global<NACC>(acc, st->sums, my_partial, nullptr, s_park);
    } else {
        const int sub = (int)threadIdx.x & 7;
        for (int base = 0; base < nW; base += 32) {   // uniform trip count: the shuffles inside need every lane
            const int w = base + ((int)threadIdx.x >> 3);
            const bool act = w < nW;
            const int e = act ? s_list[w] : 0;
            const int qi = e & 0x7FFFFFFF;
            const float vx = act ? s_wu[0][w] : 0.f, vy = act ? s_wu[1][w] : 0.f, vz = act ? s_wu[2][w] : 0.f;
            long long code; float sec;
            const int mq = m3d_coop_query(g, tab, pts, cbox, bigcum, dmax2, act, e < 0, vx, vy, vz, act ? s_wd[w] : 0.f, sub, code, sec, 0);
            if (act && sub == 0) {
                out[qi] = (m3d_i32x2){ mq, m3d_cert_pack(sec, itq) };
                if (mq == M3D_NN_NONE_CACHED) cache[qi] = code;
                if (mq >= 0) {
                    const float4 qm = m3d_ld(pts, (size_t)mq);
                    const float4 nm = (METRIC == 1) ? m3d_ld(nrm, (size_t)mq) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float ex = vx - qm.x, ey = vy - qm.y, ez = vz - qm.z;
                    const float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                    m3d_accumulate_match<METRIC, NACC>(acc, vx, vy, vz, qm, d2, nm, cx, cy, cz, S);
                }
            }
        }
        LATE_STAMP(3);
        block_reduce_to_global<NACC>(acc, st->sums, my_partial);
    }
    LATE_STAMP(4);
#ifdef M3D_LATE_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096u) g_late_stamp[blockIdx.x][6] = (unsigned long long)nW;
#endif
    m3d_pair_tail(jobs, J, st, n_pairs, pair, blk, bpp, 0, partials, tickets, seq, progress, s_T);
    LATE_STAMP(5);
}

// ---- introspection: NN of arbitrary queries --------------------------------------------------------
__global__ __launch_bounds__(256) void k_debug_nn(M3dLevelDev L, const float* __restrict__ q, int nq, float dmax2,
                                                  int32_t* __restrict__ out_idx, float* __restrict__ out_d2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float ux = q[3 * i], uy = q[3 * i + 1], uz = q[3 * i + 2];
    int32_t idx = -1; float d2 = 0.f;
    if (m3d_finite3(ux, uy, uz)) {
        float4 qq; float dd;
        const int j = m3d_nn27(L, ux, uy, uz, dmax2, dd, qq);
        if (j >= 0) { idx = (int32_t)(__float_as_uint(qq.w) & M3D_IDX_MASK); d2 = dd; }
    }
    out_idx[i] = idx; out_d2[i] = d2;
}

// ---- introspection: how many candidates the spec names for a query (all points of the 27 voxels around it) -----------------
// bench.py's gather-model bytes (12 N k + 8 * 27 N with k = mean candidates per query, measured)
__global__ __launch_bounds__(256) void k_debug_candidates(M3dLevelDev L, const float* __restrict__ q, int nq, int32_t* __restrict__ out_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float ux = q[3 * i], uy = q[3 * i + 1], uz = q[3 * i + 2];
    int32_t c = 0;
    M3dQuery Q;
    if (m3d_finite3(ux, uy, uz) && m3d_query_setup(L.g, ux, uy, uz, Q))
        for (int vz = Q.lo[2]; vz <= Q.hi[2]; vz++)
            for (int vy = Q.lo[1]; vy <= Q.hi[1]; vy++)
                for (int vx = Q.lo[0]; vx <= Q.hi[0]; vx++) { const uint2 r = m3d_find_voxel(L, vx, vy, vz); c += (int32_t)(r.y - r.x); }
    out_cnt[i] = c;
}
hipError_t m3d_launch_debug_candidates(hipStream_t s, const M3dLevelDev& L, const float* q_xyz, int nq, int32_t* out_cnt) {
    hipLaunchKernelGGL(k_debug_candidates, dim3((nq + 255) / 256), dim3(256), 0, s, L, q_xyz, nq, out_cnt);
    M3D_DBG(s, "k_debug_candidates");
    return hipGetLastError();
}

// ---- launchers ---------------------------------------------------------------------------------------
// blocks per ticket group of the reduction pass's "last block" detection: ~sqrt(blocks)
__host__ __device__ inline int m3d_ticket_group(int bpp) { int g = 1; while (g * g < bpp) g++; return g; }
// Workgroups per pair of the reduction pass (k_accumulate_matches, k_icp_late): a function of the batch alone. Its streaming loop's time is trips x
// loaded latency, so what matters is how many threads share the batch's queries: enough workgroups to put 2 on every CU (512: 8 pairs x 64 — 49,
// from "8 queries per thread", put two on 136 CUs and one on 120 and the launch took what two take) whatever the number of pairs: ONE 100 k-point
// pair used to get 49 workgroups for 256 CUs, with 256 a registration takes 0.86 instead of 1.01 ms. (Round 3 also took a third workgroup per CU when
// no other batch of the process was in flight — a process-global counter for 1 % of a serial step, profiles/r04_zoo_ab.txt: gone.)
// Bounds: at most 7 queries per thread (M3D_LATE_QPT: k_icp_late's worklist; the amortisation of the 29-term block reduction has little left to give
// beyond), at least 1.5 unless that contradicts the first; a thread's 32-bit running sums count their carries in 8-bit fields: never more than 128 per thread.
#ifndef M3D_ALONE_WGS
#define M3D_ALONE_WGS 768
#endif
#ifndef M3D_SHARED_WGS
#define M3D_SHARED_WGS 448
#endif
int m3d_acc_blocks(int max_n_src, int n_pairs, int alone) {
    const int b_min = (max_n_src + 256 * 128 - 1) / (256 * 128);
    int b = (max_n_src + 256 * M3D_LATE_QPT - 1) / (256 * M3D_LATE_QPT);
    if (b < 1) b = 1;
    if (b < b_min) b = b_min;
    if (n_pairs <= 0) return b;
    int hi = max_n_src / 384; if (hi < b) hi = b;
    // M3D_SHARED_WGS workgroups per batch by default — the default is sized for a GPU that other batches share (the headline keeps four chains in flight: 448 / 512 / 768
    // give 8693 / 8652 / 8366 registrations/s; fewer than 448 would exceed M3D_LATE_QPT queries per thread for 100 k-point pairs).
    // alone (m3dreg_set_latency_mode: the caller states that this handle's batches have the GPU to themselves): 768, all three a CU holds — serial steps +3.6 %
    // (round 5, profiles/r05_latency_mode.txt). A function of the batch and of that statement, never of what the process happens to have in flight.
    int t = ((alone ? M3D_ALONE_WGS : M3D_SHARED_WGS) + n_pairs - 1) / n_pairs;
    if (t > hi) t = hi;
    if (t > b) b = t;
    // the pair's last workgroup loads the partials 8 x M3D_TAIL_LOADS = 256 per round trip (m3d_pair_tail): a lone 100 k-point pair got 260 workgroups, and its tail a
    // second round trip for the last four (round 5) — a few workgroups over a multiple of 256 are not worth one
    if (b > 256 && b <= 320 && (long long)256 * 256 * M3D_LATE_QPT >= (long long)max_n_src) b = 256;
    return b;
}
int m3d_ticket_words(int n_pairs, int max_n_src, int alone) {
    const int bpp = m3d_acc_blocks(max_n_src, n_pairs, alone), gs = m3d_ticket_group(bpp), ng = (bpp + gs - 1) / gs;
    return n_pairs * (ng + 1) * 32;
}

// One Gauss-Newton iteration: k_nn_iter (classify; sparse blocks search cooperatively, dense blocks bin their queries by target
// tile), k_nn_tiles (the binned queries, from LDS), k_accumulate_matches (residuals, 29-term reduction; with fuse_solve the last
// block of every pair solves and updates the pose).
#ifndef M3D_COOP_LIST
#define M3D_COOP_LIST 64   // queries per workgroup of k_nn_coop_list (0: never launched; A/B builds)
#endif
static void m3d_launch_coop(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int bpp_s, int first_of_level, const M3dNnArgs& A, int list) {
    if (M3D_COOP_LIST == 0 || first_of_level || !list) {
        hipLaunchKernelGGL(k_nn_coop, dim3(8 * bpp_s * n_pairs), dim3(256), 0, s, d_jobs, n_pairs, 8 * bpp_s, first_of_level, A);
        M3D_DBG(s, "k_nn_coop");
    } else if (list == 2) {   // nearly every query certified (a level that started from a coarser level's result): 128 per workgroup
        hipLaunchKernelGGL(k_nn_coop_list<128>, dim3(2 * bpp_s * n_pairs), dim3(256), 0, s, d_jobs, n_pairs, 2 * bpp_s, first_of_level, A);
        M3D_DBG(s, "k_nn_coop_list<128>");
    } else {
        constexpr int QPB = M3D_COOP_LIST ? M3D_COOP_LIST : 64, PER = 256 / QPB;
        hipLaunchKernelGGL(k_nn_coop_list<QPB>, dim3(PER * bpp_s * n_pairs), dim3(256), 0, s, d_jobs, n_pairs, PER * bpp_s, first_of_level, A);
        M3D_DBG(s, "k_nn_coop_list");
    }
}
static void launch_iteration(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric, int first_of_level, const M3dNnWork& w,
                             hipEvent_t k0, hipEvent_t k1, long long* partials, unsigned int seq, unsigned long long* progress, int fuse_solve) {
    int bpp_s = (max_n_src + 255) / 256; if (bpp_s < 1) bpp_s = 1;
    M3dNnArgs A; A.match = w.match; A.match_stride = w.stride; A.cache = w.cache; A.ring = w.ring; A.certify = w.certify; A.seed_reach = w.seed_reach; A.coop_kernel = w.coop_kernel; A.lane_min = w.lane_min; A.states = w.states; A.rot = w.rot;
    A.tiles = w.tiles && first_of_level >= 0; A.ntile_max = w.ntile_max;
    {   // a work item of k_nn_tiles is one workgroup's pass over <= tile_chunk records: ONE 100 k-point pair makes 196 items of 512 for 256 CUs that hold three
        // workgroups each — a batch that makes fewer than 768 such items publishes items of 256 records, two lanes per record (config 3: 0.85 / 0.83 ms per
        // registration with 512 / 256, profiles/r04_zoo_ab.txt; 128: 0.855 in round 3). A function of the batch alone.
        const long long items = (long long)n_pairs * ((max_n_src + M3D_TILE_CHUNK - 1) / M3D_TILE_CHUNK);
        A.tile_chunk = items >= 768 ? M3D_TILE_CHUNK : M3D_TILE_CHUNK / 2;
    } A.rec = w.rec; A.recd = w.recd; A.rec_stride = w.rec_stride; A.tcnt = w.tcnt; A.cnt_stride = w.cnt_stride;
    A.witems = w.witems; A.wcount = w.wcount; A.wcap = w.wcap;
    if (k0) (void)hipEventRecord(k0, s);    // the correspondence step (bench.py roofline)
    // first_of_level: 1 = first iteration of a level, 0 = a later one, -1 / -2 = a late one: the searches that are left (a few per cent
    // of the queries, in blocks that are mostly certified) no longer go through the tiles, and k_nn_tiles is not launched
    const int late = first_of_level < 0;
    const bool fused_ok = first_of_level == -2;   // -2: a late iteration that may run as one launch
    if (late) first_of_level = 0;
    const int bpp_a = m3d_acc_blocks(max_n_src, n_pairs, w.alone);
    // A late iteration nobody brackets runs as ONE launch (k_icp_late). With an event bracket around the correspondence step (bench.py
    // samples some iterations; its untimed roofline step brackets all of them) the same iteration runs as the two-launch chain — same
    // bits — because the bracket's two halves do not exist inside a fused launch.
    if (fused_ok && !k0 && !k1 && fuse_solve && partials && (long long)bpp_a * ICP_THREADS * M3D_LATE_QPT >= (long long)max_n_src) {
        if (metric == 1) hipLaunchKernelGGL((k_icp_late<1>), dim3(bpp_a * n_pairs), dim3(ICP_THREADS), 0, s, d_jobs, n_pairs, bpp_a, A, partials, w.tickets, seq, progress);
        else hipLaunchKernelGGL((k_icp_late<0>), dim3(bpp_a * n_pairs), dim3(ICP_THREADS), 0, s, d_jobs, n_pairs, bpp_a, A, partials, w.tickets, seq, progress);
        M3D_DBG(s, "k_icp_late");
        return;
    }
    // the queries k_nn_iter<lean> cannot bin (M3D_NN_PENDING) are walked by the reduction pass itself (its <.., true> instantiation) — which therefore
    // must be the solving kind (the pair's last workgroup zeroes the pending count: m3d_pair_tail): the sums-only launch of m3dreg_debug_accumulate runs
    // the full k_nn_iter
    const bool lean_iter = w.tiles && !late && w.lean && fuse_solve && partials && A.coop_kernel != 2;
    const bool walk_in_acc = lean_iter;
    const bool coop_only = A.coop_kernel == 2;
    if (coop_only) {   // a dense level (every pair's, by the handle's last batch): no classifying launch that finds nothing to do, no tile launch that finds no item
        m3d_launch_coop(s, d_jobs, n_pairs, bpp_s, first_of_level, A, w.coop_list);
    } else if (lean_iter) {   // (every target of the batch has tiles: build_jobs checked)
        hipLaunchKernelGGL(k_nn_iter<true>, dim3(bpp_s * n_pairs), dim3(256), 0, s, d_jobs, n_pairs, bpp_s, first_of_level, A);
        M3D_DBG(s, "k_nn_iter<lean>");
    } else {
        hipLaunchKernelGGL(k_nn_iter<false>, dim3(bpp_s * n_pairs), dim3(256), 0, s, d_jobs, n_pairs, bpp_s, first_of_level, A);
        M3D_DBG(s, "k_nn_iter");
        if (A.coop_kernel) {   // a coarser level of a pyramid: where it is crowded (decided on the device, per pair) this kernel does the work, not the one above
            m3d_launch_coop(s, d_jobs, n_pairs, bpp_s, first_of_level, A, w.coop_list);
        }
    }
    if (w.tiles && !late && !coop_only) {
        hipLaunchKernelGGL(k_nn_tiles, dim3(M3D_TILE_GRID), dim3(M3D_TILE_THREADS), 0, s, d_jobs, first_of_level, A);
        M3D_DBG(s, "k_nn_tiles");
    }
    if (k1) (void)hipEventRecord(k1, s);
    unsigned int* gw = w.tiles ? w.tcnt : nullptr;   // the tiles' record counters and the work-item counter: zeroed here, behind their readers
#define M3D_ACC_LAUNCH(MET, WK) hipLaunchKernelGGL((k_accumulate_matches<MET, WK>), dim3(bpp_a * n_pairs), dim3(ICP_THREADS), 0, s, d_jobs, n_pairs, bpp_a, first_of_level, partials, \
                                                   w.tickets, w.states, seq, progress, fuse_solve, w.rot, gw, w.cnt_stride, w.ntile_max + (WK ? 0 : 1), w.wcount, A)
    if (metric == 1) { if (walk_in_acc) M3D_ACC_LAUNCH(1, true); else M3D_ACC_LAUNCH(1, false); }
    else { if (walk_in_acc) M3D_ACC_LAUNCH(0, true); else M3D_ACC_LAUNCH(0, false); }
#undef M3D_ACC_LAUNCH
    M3D_DBG(s, "k_accumulate_matches");
}

// Everything a job needs from the bucketing of its clouds, straight from the pipeline's device-side words (M3dLevelMeta): the
// target level's grid geometry, the fixed-point exponents (from its lbound), the source's finite-point count, and the error
// state of either cloud — the host enqueues a registration right behind the bucketing of its clouds and never reads any of it.
#define M3D_STATUS_BAD_CLOUD 5   // == M3DREG_BAD_CLOUD
__global__ void k_patch_jobs(M3dJob* __restrict__ jobs, int n_pairs, int cap_pairs, int n_levels) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs * n_levels) return;
    M3dJob& J = jobs[(size_t)(t / n_pairs) * cap_pairs + (t % n_pairs)];
    const M3dLevelMeta* M = reinterpret_cast<const M3dLevelMeta*>(J.tgt.dyn);
    const M3dLevelMeta* MS = reinterpret_cast<const M3dLevelMeta*>(J.src_dyn);
    M3dGrid g = M->g;
    g.hmask = M->dyn[1];
    g.hshift = (int32_t)M->dyn[2];
    J.tgt.g = g;
    if (M->dyn[7] == 0u) J.tgt.occ = nullptr;   // the grid has more bucket positions than the occupancy bitmap covers
    J.coop_always = m3d_dense_level((uint32_t)M->g.n_valid, M->dyn[0], (uint32_t)MS->g.n_valid, t / n_pairs) ? 1 : 0;
    J.n_src = MS->g.n_valid;
    int32_t e[6]; float S[6];
    m3d_fixed_exps(M->lbound, J.dmax, e, S);
    for (int k = 0; k < 6; k++) { J.exps[k] = e[k]; J.S[k] = S[k]; }
    if (t < n_pairs && J.ring) {   // slot 0 of the pose ring: the pose the batch starts from (iters = 0)
        const double* T = J.st->T;
        for (int rr = 0; rr < 3; rr++) { for (int c = 0; c < 3; c++) J.ring[3 * rr + c] = (float)T[c * 4 + rr]; J.ring[9 + rr] = (float)T[12 + rr]; }
    }
    if (t < n_pairs && (M->err || MS->err)) {   // a cloud that could not be bucketed: the registration ends before it starts
        J.st->status = M3D_STATUS_BAD_CLOUD;
        J.st->done = 1;
    }
}
hipError_t m3d_launch_patch_jobs(hipStream_t s, M3dJob* d_jobs, int n_pairs, int cap_pairs, int n_levels) {
    hipLaunchKernelGGL(k_patch_jobs, dim3((n_pairs * n_levels + 63) / 64), dim3(64), 0, s, d_jobs, n_pairs, cap_pairs, n_levels);
    M3D_DBG(s, "k_patch_jobs");
    return hipGetLastError();
}

hipError_t m3d_launch_accumulate_only(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric, const M3dNnWork& w) {
    launch_iteration(s, d_jobs, n_pairs, max_n_src, metric, 1, w, nullptr, nullptr, nullptr, 0u, nullptr, 0);   // sums by atomics into the state, no solve
    return hipGetLastError();
}

hipError_t m3d_launch_icp_iteration(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric, int first_of_level,
                                    const M3dNnWork& w, unsigned int seq, unsigned long long* progress, hipEvent_t k0, hipEvent_t k1) {
    launch_iteration(s, d_jobs, n_pairs, max_n_src, metric, first_of_level, w, k0, k1, w.partials, seq, progress, 1);
    return hipGetLastError();
}

hipError_t m3d_launch_debug_nn(hipStream_t s, const M3dLevelDev& L, const float* q_xyz, int nq, float dmax2, int32_t* out_idx,
                               float* out_d2) {
    hipLaunchKernelGGL(k_debug_nn, dim3((nq + 255) / 256), dim3(256), 0, s, L, q_xyz, nq, dmax2, out_idx, out_d2);
    M3D_DBG(s, "k_debug_nn");
    return hipGetLastError();
}
