// icp.hip — SURVEY.md §8 rows a5-a8: source transform, 27-voxel nearest-neighbour correspondence,
// point-to-point / point-to-plane residuals, the 29-term normal-equation reduction, the 6x6 solve and
// the SE(3) update, hand-written for gfx950. No reference source exists for this path (gpu_6dslam is
// an empty submodule); the nearest in-tree analogue of the neighbour query is the per-point
// KdTreeFLANN::radiusSearch loop at /root/reference/m3d/m3d_calibration/src/m3d_calibration_twiddle.cpp:292-304.
// The normative arithmetic is DESIGN.md §Spec; oracle/m3d_oracle.c restates it on the CPU and the
// parity tests compare every output bit for bit.
//
// Hardware mapping: source points stream coalesced from SoA arrays; candidates are 16-B gathers from
// the cell-sorted float4 array (L2/MALL-resident); the 29 sums are int64 fixed point, reduced with
// wave64 shuffles, then LDS across the 4 waves, then one 64-bit integer atomic per term and block —
// associative, so the result does not depend on launch geometry. No MFMA: there is no contraction.
#include "m3d_kernels.h"

#define ICP_THREADS 256
#define ICP_WAVES (ICP_THREADS / 64)

// ---- a6: exact NN over the 27 voxels around u --------------------------------------------------
// Returns the sorted position of the match (or -1) and its squared distance / input index.
// Pruning only ever skips a voxel whose box is provably farther than the current best (or than
// d_max), so the result is identical to the exhaustive walk of the oracle.
__device__ __forceinline__ int m3d_nn27(const M3dLevelDev& L, float ux, float uy, float uz, float dmax2, float& out_d2,
                                        float4& out_q) {
    const M3dGrid& g = L.g;
    const float fx = m3d_cell_f(ux, g.mn[0], g.inv_leaf);
    const float fy = m3d_cell_f(uy, g.mn[1], g.inv_leaf);
    const float fz = m3d_cell_f(uz, g.mn[2], g.inv_leaf);
    if (!(fx >= -1.0f && fx <= (float)g.dims[0])) return -1;
    if (!(fy >= -1.0f && fy <= (float)g.dims[1])) return -1;
    if (!(fz >= -1.0f && fz <= (float)g.dims[2])) return -1;
    const int icx = (int)fx, icy = (int)fy, icz = (int)fz;
    // distance from u to the lower / upper faces of its own voxel (conservative by prune_slack)
    const float rx = (ux - g.mn[0]) - fx * g.leaf, ry = (uy - g.mn[1]) - fy * g.leaf, rz = (uz - g.mn[2]) - fz * g.leaf;
    const float lox = fmaxf(rx - g.prune_slack, 0.f), hix = fmaxf((g.leaf - rx) - g.prune_slack, 0.f);
    const float loy = fmaxf(ry - g.prune_slack, 0.f), hiy = fmaxf((g.leaf - ry) - g.prune_slack, 0.f);
    const float loz = fmaxf(rz - g.prune_slack, 0.f), hiz = fmaxf((g.leaf - rz) - g.prune_slack, 0.f);
    int best = -1;
    float bd = 3.0e38f;
    uint32_t boi = 0;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    float bound = dmax2 * 1.0001f;
    for (int k = 0; k < 27; k++) {
        int idx = k + 13; if (idx >= 27) idx -= 27;     // own voxel first
        const int dz = idx / 9 - 1, dy = (idx / 3) % 3 - 1, dx = idx % 3 - 1;
        const int cx = icx + dx, cy = icy + dy, cz = icz + dz;
        if (cx < 0 || cx >= g.dims[0] || cy < 0 || cy >= g.dims[1] || cz < 0 || cz >= g.dims[2]) continue;
        const float gx = dx < 0 ? lox : (dx > 0 ? hix : 0.f);
        const float gy = dy < 0 ? loy : (dy > 0 ? hiy : 0.f);
        const float gz = dz < 0 ? loz : (dz > 0 ? hiz : 0.f);
        if (gx * gx + gy * gy + gz * gz > bound) continue;
        const uint32_t key = (uint32_t)cx | ((uint32_t)cy << g.sy) | ((uint32_t)cz << g.sz);
        int t = m3d_find_cell(L.htab, g.hmask, g.hshift, key);
        if (t < 0) continue;
        for (;;) {
            const float4 q = L.pts[t];
            const float ex = ux - q.x, ey = uy - q.y, ez = uz - q.z;
            const float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            const uint32_t w = __float_as_uint(q.w);
            const uint32_t oi = w & ~M3D_LAST_FLAG;
            if (best < 0 || d2 < bd || (d2 == bd && oi < boi)) { best = t; bd = d2; boi = oi; bq = q; }
            if (w & M3D_LAST_FLAG) break;
            t++;
        }
        bound = fminf(bound, bd * 1.0001f);
    }
    if (best < 0 || !(bd <= dmax2)) return -1;
    out_d2 = bd;
    out_q = bq;
    return best;
}

__device__ __forceinline__ long long m3d_quant(float term, float scale) {
    return (long long)(int)rintf(term * scale);
}

// slot of H(k,l), k <= l, in the row-major upper triangle
__host__ __device__ constexpr int hslot21(int k, int l) { return k * 6 - (k * (k - 1)) / 2 + (l - k); }

template <int NACC>
__device__ __forceinline__ void block_reduce_to_global(long long (&acc)[NACC], long long* __restrict__ sums) {
    __shared__ long long red[ICP_WAVES][NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        long long v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        long long v = 0;
#pragma unroll
        for (int w = 0; w < ICP_WAVES; w++) v += red[w][threadIdx.x];
        if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&sums[threadIdx.x]), (unsigned long long)v);
    }
}

// One linearisation of every pair of the batch: grid = (blocks per pair, pairs).
template <int METRIC>
__global__ __launch_bounds__(ICP_THREADS) void k_icp_accumulate(const M3dJob* __restrict__ jobs, int first_of_level) {
    const M3dJob& J = jobs[blockIdx.y];
    M3dPairState* st = J.st;
    if (st->done || (!first_of_level && st->level_done)) return;
    // current pose rounded to float (spec: R row-major from the column-major double pose)
    float R[9], tt[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) R[3 * r + c] = (float)st->T[c * 4 + r];
        tt[r] = (float)st->T[12 + r];
    }
    const M3dLevelDev& L = J.tgt;
    const float cx = L.g.center[0], cy = L.g.center[1], cz = L.g.center[2];
    const float dmax2 = J.dmax2;
    const float S0 = J.S[0], S1 = J.S[1], S2 = J.S[2], S3 = J.S[3], S4 = J.S[4], S5 = J.S[5];

    constexpr int NACC = (METRIC == 1) ? 29 : 17;
    long long acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0;

    const int n = J.n_src;
    for (int i = blockIdx.x * ICP_THREADS + threadIdx.x; i < n; i += gridDim.x * ICP_THREADS) {
        const float px = J.sx[i], py = J.sy[i], pz = J.sz[i];
        if (!m3d_finite3(px, py, pz)) continue;
        const float ux = fmaf(R[0], px, fmaf(R[1], py, fmaf(R[2], pz, tt[0])));
        const float uy = fmaf(R[3], px, fmaf(R[4], py, fmaf(R[5], pz, tt[1])));
        const float uz = fmaf(R[6], px, fmaf(R[7], py, fmaf(R[8], pz, tt[2])));
        if (!m3d_finite3(ux, uy, uz)) continue;
        float d2; float4 q;
        const int j = m3d_nn27(L, ux, uy, uz, dmax2, d2, q);
        if (j < 0) continue;
        const float ex = ux - q.x, ey = uy - q.y, ez = uz - q.z;
        const float wx = ux - cx, wy = uy - cy, wz = uz - cz;
        if (METRIC == 1) {
            const float4 nq = L.nrm[j];
            const float nx = nq.x, ny = nq.y, nz = nq.z;
            if (nx == 0.0f && ny == 0.0f && nz == 0.0f) continue;
            float Jv[6];
            Jv[0] = wy * nz - wz * ny; Jv[1] = wz * nx - wx * nz; Jv[2] = wx * ny - wy * nx;
            Jv[3] = nx; Jv[4] = ny; Jv[5] = nz;
            const float r = nx * ex + ny * ey + nz * ez;
#pragma unroll
            for (int k = 0; k < 6; k++)
#pragma unroll
                for (int l = k; l < 6; l++) {
                    const float sc = (l < 3) ? S0 : (k < 3 ? S1 : S2);
                    acc[hslot21(k, l)] += m3d_quant(Jv[k] * Jv[l], sc);
                }
#pragma unroll
            for (int k = 0; k < 3; k++) acc[21 + k] += m3d_quant(Jv[k] * r, S3);
#pragma unroll
            for (int k = 3; k < 6; k++) acc[21 + k] += m3d_quant(Jv[k] * r, S4);
            acc[27] += m3d_quant(r * r, S5);
            acc[28] += 1;
        } else {
            // 17 running sums: Hrr(6) | sum w (3) | g(6) | ssr | count
            acc[0] += m3d_quant(wy * wy + wz * wz, S0);
            acc[1] += m3d_quant(-(wx * wy), S0);
            acc[2] += m3d_quant(-(wx * wz), S0);
            acc[3] += m3d_quant(wx * wx + wz * wz, S0);
            acc[4] += m3d_quant(-(wy * wz), S0);
            acc[5] += m3d_quant(wx * wx + wy * wy, S0);
            acc[6] += m3d_quant(wx, S1);
            acc[7] += m3d_quant(wy, S1);
            acc[8] += m3d_quant(wz, S1);
            acc[9] += m3d_quant(wy * ez - wz * ey, S3);
            acc[10] += m3d_quant(wz * ex - wx * ez, S3);
            acc[11] += m3d_quant(wx * ey - wy * ex, S3);
            acc[12] += m3d_quant(ex, S4);
            acc[13] += m3d_quant(ey, S4);
            acc[14] += m3d_quant(ez, S4);
            acc[15] += m3d_quant(d2, S5);
            acc[16] += 1;
        }
    }
    // point-to-plane: the 29 spec slots; point-to-point: 17 transport slots (see expand_pt2pt)
    block_reduce_to_global<NACC>(acc, st->sums);
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
    double Lm[6][6], D[6];
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
        for (int k = 0; k < j; k++) d = d - Lm[j][k] * Lm[j][k] * D[k];
        if (!(d > tol)) return 3;
        D[j] = d;
        for (int i = j + 1; i < 6; i++) {
            double v = A[i][j];
            for (int k = 0; k < j; k++) v = v - Lm[i][k] * Lm[j][k] * D[k];
            Lm[i][j] = v / d;
        }
    }
    double y[6], x[6];
    for (int i = 0; i < 6; i++) { double v = b[i]; for (int k = 0; k < i; k++) v = v - Lm[i][k] * y[k]; y[i] = v; }
    for (int i = 0; i < 6; i++) y[i] = y[i] / D[i];
    for (int i = 5; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 6; k++) v = v - Lm[k][i] * x[k]; x[i] = v; }
    const double w0 = x[0], w1 = x[1], w2 = x[2], v0 = x[3], v1 = x[4], v2 = x[5];
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    const double tr2 = v0 * v0 + v1 * v1 + v2 * v2;
    th2_out = th2; tr2_out = tr2;
    if (!(th2 <= 4.0) || !(tr2 < 1e300)) return 4;
    double sa = 1.0, sb = 1.0, sc = 1.0;
    for (int k = 12; k >= 1; k--) {
        sa = 1.0 - th2 * sa / (double)((2 * k) * (2 * k + 1));
        sb = 1.0 - th2 * sb / (double)((2 * k + 1) * (2 * k + 2));
        sc = 1.0 - th2 * sc / (double)((2 * k + 2) * (2 * k + 3));
    }
    const double Ac = sa, Bc = sb / 2.0, Cc = sc / 6.0;
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
__global__ __launch_bounds__(64) void k_solve_update(const M3dJob* __restrict__ jobs, int n_pairs, int first_of_level) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const M3dJob& J = jobs[p];
    M3dPairState* st = J.st;
    if (st->done || (!first_of_level && st->level_done)) return;
    long long sums[M3D_NSUMS];
    if (J.metric == 1) { for (int i = 0; i < M3D_NSUMS; i++) sums[i] = st->sums[i]; }
    else { long long in[17]; for (int i = 0; i < 17; i++) in[i] = st->sums[i]; expand_pt2pt(in, sums); }
    for (int i = 0; i < M3D_NSUMS; i++) st->sums[i] = 0;
    int exps[6];
    for (int i = 0; i < 6; i++) exps[i] = J.exps[i];
    double T[16];
    for (int i = 0; i < 16; i++) T[i] = st->T[i];
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
        if (j >= 0) { idx = (int32_t)(__float_as_uint(qq.w) & ~M3D_LAST_FLAG); d2 = dd; }
    }
    out_idx[i] = idx; out_d2[i] = d2;
}

// ---- launchers ---------------------------------------------------------------------------------------
static inline int icp_blocks(int max_n_src) {
    int b = (max_n_src + ICP_THREADS - 1) / ICP_THREADS;
    return b < 1 ? 1 : b;
}

hipError_t m3d_launch_accumulate_only(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric) {
    dim3 grid(icp_blocks(max_n_src), n_pairs);
    if (metric == 1) hipLaunchKernelGGL(k_icp_accumulate<1>, grid, dim3(ICP_THREADS), 0, s, d_jobs, 1);
    else hipLaunchKernelGGL(k_icp_accumulate<0>, grid, dim3(ICP_THREADS), 0, s, d_jobs, 1);
    return hipGetLastError();
}

hipError_t m3d_launch_icp_iteration(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric, int first_of_level,
                                    hipEvent_t e0, hipEvent_t e1) {
    dim3 grid(icp_blocks(max_n_src), n_pairs);
    if (e0) (void)hipEventRecord(e0, s);
    if (metric == 1) hipLaunchKernelGGL(k_icp_accumulate<1>, grid, dim3(ICP_THREADS), 0, s, d_jobs, first_of_level);
    else hipLaunchKernelGGL(k_icp_accumulate<0>, grid, dim3(ICP_THREADS), 0, s, d_jobs, first_of_level);
    if (e1) (void)hipEventRecord(e1, s);
    hipLaunchKernelGGL(k_solve_update, dim3((n_pairs + 63) / 64), dim3(64), 0, s, d_jobs, n_pairs, first_of_level);
    return hipGetLastError();
}

hipError_t m3d_launch_debug_nn(hipStream_t s, const M3dLevelDev& L, const float* q_xyz, int nq, float dmax2, int32_t* out_idx,
                               float* out_d2) {
    hipLaunchKernelGGL(k_debug_nn, dim3((nq + 255) / 256), dim3(256), 0, s, L, q_xyz, nq, dmax2, out_idx, out_d2);
    return hipGetLastError();
}
