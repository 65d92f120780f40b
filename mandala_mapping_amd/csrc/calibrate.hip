// calibrate.hip — SURVEY.md §8 row f2: the calibration cost function as hand-written HIP for gfx950.
// Reference: /root/reference/m3d/m3d_calibration/src/m3d_calibration_twiddle.cpp:199-308 (`testData`; the same lines
// at m3d_calibration_sa.cpp:199-277): transform every scan segment by original_Transform * laserOffsetMatrix (:229-230),
// split the points on the sign of a RAW coordinate (:234-266), pcl::VoxelGrid(0.1) both halves (:279-286), count the
// second half's voxel centroids that have no first-half centroid within 0.05 m (:288-304). The reference evaluates this
// ~10 times per twiddle sweep and 688 times per annealing run, each a full PCL pass over the sweep on one core.
//
// Here MANY parameter candidates are evaluated per launch (blockIdx.y = candidate): the composed 3x4 transforms come from
// the host (one per candidate and segment — cosf/sinf stay on the host so both sides of the parity test use the same
// libm), everything per point runs on the device:
//   k_cal_insert  one thread per raw point: transform, split, voxel = floor(u * (1.0f / 0.1f)), open-addressing hash on the
//                 64-bit (half, voxel) key, 2^-16 m fixed-point coordinate sums by 64-bit integer atomics — associative, so
//                 the centroid does not depend on arrival order (PCL's float sums follow an unstable std::sort).
//   k_cal_count   grid-stride over the table: every second-half centroid probes the (at most 2x2x2) first-half voxels that
//                 can hold a centroid within the radius; wave shuffle -> LDS -> ONE atomic per block (and at most 64 blocks
//                 per candidate: contended device-scope atomics retire at ~5 per microsecond on this chip).
// HBM-bound integer/byte work, no MFMA. The normative arithmetic is restated in oracle/m3d_cal_oracle.c (parity: the
// counts must be identical).
#include "m3d_kernels.h"

#define CAL_OFF 1048576
#define CAL_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ unsigned long long cal_key(int set, int ix, int iy, int iz) {
    return ((unsigned long long)set << 63) | ((unsigned long long)(uint32_t)(ix + CAL_OFF) << 42) |
           ((unsigned long long)(uint32_t)(iy + CAL_OFF) << 21) | (unsigned long long)(uint32_t)(iz + CAL_OFF);
}
__device__ __forceinline__ uint32_t cal_slot(unsigned long long key, int shift) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift); }

__global__ __launch_bounds__(256) void k_cal_insert(M3dCalArgs A) {
    const int k = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 p = A.pts[i];
    const int seg = (int)__float_as_uint(p.w);
    const float* m = A.mm + ((size_t)k * A.n_seg + seg) * 12;
    float u[3];
#pragma unroll
    for (int r = 0; r < 3; r++) u[r] = ((m[3 * r] * p.x + m[3 * r + 1] * p.y) + m[3 * r + 2] * p.z) + m[9 + r];   // :230
    const float raw = A.axis == 0 ? p.x : (A.axis == 1 ? p.y : p.z);
    const int set = (raw > 0.0f) ? 0 : 1;                                                                            // :234-266
    if (!m3d_finite3(u[0], u[1], u[2])) return;
    const float inv_leaf = 1.0f / 0.1f;
    int v[3]; long long q[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float f = floorf(u[r] * inv_leaf);
        if (!(f > -(float)CAL_OFF && f < (float)(CAL_OFF - 1))) { A.status[k] = 1; return; }
        v[r] = (int)f;
        q[r] = (long long)rintf(u[r] * 65536.0f);
    }
    const unsigned long long key = cal_key(set, v[0], v[1], v[2]);
    unsigned long long* keys = A.keys + (size_t)k * A.tsize;
    uint32_t h = cal_slot(key, A.tshift);
    for (;;) {
        const unsigned long long old = atomicCAS(&keys[h], CAL_EMPTY, key);
        if (old == CAL_EMPTY || old == key) break;
        h = (h + 1) & (A.tsize - 1);
    }
    const size_t e = (size_t)k * A.tsize + h;
    atomicAdd(&A.cnt[e], 1u);
    atomicAdd(reinterpret_cast<unsigned long long*>(&A.sums[3 * e]), (unsigned long long)q[0]);
    atomicAdd(reinterpret_cast<unsigned long long*>(&A.sums[3 * e + 1]), (unsigned long long)q[1]);
    atomicAdd(reinterpret_cast<unsigned long long*>(&A.sums[3 * e + 2]), (unsigned long long)q[2]);
}

__device__ __forceinline__ void cal_centroid(const M3dCalArgs& A, size_t e, float c[3]) {
    const double n = (double)A.cnt[e];
#pragma unroll
    for (int r = 0; r < 3; r++) c[r] = (float)(((double)A.sums[3 * e + r] / n) * (1.0 / 65536.0));
}

__global__ __launch_bounds__(256) void k_cal_count(M3dCalArgs A) {
    const int k = blockIdx.y;
    const unsigned long long* keys = A.keys + (size_t)k * A.tsize;
    const float inv_leaf = 1.0f / 0.1f;
    const float r2 = (float)(0.05 * 0.05);
    const float rr = 0.0505f;
    unsigned int c = 0, nf = 0, ns = 0;
    for (uint32_t h = blockIdx.x * 256 + threadIdx.x; h < A.tsize; h += gridDim.x * 256) {
        const unsigned long long key = keys[h];
        if (key == CAL_EMPTY) continue;
        if (!(key >> 63)) { nf++; continue; }
        ns++;
        float q[3];
        cal_centroid(A, (size_t)k * A.tsize + h, q);
        int lo[3], hi[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { lo[r] = (int)floorf((q[r] - rr) * inv_leaf); hi[r] = (int)floorf((q[r] + rr) * inv_leaf); }
        bool found = false;
        for (int ix = lo[0]; ix <= hi[0] && !found; ix++)
            for (int iy = lo[1]; iy <= hi[1] && !found; iy++)
                for (int iz = lo[2]; iz <= hi[2] && !found; iz++) {
                    if (ix <= -CAL_OFF || ix >= CAL_OFF - 1 || iy <= -CAL_OFF || iy >= CAL_OFF - 1 || iz <= -CAL_OFF || iz >= CAL_OFF - 1) continue;
                    const unsigned long long fk = cal_key(0, ix, iy, iz);
                    uint32_t g = cal_slot(fk, A.tshift);
                    unsigned long long o;
                    while ((o = keys[g]) != fk && o != CAL_EMPTY) g = (g + 1) & (A.tsize - 1);
                    if (o != fk) continue;
                    float f[3];
                    cal_centroid(A, (size_t)k * A.tsize + g, f);
                    const float dx = q[0] - f[0], dy = q[1] - f[1], dz = q[2] - f[2];
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < r2) found = true;
                }
        if (!found) c++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_down((int)c, o); nf += __shfl_down((int)nf, o); ns += __shfl_down((int)ns, o); }
    __shared__ unsigned int red[4][3];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = c; red[threadIdx.x >> 6][1] = nf; red[threadIdx.x >> 6][2] = ns; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned int v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v) atomicAdd(&A.result[3 * k + threadIdx.x], v);
    }
}

hipError_t m3d_launch_calibration(hipStream_t s, const M3dCalArgs& A, int n_candidates) {
    const size_t slots = (size_t)n_candidates * A.tsize;
    hipError_t e = hipMemsetAsync(A.keys, 0xFF, sizeof(unsigned long long) * slots, s);
    if (e == hipSuccess) e = hipMemsetAsync(A.cnt, 0, sizeof(uint32_t) * slots, s);
    if (e == hipSuccess) e = hipMemsetAsync(A.sums, 0, sizeof(long long) * 3 * slots, s);
    if (e == hipSuccess) e = hipMemsetAsync(A.result, 0, sizeof(unsigned int) * 3 * (size_t)n_candidates, s);
    if (e == hipSuccess) e = hipMemsetAsync(A.status, 0, sizeof(int) * (size_t)n_candidates, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cal_insert, dim3((A.n + 255) / 256, n_candidates), dim3(256), 0, s, A);
    M3D_DBG(s, "k_cal_insert");
    uint32_t cb = (A.tsize + 255u) / 256u;
    if (cb > 64u) cb = 64u;
    hipLaunchKernelGGL(k_cal_count, dim3(cb, n_candidates), dim3(256), 0, s, A);
    M3D_DBG(s, "k_cal_count");
    return hipGetLastError();
}
