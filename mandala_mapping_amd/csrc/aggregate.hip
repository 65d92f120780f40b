// aggregate.hip — SURVEY.md §8 row f1: the step BEFORE the registration path, on the device, so that sweeps
// are born in HBM and never cross PCIe as full clouds. Follows
//   /root/reference/m3d/m3d_aggregator/src/m3d_aggregator.cpp:53-73  (addPoints: p' = T*p in double via tf::Transform,
//        rounded to float, kept iff OUTSIDE the self-filter box) and
//   :256-288 (rotLaserScanCallback: ang = angle_min + i*angle_increment in float, x = cos(ang)*r, y = sin(ang)*r, z = 0).
// Kept points are appended in input order (stable compaction: wave64 ballot ranks + per-block offsets), exactly
// like the reference's push_back loop, into a pcl::PointXYZ-layout buffer (16 B per point) that
// m3dreg_cloud_create can bucket in place. The angular-distance bookkeeping (:75-87) is scalar per message
// and stays on the host (m3dreg_api.cpp).
#include "m3d_kernels.h"

__device__ __forceinline__ bool agg_point(const M3dAggArgs& A, int i, float4& pp) {
    float px, py, pz;
    if (A.mode == 0) {
        const uint8_t* p = A.raw + (size_t)i * A.step;
        px = *reinterpret_cast<const float*>(p + A.ox);
        py = *reinterpret_cast<const float*>(p + A.oy);
        pz = *reinterpret_cast<const float*>(p + A.oz);
    } else {
        const float ang = A.angle_min + (float)i * A.angle_inc;   // :272
        const float dist = A.ranges[i];
        // :281-283 `cos(ang)*dist` with float operands: mode 1 = C's double cos(double), the product formed in double and rounded once into the
        // float field (the default: what the pre-GCC-6 toolchains of this ROS1 code resolve the unqualified call to); mode 2 = the float overload (Spec §Trig, m3d_device.h: m3d_sincosf_spec)
        if (A.mode == 1) { px = (float)(cos((double)ang) * (double)dist); py = (float)(sin((double)ang) * (double)dist); }
        else { float sn, cs; m3d_sincosf_spec(ang, sn, cs); px = cs * dist; py = sn * dist; }   // (Spec §Trig: the library's own float sine / cosine, bit-identical to the oracle's)
        pz = 0.0f;
    }
    const double x = (double)px, y = (double)py, z = (double)pz;
    // tf::Transform::operator*(Vector3): basis row dot x + origin, all double (built with -ffp-contract=off)
    const double p0 = A.m[0] * x + A.m[1] * y + A.m[2] * z + A.o[0];
    const double p1 = A.m[3] * x + A.m[4] * y + A.m[5] * z + A.o[1];
    const double p2 = A.m[6] * x + A.m[7] * y + A.m[8] * z + A.o[2];
    pp = make_float4((float)p0, (float)p1, (float)p2, 0.0f);
    // :65-70 keep iff any coordinate lies outside [down, up]
    return ((double)pp.x > A.bb[0]) || ((double)pp.x < A.bb[1]) || ((double)pp.y > A.bb[2]) || ((double)pp.y < A.bb[3]) ||
           ((double)pp.z > A.bb[4]) || ((double)pp.z < A.bb[5]);
}

__global__ __launch_bounds__(256) void k_agg_count(M3dAggArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 pp;
    const bool keep = (i < A.n) && agg_point(A, i, pp);
    const unsigned long long b = __ballot(keep);
    __shared__ uint32_t w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) A.block_counts[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// one workgroup: exclusive scan of the block counts, appended behind the points already aggregated
__global__ __launch_bounds__(1024) void k_agg_scan(M3dAggArgs A, int nblocks) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b = t * per, e = min(b + per, nblocks);
    uint32_t s = 0;
    for (int i = b; i < e; i++) s += A.block_counts[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        uint32_t v = (t >= o) ? part[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t base = A.count[0];
    uint32_t run = base + part[t] - s;
    for (int i = b; i < e; i++) { uint32_t v = A.block_counts[i]; A.block_counts[i] = run; run += v; }
    __syncthreads();
    if (t == 1023) {
        const uint32_t total = base + part[1023];
        A.count[0] = total <= A.capacity ? total : A.capacity;
        if (total > A.capacity) A.count[1] = 1u;   // overflow flag: the host reports it, nothing is written out of bounds
    }
}

__global__ __launch_bounds__(256) void k_agg_scatter(M3dAggArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool keep = (i < A.n) && agg_point(A, i, pp);
    const unsigned long long b = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t w[4];
    if (lane == 0) w[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t off = A.block_counts[blockIdx.x];
    for (int k = 0; k < wave; k++) off += w[k];
    off += (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (keep && off < A.capacity) A.out[off] = pp;
}

hipError_t m3d_launch_aggregate(hipStream_t s, const M3dAggArgs& A) {
    const int nblocks = (A.n + 255) / 256;
    hipLaunchKernelGGL(k_agg_count, dim3(nblocks), dim3(256), 0, s, A);
    M3D_DBG(s, "k_agg_count");
    hipLaunchKernelGGL(k_agg_scan, dim3(1), dim3(1024), 0, s, A, nblocks);
    M3D_DBG(s, "k_agg_scan");
    hipLaunchKernelGGL(k_agg_scatter, dim3(nblocks), dim3(256), 0, s, A);
    M3D_DBG(s, "k_agg_scatter");
    return hipGetLastError();
}
