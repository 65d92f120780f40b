// map.hip — SURVEY.md §8 row f4: a persistent voxel-deduplicated map in HBM (BASELINE config 5: a dense aggregated
// reference that a live scan is registered against). No reference source exists for this step (north_star names it; the
// reference's gpu_6dslam is an empty submodule); the normative behaviour is DESIGN.md §8 / oracle/m3d_map_oracle.c:
//   insert(scan, T): u = R p + t (the fma chain of spec row a5) for every finite point of the scan IN INPUT ORDER; a point
//   is kept iff its dedup voxel floor(u * inv_leaf) is not occupied by an EARLIER insert and no earlier point of the same
//   scan falls into it (lowest input index wins); kept points are appended in input order.
// A registered scan therefore only adds what is new — the map stays bounded by the mapped surface, not by the number of
// scans — and the whole thing never leaves the device: m3dmap_as_cloud buckets the point buffer in place.
//   k_map_mark     transform + 64-bit voxel key + CAS insert into the persistent occupancy table; slots created by this
//                  insert (epoch tag) collect the lowest input index by atomicMin — deterministic, unlike "first CAS wins"
//   k_map_count / k_map_scan / k_map_scatter   stable compaction of the winners (wave64 ballot ranks, per-block offsets,
//                  one-workgroup scan: no contended atomics) behind the points already in the map
#include "m3d_kernels.h"

#define MAP_OFF 1048576
#define MAP_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ bool map_point(const M3dMapArgs& A, int i, float4& u, unsigned long long& key) {
    const float4 p = A.src[i];
    if (!m3d_finite3(p.x, p.y, p.z)) return false;
    u.x = fmaf(A.R[0], p.x, fmaf(A.R[1], p.y, fmaf(A.R[2], p.z, A.t[0])));
    u.y = fmaf(A.R[3], p.x, fmaf(A.R[4], p.y, fmaf(A.R[5], p.z, A.t[1])));
    u.z = fmaf(A.R[6], p.x, fmaf(A.R[7], p.y, fmaf(A.R[8], p.z, A.t[2])));
    u.w = 0.0f;
    if (!m3d_finite3(u.x, u.y, u.z)) return false;
    const float fx = floorf(u.x * A.inv_leaf), fy = floorf(u.y * A.inv_leaf), fz = floorf(u.z * A.inv_leaf);
    const float lim = (float)(MAP_OFF - 1);
    if (!(fx > -lim && fx < lim && fy > -lim && fy < lim && fz > -lim && fz < lim)) { A.flags[1] = 1u; return false; }   // beyond +-2^20 voxels: dropped, reported
    key = ((unsigned long long)(uint32_t)((int)fx + MAP_OFF) << 42) | ((unsigned long long)(uint32_t)((int)fy + MAP_OFF) << 21) |
          (unsigned long long)(uint32_t)((int)fz + MAP_OFF);
    return true;
}
__device__ __forceinline__ uint32_t map_slot(unsigned long long key, int shift) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift); }

__global__ __launch_bounds__(256) void k_map_mark(M3dMapArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    float4 u; unsigned long long key;
    if (!map_point(A, i, u, key)) { A.slot_of[i] = 0xFFFFFFFFu; return; }
    uint32_t h = map_slot(key, A.tshift);
    for (uint32_t probes = 0;; probes++) {
        const unsigned long long old = atomicCAS(&A.keys[h], MAP_EMPTY, key);
        if (old == MAP_EMPTY) { A.epoch[h] = A.cur_epoch; break; }   // created by this insert (read by the NEXT kernel only)
        if (old == key) break;
        h = (h + 1) & (A.tsize - 1);
        if (probes >= A.tsize) { A.flags[0] = 1u; A.slot_of[i] = 0xFFFFFFFFu; return; }   // table full: reported, nothing lost silently
    }
    A.slot_of[i] = h;
    // owner starts at ~0 (table clear). It only matters during the insert that CREATES the slot (keys are never removed, and
    // map_keep ignores slots of earlier epochs), so it never needs a reset.
    atomicMin(&A.owner[h], (uint32_t)i);
}

__device__ __forceinline__ bool map_keep(const M3dMapArgs& A, int i) {
    if (i >= A.n) return false;
    const uint32_t h = A.slot_of[i];
    return h != 0xFFFFFFFFu && A.epoch[h] == A.cur_epoch && A.owner[h] == (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_map_count(M3dMapArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(map_keep(A, i));
    __shared__ uint32_t w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) A.block_counts[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(1024) void k_map_scan(M3dMapArgs A, int nblocks) {
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
        if (total > A.capacity) A.flags[3] = 1u;   // point buffer full: reported by the host, nothing is written out of bounds
    }
}

__global__ __launch_bounds__(256) void k_map_scatter(M3dMapArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool keep = map_keep(A, i);
    const unsigned long long b = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t w[4];
    if (lane == 0) w[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t off = A.block_counts[blockIdx.x];
    for (int k = 0; k < wave; k++) off += w[k];
    off += (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (keep) {
        float4 u; unsigned long long key;
        map_point(A, i, u, key);
        if (off < A.capacity) A.out[off] = u;
    }
}

hipError_t m3d_launch_map_insert(hipStream_t s, const M3dMapArgs& A) {
    const int nblocks = (A.n + 255) / 256;
    hipLaunchKernelGGL(k_map_mark, dim3(nblocks), dim3(256), 0, s, A);
    M3D_DBG(s, "k_map_mark");
    hipLaunchKernelGGL(k_map_count, dim3(nblocks), dim3(256), 0, s, A);
    M3D_DBG(s, "k_map_count");
    hipLaunchKernelGGL(k_map_scan, dim3(1), dim3(1024), 0, s, A, nblocks);
    M3D_DBG(s, "k_map_scan");
    hipLaunchKernelGGL(k_map_scatter, dim3(nblocks), dim3(256), 0, s, A);
    M3D_DBG(s, "k_map_scatter");
    return hipGetLastError();
}
