// m3d_kernels.h — internal launcher interface between the C-ABI layer (m3dreg_api.cpp) and the HIP
// kernels (bucket.hip, icp.hip). Nothing here is exported from libm3dreg.so.
#pragma once
#include "m3d_device.h"

struct M3dBucketArgs {
    int n;
    const float *x, *y, *z;      // SoA input-order coordinates
    M3dGrid grid;
    int sort_passes;             // 8-bit LSD passes needed to order the keys
    uint32_t* keys;              // [n] out: key per input point
    uint32_t *ka, *va, *kb, *vb; // [n] sort ping-pong workspace
    uint32_t* hist;              // [256 * tiles]
    uint32_t* skey_out;          // [n] out: sorted keys
    uint32_t* perm_out;          // [n] out: permutation
    const float4* nrm_in;        // [n] normals in input order or null
    float4* pts;                 // [n] out
    float4* nrm;                 // [n] out or null
    uint2* htab;                 // [hmask+1] out
    uint32_t* n_cells;           // out: occupied voxel count
};

float m3d_unord_f32(uint32_t u);
int m3d_sort_tiles(int n);
hipError_t m3d_launch_decode_aabb(hipStream_t s, const uint8_t* raw, int n, int step, int ox, int oy, int oz, float* x, float* y,
                                  float* z, uint32_t* aabb);
hipError_t m3d_launch_bucket_level(hipStream_t s, const M3dBucketArgs& a);
// mom: workspace of 10 * n_valid int64 (per-voxel moments, indexed by the voxel's first sorted position)
hipError_t m3d_launch_normals(hipStream_t s, const M3dLevelDev& L, const uint32_t* skey, long long* mom, float plane_ratio, int min_pts,
                              float min_spread, float4* nrm_in, int n);
hipError_t m3d_launch_export_sorted(hipStream_t s, const float4* pts, const float4* nrm, int n, float* xyz, float* nxyz);

// icp.hip
// e0/e1 (optional): events recorded immediately before / after the k_icp_accumulate launch
hipError_t m3d_launch_icp_iteration(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric, int first_of_level,
                                    hipEvent_t e0, hipEvent_t e1);
hipError_t m3d_launch_accumulate_only(hipStream_t s, const M3dJob* d_jobs, int n_pairs, int max_n_src, int metric);
hipError_t m3d_launch_debug_nn(hipStream_t s, const M3dLevelDev& L, const float* q_xyz, int nq, float dmax2, int32_t* out_idx,
                               float* out_d2);
