// debug.hip — diagnosis only (M3DREG_POISON, DESIGN.md §8 "the unexplained GPU memory fault"): fills a device allocation with a chosen word or with pseudo-random words,
// so that any kernel that READS memory the library never wrote meets hostile contents (indices of 2^31, NaNs, all-ones keys) in every run instead of in one
// process in five hundred — the bit-exact parity tests then fail, or the process faults, deterministically. Never launched unless the environment asks.
#include "m3d_kernels.h"

__global__ __launch_bounds__(256) void k_poison(uint32_t* __restrict__ p, size_t words, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;   // (a hash of the word's index: every run sees the same garbage)
        x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
        p[i] = x;
    }
}

hipError_t m3d_launch_poison(hipStream_t s, void* p, size_t bytes, uint32_t seed) {
    const size_t words = bytes / 4;
    if (!words) return hipSuccess;
    const size_t blocks = (words + 255) / 256;
    hipLaunchKernelGGL(k_poison, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, static_cast<uint32_t*>(p), words, seed);
    return hipGetLastError();
}
