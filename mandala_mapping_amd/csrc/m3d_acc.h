// m3d_acc.h — a thread's running sums in the reduction kernels (k_accumulate_matches, k_icp_late). Plain C++ apart from the M3D_HD
// qualifier, so that tests/test_acc32.py can compile it with g++ and check the carry arithmetic against 64-bit sums on the CPU.
#pragma once
#if defined(__HIPCC__)
#define M3D_HD __device__ __forceinline__
#else
#define M3D_HD inline
#endif
// A thread's running sums. Every term is an int32 (m3d_quant) and a thread adds a few dozen of them at most, yet 64-bit registers for
// each of the 29 sums were 58 of the reduction kernels' VGPRs — and those kernels are chains of dependent gathers whose speed is the
// number of waves the register file holds (k_icp_late at 1 / 2 waves per SIMD: 65 / 42 us). M3dAcc32: 32-bit running sums of the terms
// BIASED by 2^31 (sign bit flipped: every addend is then non-negative and a wrap is the unsigned carry), which are exact modulo 2^32, and
// a carry count per sum in an 8-bit field (four per register: 8 more VGPRs); wide() puts them together and takes the bias out again
// (2^31 x the number of terms, which is the count slot): the same integers, bit for bit. A field holds 255 carries: a thread adds at
// most 128 streamed terms (m3d_acc_blocks) or 7 streamed + 56 walked ones (M3D_LATE_CAP / 32 rounds) per sum, one carry each at most.
// (Carries are NOT rare — a floor point's n_z^2 term alone is ~2^30 — so handing them to LDS atomics made both kernels a quarter
// slower; the signed-overflow rule instead of the bias cost ten instructions per term instead of five and gave half the gain away.)
template <int NACC> struct M3dAcc64 {
    long long v[NACC];
    M3D_HD void clear() {
#pragma unroll
        for (int i = 0; i < NACC; i++) v[i] = 0;
    }
    template <int SLOT> M3D_HD void add(int t) { v[SLOT] += (long long)t; }
    M3D_HD long long wide(int i) const { return v[i]; }
};
template <int NACC> struct M3dAcc32 {
    // v: the running sums of (term + 2^31) — every addend non-negative, so a wrap is the plain unsigned carry — modulo 2^32; cw: carries per sum,
    // an 8-bit field each; n: terms added per sum (the count slot, the last one, is a plain counter: it adds 1 per match)
    unsigned int v[NACC];
    unsigned int cw[(NACC + 3) / 4];
    M3D_HD void clear() {
#pragma unroll
        for (int i = 0; i < NACC; i++) v[i] = 0u;
#pragma unroll
        for (int i = 0; i < (NACC + 3) / 4; i++) cw[i] = 0u;
    }
    template <int SLOT> M3D_HD void add(int t) {
        if constexpr (SLOT == NACC - 1) { v[SLOT] += (unsigned int)t; return; }   // the count
        else {
            const unsigned int a = v[SLOT];
            const unsigned int sum = a + ((unsigned int)t ^ 0x80000000u);
            constexpr unsigned int K = 1u << (8 * (SLOT & 3));
            cw[SLOT >> 2] += (sum < a) ? K : 0u;
            v[SLOT] = sum;
        }
    }
    M3D_HD long long wide(int i) const {
        if (i == NACC - 1) return (long long)v[i];
        const long long carries = (long long)((cw[i >> 2] >> (8 * (i & 3))) & 0xFFu);
        return (long long)v[i] + carries * (1ll << 32) - (long long)v[NACC - 1] * (1ll << 31);
    }
};
