// Test infrastructure (tests/test_acc32.py): the reduction kernels' 32-bit running sums (mandala_mapping_amd/csrc/m3d_acc.h, compiled here with g++)
// against plain 64-bit sums, for random and adversarial int32 terms. Exit code 0 = every sum identical.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include "m3d_acc.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

template <int NACC, int... I>
static void add_all(M3dAcc32<NACC>& a, M3dAcc64<NACC>& b, const int* t, std::integer_sequence<int, I...>) {
    ((a.template add<I>(t[I]), b.template add<I>(t[I])), ...);
}

template <int NACC>
static int run(int n_terms, int mode) {
    M3dAcc32<NACC> a; M3dAcc64<NACC> b;
    a.clear(); b.clear();
    int t[NACC];
    for (int k = 0; k < n_terms; k++) {
        for (int i = 0; i < NACC - 1; i++) {
            switch (mode) {
                case 0: t[i] = (int)rnd(); break;                                         // anything
                case 1: t[i] = INT32_MAX; break;                                          // every term wraps upwards
                case 2: t[i] = INT32_MIN; break;                                          // ... downwards
                case 3: t[i] = (k & 1) ? INT32_MIN : INT32_MAX; break;                    // alternating
                case 4: t[i] = (int)(rnd() & 0x3FFFFFFFu) + 0x30000000; break;            // ~2^30 and positive: a floor point's n_z^2
                case 5: t[i] = (i & 1) ? -(int)(rnd() >> 1) : (int)(rnd() >> 1); break;   // a sign per slot
                default: t[i] = (int)(rnd() % 7u) - 3; break;                             // tiny
            }
        }
        t[NACC - 1] = 1;   // the count slot
        add_all<NACC>(a, b, t, std::make_integer_sequence<int, NACC>{});
    }
    for (int i = 0; i < NACC; i++)
        if (a.wide(i) != b.wide(i)) { std::printf("NACC %d mode %d terms %d slot %d: %lld != %lld\n", NACC, mode, n_terms, i, a.wide(i), b.wide(i)); return 1; }
    return 0;
}

int main() {
    int bad = 0, cases = 0;
    for (int mode = 0; mode < 7; mode++)
        for (int n = 0; n <= 136; n++) { bad += run<29>(n, mode); bad += run<17>(n, mode); cases += 2; }
    for (int rep = 0; rep < 2000; rep++) { bad += run<29>((int)(rnd() % 129u), 0); cases++; }
    for (int n : { 200, 255 }) { bad += run<29>(n, 1); bad += run<29>(n, 2); cases += 2; }   // the 8-bit carry fields hold 255
    std::printf("%d cases, %d mismatches\n", cases, bad);
    return bad ? 1 : 0;
}
