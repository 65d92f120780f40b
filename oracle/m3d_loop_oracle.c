/* m3d_loop_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into libm3dreg.so).
 * CPU restatement of the loop-closure candidate generation of SURVEY.md §8 row f4, second half (csrc/loop.hip, m3dloop_* in
 * include/m3dreg.h). No reference source exists for this step — the reference launches `gpu_6dslam_node`
 * (/root/reference/m3d/m3d_husky_launch/launch/m3d_husky_bringup.launch:13) from an empty, un-vendored submodule
 * (/root/reference/.gitmodules:1-3): PARITY UNPINNED. The spec:
 *   keyframe k = (pose T_k column-major float[16], cloud), numbered in insertion order;
 *   signature: for every point p of the cloud with finite coordinates: u = R p + t as the fma chain of spec row a5 (R, t = T rounded to
 *     float); skipped when u is not finite; v_a = floorf(u_a * (1.0f / sig_leaf)), skipped unless |v_a| < 2^20 - 1 on every axis;
 *     key = (v_x + 2^20) << 42 | (v_y + 2^20) << 21 | (v_z + 2^20); bit = (key * 0x9E3779B97F4A7C15) >> (64 - sig_log2_bits); the bit is set;
 *   pair (i, j) is examined iff j <= i - min_gap and d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) <= radius * radius (float, d = t_i - t_j);
 *     overlap = popcount(sig_i & sig_j); it qualifies iff overlap * 65536 >= lrintf(min_overlap * 65536) * min(pop_i, pop_j);
 *   candidates of i: the top_k qualifying j by overlap, ties towards the smaller j;
 *   init_T = inv(T_j) * T_i in double, the sums left to right, rounded to float. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LOOP_OFF 1048576

typedef struct {
    float sig_leaf; int32_t sig_log2_bits; float radius; int32_t min_gap; int32_t top_k; float min_overlap; int32_t max_keyframes; int32_t reserved;
} orc_loop_params;   /* == m3dloop_params */
typedef struct {
    int32_t source, target; uint32_t overlap, pop_source, pop_target; float dist2; float init_T[16];
} orc_loop_candidate;   /* == m3dloop_candidate */

typedef struct {
    orc_loop_params P; int W; int n;
    uint32_t* sig;    /* [max_keyframes][W] */
    uint32_t* pop;    /* [max_keyframes] */
    float* T;         /* [max_keyframes][16] */
} orc_loop;

orc_loop* orc_loop_create(const orc_loop_params* P) {
    orc_loop* l = (orc_loop*)calloc(1, sizeof(orc_loop));
    l->P = *P; l->W = 1 << (P->sig_log2_bits - 5);
    l->sig = (uint32_t*)calloc((size_t)P->max_keyframes * (size_t)l->W, sizeof(uint32_t));
    l->pop = (uint32_t*)calloc((size_t)P->max_keyframes, sizeof(uint32_t));
    l->T = (float*)calloc((size_t)P->max_keyframes * 16, sizeof(float));
    return l;
}
void orc_loop_destroy(orc_loop* l) { if (l) { free(l->sig); free(l->pop); free(l->T); free(l); } }
int orc_loop_size(const orc_loop* l) { return l->n; }

static uint32_t popc32(uint32_t v) { uint32_t c = 0; while (v) { v &= v - 1; c++; } return c; }

static void sign_into(orc_loop* l, int k, const float* xyz, size_t n, const float T[16]) {
    uint32_t* s = l->sig + (size_t)k * (size_t)l->W;
    memset(s, 0, sizeof(uint32_t) * (size_t)l->W);
    float R[9], t[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = T[c * 4 + r]; t[r] = T[12 + r]; }
    const float inv = 1.0f / l->P.sig_leaf;
    const float lim = (float)(LOOP_OFF - 1);
    for (size_t i = 0; i < n; i++) {
        const float* p = xyz + 3 * i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        float u[3];
        for (int r = 0; r < 3; r++) u[r] = fmaf(R[3 * r], p[0], fmaf(R[3 * r + 1], p[1], fmaf(R[3 * r + 2], p[2], t[r])));
        if (!(isfinite(u[0]) && isfinite(u[1]) && isfinite(u[2]))) continue;
        const float fx = floorf(u[0] * inv), fy = floorf(u[1] * inv), fz = floorf(u[2] * inv);
        if (!(fx > -lim && fx < lim && fy > -lim && fy < lim && fz > -lim && fz < lim)) continue;
        const uint64_t key = ((uint64_t)(uint32_t)((int)fx + LOOP_OFF) << 42) | ((uint64_t)(uint32_t)((int)fy + LOOP_OFF) << 21) | (uint64_t)(uint32_t)((int)fz + LOOP_OFF);
        const uint32_t bit = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - l->P.sig_log2_bits));
        s[bit >> 5] |= 1u << (bit & 31u);
    }
    uint32_t c = 0;
    for (int w = 0; w < l->W; w++) c += popc32(s[w]);
    l->pop[k] = c;
    memcpy(l->T + 16 * (size_t)k, T, sizeof(float) * 16);
}

/* xyz: n points, 3 floats each. Returns the keyframe's index, or -1 when the database is full. */
int orc_loop_add(orc_loop* l, const float* xyz, size_t n, const float T[16]) {
    if (l->n >= l->P.max_keyframes) return -1;
    sign_into(l, l->n, xyz, n, T);
    return l->n++;
}
void orc_loop_update(orc_loop* l, int k, const float* xyz, size_t n, const float T[16]) { sign_into(l, k, xyz, n, T); }
void orc_loop_signature(const orc_loop* l, int k, uint32_t* words, uint32_t* pop) {
    if (words) memcpy(words, l->sig + (size_t)k * (size_t)l->W, sizeof(uint32_t) * (size_t)l->W);
    if (pop) *pop = l->pop[k];
}

void orc_loop_rel(const float* Tj, const float* Ti, float out[16]) {
    double Rj[3][3], Ri[3][3], d[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { Rj[r][c] = (double)Tj[c * 4 + r]; Ri[r][c] = (double)Ti[c * 4 + r]; } d[r] = (double)Ti[12 + r] - (double)Tj[12 + r]; }
    for (int k = 0; k < 16; k++) out[k] = 0.0f;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) out[c * 4 + r] = (float)((Rj[0][r] * Ri[0][c] + Rj[1][r] * Ri[1][c]) + Rj[2][r] * Ri[2][c]);
        out[12 + r] = (float)((Rj[0][r] * d[0] + Rj[1][r] * d[1]) + Rj[2][r] * d[2]);
    }
    out[15] = 1.0f;
}

/* rows first .. first + count - 1 (count < 0: to the newest); at most cap written; returns how many exist */
size_t orc_loop_candidates(const orc_loop* l, int first, int count, orc_loop_candidate* out, size_t cap) {
    const int n = l->n;
    const int last = (count < 0 || first + count > n) ? n : first + count;
    const uint32_t thr = (uint32_t)lrintf(l->P.min_overlap * 65536.0f);
    const float r2 = l->P.radius * l->P.radius;
    size_t found = 0;
    for (int i = first; i < last; i++) {
        const uint32_t* si = l->sig + (size_t)i * (size_t)l->W;
        const float* ti = l->T + 16 * (size_t)i + 12;
        uint64_t below = ~0ull;
        for (int k = 0; k < l->P.top_k; k++) {
            uint64_t best = 0; float best_d2 = 0.f;
            for (int j = 0; j <= i - l->P.min_gap; j++) {
                const float* tj = l->T + 16 * (size_t)j + 12;
                const float dx = ti[0] - tj[0], dy = ti[1] - tj[1], dz = ti[2] - tj[2];
                const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (!(d2 <= r2)) continue;
                const uint32_t* sj = l->sig + (size_t)j * (size_t)l->W;
                uint32_t ov = 0;
                for (int w = 0; w < l->W; w++) ov += popc32(si[w] & sj[w]);
                const uint32_t pm = l->pop[i] < l->pop[j] ? l->pop[i] : l->pop[j];
                if (((uint64_t)ov << 16) < (uint64_t)thr * (uint64_t)pm) continue;
                const uint64_t key = ((uint64_t)ov << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)j);
                if (key < below && key > best) { best = key; best_d2 = d2; }
            }
            if (!best) break;
            below = best;
            if (found < cap) {
                orc_loop_candidate* c = &out[found];
                const int j = (int)(0xFFFFFFFFu - (uint32_t)best);
                c->source = i; c->target = j; c->overlap = (uint32_t)(best >> 32);
                c->pop_source = l->pop[i]; c->pop_target = l->pop[j]; c->dist2 = best_d2;
                orc_loop_rel(l->T + 16 * (size_t)j, l->T + 16 * (size_t)i, c->init_T);
            }
            found++;
        }
    }
    return found;
}
