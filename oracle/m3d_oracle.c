/*
 * m3d_oracle.c — CPU restatement (plain C99) of the 6-DoF scan-registration hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it; libm3dreg.so never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference implementation of this path (`gpu_6dslam`) is an empty,
 * un-vendored submodule (/root/reference/.gitmodules:1-3, pinned commit unknown) and the tree holds
 * no CPU ICP, no tests, no golden vectors. There is nothing to compile or
 * diff against, so this file restates the algorithm that BASELINE.json `north_star` names and that
 * DESIGN.md §Spec fixes operation by operation; it is pinned instead by ground truth
 * (tests/test_oracle_*.py: known SE(3) recovery, scipy cKDTree cross-check of the voxel NN).
 *
 * In-tree anchors that the restatement does follow:
 *   - input contract: XYZ float32 at point_step 16, unorganised, produced by
 *     m3d/m3d_aggregator/src/m3d_aggregator.cpp:196-201 (pcl::toPCLPointCloud2 of PointXYZ);
 *   - voxel leaf usage (0.1 m regular grid): m3d/m3d_calibration/src/m3d_calibration_twiddle.cpp:279-286;
 *   - per-point neighbour query loop: m3d_calibration_twiddle.cpp:292-304 (KdTreeFLANN::radiusSearch),
 *     restated here as an exact search over the 27 voxels around the query.
 *
 * Every floating-point expression below is the normative operation order of the spec: build with
 * -ffp-contract=off and without -ffast-math. The HIP kernels implement the same sequence, so all
 * outputs (voxel keys, permutation, NN indices, the 29 fixed-point sums, every pose) are compared
 * BIT-EXACTLY, not to a tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NSUMS 29
#define ORC_MAX_LEVELS 4
#define ORC_INVALID_KEY 0xFFFFFFFFu

enum { ORC_OK = 0, ORC_ERR_INVALID_ARG = -1, ORC_ERR_GRID_TOO_LARGE = -4, ORC_ERR_EMPTY_CLOUD = -5 };
enum { ORC_CONVERGED = 0, ORC_MAX_ITERATIONS = 1, ORC_TOO_FEW_CORR = 2, ORC_RANK_DEFICIENT = 3, ORC_DIVERGED = 4 };
enum { ORC_PT2PT = 0, ORC_PT2PLANE = 1 };

/* Same field layout as m3dreg_params / m3dreg_stats / m3dreg_grid_info in include/m3dreg.h so that
 * the tests can hand one ctypes structure to both sides. */
typedef struct {
    int32_t n_levels;
    float leaf[ORC_MAX_LEVELS];
    int32_t iterations[ORC_MAX_LEVELS];
    float max_corr_dist[ORC_MAX_LEVELS];
    int32_t metric;
    int32_t min_correspondences;
    double eps_rot, eps_trans, pivot_rel_tol;
    float plane_ratio;
    int32_t normal_min_pts;
    float normal_leaf;
    float normal_min_spread;
} orc_params;

typedef struct {
    int32_t status, iterations;
    int64_t n_corr;
    double rms, last_rot, last_trans;
} orc_stats;

typedef struct {
    int32_t n, n_valid, n_cells;
    int32_t dims[3], bits[3];
    float mn[3], mx[3], center[3];
    float leaf, inv_leaf, lbound;
    int32_t has_normals;
} orc_grid_info;

typedef struct {
    orc_grid_info g;
    uint32_t* key;       /* [n] input order */
    uint32_t* skey;      /* [n] sorted */
    int32_t* perm;       /* [n] */
    float* sxyz;         /* [3n] sorted, interleaved */
    uint32_t* cell_key;  /* [n_cells] */
    int32_t* cell_start; /* [n_cells+1] */
    float* nrm;          /* [3n] sorted order, or NULL */
    /* oracle-internal lookup accelerator (open addressing on cell_key); not part of the spec */
    uint32_t hmask;
    int32_t* hslot;
} orc_level;

typedef struct {
    int32_t n, n_levels;
    float* xyz; /* [3n] input order, interleaved (as decoded from the PointCloud2 payload) */
    orc_level lv[ORC_MAX_LEVELS];
    orc_level ng;   /* dedicated normal-estimation grid (leaf = normal_leaf), point-to-plane only */
    float* nrm_in;  /* [3n] normals in input order, (0,0,0) = invalid */
} orc_cloud;

/* ------------------------------------------------------------------------------------------- */
/* Spec §Grid: integer helpers                                                                 */
/* ------------------------------------------------------------------------------------------- */
static int finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

/* smallest b >= 1 with (1 << b) >= d */
static int bits_for(int32_t d) {
    int b = 1;
    while (((int64_t)1 << b) < (int64_t)d) b++;
    return b;
}

/* smallest integer k with 2^k >= x (x > 0, finite), via frexp so it is exact */
static int ceil_log2_d(double x) {
    int ex;
    double m = frexp(x, &ex); /* x = m * 2^ex, m in [0.5,1) */
    return (m == 0.5) ? ex - 1 : ex;
}

/* Spec §Grid: voxel sort key. Voxels are grouped into 2x2x2 buckets: bucket coords c = i >> 1, the
 * buckets are ordered by a compact Morton code (bits of cx, cy, cz interleaved from the LSB up, an axis
 * drops out once its bit width cb[a] is exhausted) and the 8 voxels of a bucket by
 * sub = (ix&1) | (iy&1)<<1 | (iz&1)<<2:   key = morton(cx,cy,cz) << 3 | sub.
 * The key is a bijection of the voxel coordinates, so a run of equal keys is one voxel, every bucket is
 * one contiguous run of up to 8 voxel runs, and the sorted cloud follows a space-filling curve. */
static uint32_t voxel_key(const int32_t cb[3], int32_t ix, int32_t iy, int32_t iz) {
    const uint32_t c[3] = { (uint32_t)ix >> 1, (uint32_t)iy >> 1, (uint32_t)iz >> 1 };
    uint32_t code = 0; int pos = 0;
    for (int b = 0; b < 11; b++)
        for (int a = 0; a < 3; a++)
            if (b < cb[a]) { code |= ((c[a] >> b) & 1u) << pos; pos++; }
    return (code << 3) | ((uint32_t)ix & 1u) | (((uint32_t)iy & 1u) << 1) | (((uint32_t)iz & 1u) << 2);
}

/* Spec §Grid: cell coordinate of a coordinate value along one axis (un-fused sub, mul, floor). */
static float cell_f(float v, float mn, float inv_leaf) {
    float d = v - mn;
    float s = d * inv_leaf;
    return floorf(s);
}

/* ------------------------------------------------------------------------------------------- */
/* a2: PointCloud2 payload decode (little-endian FLOAT32 fields at byte offsets)               */
/* ------------------------------------------------------------------------------------------- */
static void decode_xyz(const uint8_t* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, float* xyz) {
    for (size_t i = 0; i < n; i++) {
        memcpy(&xyz[3 * i + 0], data + i * step + ox, 4);
        memcpy(&xyz[3 * i + 1], data + i * step + oy, 4);
        memcpy(&xyz[3 * i + 2], data + i * step + oz, 4);
    }
}

static uint32_t hash_key(uint32_t k) { return k * 2654435761u; }

static void level_free(orc_level* L) {
    free(L->key); free(L->skey); free(L->perm); free(L->sxyz); free(L->cell_key);
    free(L->cell_start); free(L->nrm); free(L->hslot);
    memset(L, 0, sizeof(*L));
}

/* cell lookup: index into cell_key/cell_start or -1 */
static int32_t find_cell(const orc_level* L, uint32_t key) {
    uint32_t h = hash_key(key) & L->hmask;
    for (;;) {
        int32_t s = L->hslot[h];
        if (s < 0) return -1;
        if (L->cell_key[s] == key) return s;
        h = (h + 1) & L->hmask;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* a3 + a4: voxel keys, stable sort by key, cell table                                          */
/* ------------------------------------------------------------------------------------------- */
static int level_build(orc_level* L, const float* xyz, int32_t n, float leaf) {
    memset(L, 0, sizeof(*L));
    orc_grid_info* g = &L->g;
    g->n = n; g->leaf = leaf; g->inv_leaf = 1.0f / leaf;
    if (!(leaf > 0.0f) || n <= 0) return ORC_ERR_INVALID_ARG;
    /* AABB of finite points (exact: min/max only) */
    int32_t nv = 0;
    for (int32_t i = 0; i < n; i++) {
        const float* p = &xyz[3 * i];
        if (!finite3(p[0], p[1], p[2])) continue;
        if (nv == 0) { for (int a = 0; a < 3; a++) g->mn[a] = g->mx[a] = p[a]; }
        else for (int a = 0; a < 3; a++) { if (p[a] < g->mn[a]) g->mn[a] = p[a]; if (p[a] > g->mx[a]) g->mx[a] = p[a]; }
        nv++;
    }
    g->n_valid = nv;
    if (nv == 0) return ORC_ERR_EMPTY_CLOUD;
    float half_max = 0.0f;
    int total_bits = 0;
    for (int a = 0; a < 3; a++) {
        float fc = cell_f(g->mx[a], g->mn[a], g->inv_leaf);
        if (!(fc < 1073741824.0f)) return ORC_ERR_GRID_TOO_LARGE;
        g->dims[a] = (int32_t)fc + 1;
        g->bits[a] = bits_for((g->dims[a] + 1) >> 1);   /* bit width of the BUCKET coordinate */
        if (g->bits[a] > 11) return ORC_ERR_GRID_TOO_LARGE;
        total_bits += g->bits[a];
        float ext = g->mx[a] - g->mn[a];
        float half = ext * 0.5f;
        g->center[a] = g->mn[a] + half;
        if (half > half_max) half_max = half;
    }
    if (total_bits + 3 > 31) return ORC_ERR_GRID_TOO_LARGE;
    g->lbound = half_max + 3.0f * leaf;

    L->key = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    L->skey = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    L->perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    L->sxyz = (float*)malloc(sizeof(float) * 3 * (size_t)n);
    for (int32_t i = 0; i < n; i++) {
        const float* p = &xyz[3 * i];
        if (!finite3(p[0], p[1], p[2])) { L->key[i] = ORC_INVALID_KEY; continue; }
        int32_t ix = (int32_t)cell_f(p[0], g->mn[0], g->inv_leaf);
        int32_t iy = (int32_t)cell_f(p[1], g->mn[1], g->inv_leaf);
        int32_t iz = (int32_t)cell_f(p[2], g->mn[2], g->inv_leaf);
        L->key[i] = voxel_key(g->bits, ix, iy, iz);
    }
    /* stable LSD radix sort (4 x 8 bit) of (key, index): the spec only says "stable sort by key" */
    {
        uint32_t* ka = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
        int32_t* pa = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
        uint32_t* kb = L->skey; int32_t* pb = L->perm;
        for (int32_t i = 0; i < n; i++) { ka[i] = L->key[i]; pa[i] = i; }
        for (int pass = 0; pass < 4; pass++) {
            size_t cnt[257]; memset(cnt, 0, sizeof(cnt));
            int sh = 8 * pass;
            for (int32_t i = 0; i < n; i++) cnt[((ka[i] >> sh) & 255u) + 1]++;
            for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
            for (int32_t i = 0; i < n; i++) { size_t o = cnt[(ka[i] >> sh) & 255u]++; kb[o] = ka[i]; pb[o] = pa[i]; }
            uint32_t* tk = ka; ka = kb; kb = tk; int32_t* tp = pa; pa = pb; pb = tp;
        }
        /* an even number of passes leaves the result in the malloc'ed pair (ka,pa) */
        memcpy(L->skey, ka, sizeof(uint32_t) * (size_t)n);
        memcpy(L->perm, pa, sizeof(int32_t) * (size_t)n);
        free(ka); free(pa);
    }
    for (int32_t j = 0; j < n; j++) memcpy(&L->sxyz[3 * j], &xyz[3 * L->perm[j]], 12);
    /* cell table */
    int32_t nc = 0;
    for (int32_t j = 0; j < nv; j++) if (j == 0 || L->skey[j] != L->skey[j - 1]) nc++;
    g->n_cells = nc;
    L->cell_key = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nc);
    L->cell_start = (int32_t*)malloc(sizeof(int32_t) * ((size_t)nc + 1));
    nc = 0;
    for (int32_t j = 0; j < nv; j++) if (j == 0 || L->skey[j] != L->skey[j - 1]) { L->cell_key[nc] = L->skey[j]; L->cell_start[nc] = j; nc++; }
    L->cell_start[nc] = nv;
    uint32_t hs = 16; while (hs < 2u * (uint32_t)nc) hs <<= 1;
    L->hmask = hs - 1;
    L->hslot = (int32_t*)malloc(sizeof(int32_t) * hs);
    for (uint32_t i = 0; i < hs; i++) L->hslot[i] = -1;
    for (int32_t c = 0; c < nc; c++) { uint32_t h = hash_key(L->cell_key[c]) & L->hmask; while (L->hslot[h] >= 0) h = (h + 1) & L->hmask; L->hslot[h] = c; }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* a6: exact nearest neighbour inside the 27 voxels around the query                            */
/* returns sorted position of the match or -1; *d2_out = fmaf-chain squared distance            */
/* ------------------------------------------------------------------------------------------- */
static int32_t nn27(const orc_level* L, float ux, float uy, float uz, float dmax2, float* d2_out) {
    const orc_grid_info* g = &L->g;
    if (!finite3(ux, uy, uz)) return -1;
    float fc[3] = { cell_f(ux, g->mn[0], g->inv_leaf), cell_f(uy, g->mn[1], g->inv_leaf), cell_f(uz, g->mn[2], g->inv_leaf) };
    int32_t ic[3];
    for (int a = 0; a < 3; a++) {
        if (!(fc[a] >= -1.0f && fc[a] <= (float)g->dims[a])) return -1;
        ic[a] = (int32_t)fc[a];
    }
    int32_t best = -1; float bd = 0.0f; int32_t bidx = 0;
    for (int dz = -1; dz <= 1; dz++) { int32_t cz = ic[2] + dz; if (cz < 0 || cz >= g->dims[2]) continue;
    for (int dy = -1; dy <= 1; dy++) { int32_t cy = ic[1] + dy; if (cy < 0 || cy >= g->dims[1]) continue;
    for (int dx = -1; dx <= 1; dx++) { int32_t cx = ic[0] + dx; if (cx < 0 || cx >= g->dims[0]) continue;
        uint32_t key = voxel_key(g->bits, cx, cy, cz);
        int32_t c = find_cell(L, key);
        if (c < 0) continue;
        for (int32_t j = L->cell_start[c]; j < L->cell_start[c + 1]; j++) {
            float ex = ux - L->sxyz[3 * j], ey = uy - L->sxyz[3 * j + 1], ez = uz - L->sxyz[3 * j + 2];
            float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            int32_t oi = L->perm[j];
            if (best < 0 || d2 < bd || (d2 == bd && oi < bidx)) { best = j; bd = d2; bidx = oi; }
        }
    }}}
    if (best < 0 || !(bd <= dmax2)) return -1;
    *d2_out = bd;
    return best;
}

/* ------------------------------------------------------------------------------------------- */
/* a9: per-point normals from the 27-voxel neighbourhood (PCA, smallest eigenvector)            */
/* ------------------------------------------------------------------------------------------- */
static void sym3_square(const double m[6], double o[6]) {
    /* m = [m00 m01 m02 m11 m12 m22]; o = m*m (symmetric) */
    o[0] = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
    o[1] = m[0] * m[1] + m[1] * m[3] + m[2] * m[4];
    o[2] = m[0] * m[2] + m[1] * m[4] + m[2] * m[5];
    o[3] = m[1] * m[1] + m[3] * m[3] + m[4] * m[4];
    o[4] = m[1] * m[2] + m[3] * m[4] + m[4] * m[5];
    o[5] = m[2] * m[2] + m[4] * m[4] + m[5] * m[5];
}
static double sym3_maxabs(const double m[6]) {
    double a = 0.0;
    for (int i = 0; i < 6; i++) { double v = fabs(m[i]); if (v > a) a = v; }
    return a;
}

/* Spec §Normals (v2): the max-abs renormalisations scale by a power of two — 2^-e, e the binary exponent of the maximum, which then
 * lies in [1, 2) — instead of dividing by the maximum: exact (no rounding at all), and 42 of the 45 double divisions per voxel gone
 * (they were most of the GPU kernel's time). x: positive, normal. */
static double pow2_recip(double x) {
    uint64_t b; memcpy(&b, &x, 8);
    b = (uint64_t)(2046u - (unsigned)((b >> 52) & 0x7FFu)) << 52;
    double y; memcpy(&y, &b, 8);
    return y;
}

/* Spec §Normals: reciprocal square root from +,-,* only (bit trick seed + 5 Newton steps), so that the
 * CPU and the GPU produce the same bits without relying on either side's sqrt implementation. */
static double det_rsqrt(double x) {
    uint64_t b; memcpy(&b, &x, 8);
    b = 0x5FE6EB50C7B537A9ull - (b >> 1);
    double y; memcpy(&y, &b, 8);
    const double hx = 0.5 * x;
    for (int i = 0; i < 5; i++) y = y * (1.5 - hx * y * y);
    return y;
}

/* normals of every finite point from its 27-voxel neighbourhood in the normal grid L; out: input order */
static void grid_normals(const orc_level* L, float plane_ratio, int32_t min_pts, float min_spread, float* out) {
    const orc_grid_info* g = &L->g;
    const int32_t nv = g->n_valid;
    /* Spec §Normals: positions are quantised to 1/65536 of a voxel INSIDE their voxel
     * (q = rint(frac * 65536), frac = s - floor(s) is exact in float), neighbour voxels are shifted by
     * whole multiples of 65536, and all first/second moments are accumulated as exact int64 — so any
     * evaluation order (this naive per-neighbour loop, or the GPU's per-voxel moments shifted and added)
     * gives the same integers. Units of the covariance below: (leaf/65536)^2. */
    const double spread_q = (double)min_spread * 65536.0;
    const double l2_min_abs = spread_q * spread_q;
    /* every iteration writes only out[3*perm[j]..]: independent, so the OpenMP split changes nothing */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (int32_t j = 0; j < nv; j++) {
        int32_t ic[3];
        for (int a = 0; a < 3; a++) ic[a] = (int32_t)cell_f(L->sxyz[3 * j + a], g->mn[a], g->inv_leaf);
        int64_t s[3] = { 0, 0, 0 }, q[6] = { 0, 0, 0, 0, 0, 0 };
        int64_t k = 0;
        for (int dz = -1; dz <= 1; dz++) { int32_t cz = ic[2] + dz; if (cz < 0 || cz >= g->dims[2]) continue;
        for (int dy = -1; dy <= 1; dy++) { int32_t cy = ic[1] + dy; if (cy < 0 || cy >= g->dims[1]) continue;
        for (int dx = -1; dx <= 1; dx++) { int32_t cx = ic[0] + dx; if (cx < 0 || cx >= g->dims[0]) continue;
            uint32_t ck = voxel_key(g->bits, cx, cy, cz);
            int32_t c = find_cell(L, ck);
            if (c < 0) continue;
            const int dd[3] = { dx, dy, dz };
            for (int32_t t = L->cell_start[c]; t < L->cell_start[c + 1]; t++) {
                int64_t Q[3];
                for (int a = 0; a < 3; a++) {
                    float sv = (L->sxyz[3 * t + a] - g->mn[a]) * g->inv_leaf;
                    float fr = sv - floorf(sv);
                    Q[a] = (int64_t)(int32_t)rintf(fr * 65536.0f) + (int64_t)dd[a] * 65536;
                }
                s[0] += Q[0]; s[1] += Q[1]; s[2] += Q[2];
                q[0] += Q[0] * Q[0]; q[1] += Q[0] * Q[1]; q[2] += Q[0] * Q[2]; q[3] += Q[1] * Q[1]; q[4] += Q[1] * Q[2]; q[5] += Q[2] * Q[2];
                k++;
            }
        }}}
        if (k < (int64_t)min_pts || k < 3) continue;
        double inv = 1.0 / (double)k;
        double m0 = (double)s[0] * inv, m1 = (double)s[1] * inv, m2 = (double)s[2] * inv;
        double c[6] = { (double)q[0] * inv - m0 * m0, (double)q[1] * inv - m0 * m1, (double)q[2] * inv - m0 * m2,
                        (double)q[3] * inv - m1 * m1, (double)q[4] * inv - m1 * m2, (double)q[5] * inv - m2 * m2 };
        double cm = sym3_maxabs(c);
        if (!(cm > 0.0)) continue;
        const double sc = pow2_recip(cm);
        for (int i = 0; i < 6; i++) c[i] = c[i] * sc;
        /* adjugate: eigenvalues l1*l2 (for the eigenvector of l3), l1*l3, l2*l3 */
        double a[6] = { c[3] * c[5] - c[4] * c[4], c[2] * c[4] - c[1] * c[5], c[1] * c[4] - c[2] * c[3],
                        c[0] * c[5] - c[2] * c[2], c[1] * c[2] - c[0] * c[4], c[0] * c[3] - c[1] * c[1] };
        double am = sym3_maxabs(a);
        if (!(am > 1e-12)) continue; /* rank <= 1 neighbourhood (points on a line) */
        double p[6], t2[6];
        { const double sa = pow2_recip(am); for (int i = 0; i < 6; i++) p[i] = a[i] * sa; }
        for (int it = 0; it < 5; it++) { /* p <- p^2, renormalised: adj^(32) */
            sym3_square(p, t2);
            double tm = sym3_maxabs(t2);
            const double st = pow2_recip(tm);
            for (int i = 0; i < 6; i++) p[i] = t2[i] * st;
        }
        /* column with the largest diagonal entry */
        double v[3];
        if (p[0] >= p[3] && p[0] >= p[5]) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }
        else if (p[3] >= p[5]) { v[0] = p[1]; v[1] = p[3]; v[2] = p[4]; }
        else { v[0] = p[2]; v[1] = p[4]; v[2] = p[5]; }
        double nn = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        if (!(nn > 0.0)) continue;
        double rn = det_rsqrt(nn);
        v[0] = v[0] * rn; v[1] = v[1] * rn; v[2] = v[2] * rn;
        /* planarity: l3 = v^T C v, l1*l2 = v^T adj(C) v, l1+l2 = tr - l3; valid iff l2 >= l3/ratio,
           tested on the quadratic x^2 - s x + pr (whose smaller root is l2) without a sqrt */
        double cv0 = c[0] * v[0] + c[1] * v[1] + c[2] * v[2], cv1 = c[1] * v[0] + c[3] * v[1] + c[4] * v[2], cv2 = c[2] * v[0] + c[4] * v[1] + c[5] * v[2];
        double l3 = v[0] * cv0 + v[1] * cv1 + v[2] * cv2;
        double av0 = a[0] * v[0] + a[1] * v[1] + a[2] * v[2], av1 = a[1] * v[0] + a[3] * v[1] + a[4] * v[2], av2 = a[2] * v[0] + a[4] * v[1] + a[5] * v[2];
        double pr = v[0] * av0 + v[1] * av1 + v[2] * av2;
        double sm = (c[0] + c[3] + c[5]) - l3;
        if (l3 < 0.0) l3 = 0.0;
        double mth = l3 / (double)plane_ratio;
        int planar = (mth <= 0.5 * sm) && ((mth * mth - sm * mth) + pr >= 0.0);
        if (!planar) continue;
        /* in-plane extent: l2 (unscaled) >= (min_spread*leaf)^2 rejects single scan-line neighbourhoods */
        double mw = l2_min_abs * sc;   /* the threshold in the units c was scaled to */
        int wide = (mw <= 0.5 * sm) && ((mw * mw - sm * mw) + pr >= 0.0);
        if (!wide) continue;
        /* canonical sign: the component of largest magnitude is positive (first wins ties) */
        int im = 0; if (fabs(v[1]) > fabs(v[im])) im = 1; if (fabs(v[2]) > fabs(v[im])) im = 2;
        if (v[im] < 0.0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
        const int32_t oi = L->perm[j];
        out[3 * oi] = (float)v[0]; out[3 * oi + 1] = (float)v[1]; out[3 * oi + 2] = (float)v[2];
    }
}

/* ------------------------------------------------------------------------------------------- */
/* a7: fixed-point exponents and one linearisation                                              */
/* ------------------------------------------------------------------------------------------- */
/* exps: [rr, rt, tt, gr, gt, ss]; quantised term = (int32) rintf(term * 2^exp) */
static void fixed_exps(const orc_grid_info* g, float max_corr_dist, int32_t e[6]) {
    double lb = (double)g->lbound, D = (double)max_corr_dist * 1.001;
    e[0] = 30 - ceil_log2_d(3.0 * lb * lb);
    e[1] = 30 - ceil_log2_d(1.7320508075688772 * lb);
    e[2] = 30;
    e[3] = 30 - ceil_log2_d(1.7320508075688772 * lb * D);
    e[4] = 30 - ceil_log2_d(D);
    e[5] = 30 - ceil_log2_d(D * D);
}

static int64_t quant(float term, float scale) { return (int64_t)(int32_t)rintf(term * scale); }

/* slot index of H(k,l), k <= l, in the 21-entry upper triangle (row-major) */
static int hslot21(int k, int l) { return k * 6 - (k * (k - 1)) / 2 + (l - k); }

static void pose_to_float(const double T[16], float R[9], float t[3]) {
    /* T column-major: T[c*4+r]; R row-major here */
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = (float)T[c * 4 + r]; t[r] = (float)T[12 + r]; }
}

static void accumulate(const float* src_xyz, int32_t n_src, const orc_level* L, const double T[16], int metric,
                       float max_corr_dist, int64_t sums[ORC_NSUMS], int32_t exps[6], int32_t* nn_out, float* d2_out) {
    const orc_grid_info* g = &L->g;
    float R[9], t[3];
    pose_to_float(T, R, t);
    fixed_exps(g, max_corr_dist, exps);
    float S[6];
    for (int i = 0; i < 6; i++) S[i] = ldexpf(1.0f, exps[i]);
    const float dmax2 = max_corr_dist * max_corr_dist;
    memset(sums, 0, sizeof(int64_t) * ORC_NSUMS);
    /* integer sums are associative, so the OpenMP split (cpu_baseline leg of bench.py) cannot change a bit */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : sums[:ORC_NSUMS])
#endif
    for (int32_t i = 0; i < n_src; i++) {
        const float px = src_xyz[3 * i], py = src_xyz[3 * i + 1], pz = src_xyz[3 * i + 2];
        if (nn_out) nn_out[i] = -1;
        if (d2_out) d2_out[i] = 0.0f;
        if (!finite3(px, py, pz)) continue;
        /* a5: u = R p + t, explicit fma chain */
        float ux = fmaf(R[0], px, fmaf(R[1], py, fmaf(R[2], pz, t[0])));
        float uy = fmaf(R[3], px, fmaf(R[4], py, fmaf(R[5], pz, t[1])));
        float uz = fmaf(R[6], px, fmaf(R[7], py, fmaf(R[8], pz, t[2])));
        float d2;
        int32_t j = nn27(L, ux, uy, uz, dmax2, &d2);
        if (j < 0) continue;
        if (nn_out) nn_out[i] = L->perm[j];
        if (d2_out) d2_out[i] = d2;
        const float ex = ux - L->sxyz[3 * j], ey = uy - L->sxyz[3 * j + 1], ez = uz - L->sxyz[3 * j + 2];
        const float wx = ux - g->center[0], wy = uy - g->center[1], wz = uz - g->center[2];
        float J[6], Hrr[6], Hrt[9], gg[6], ssr;
        if (metric == ORC_PT2PLANE) {
            const float nx = L->nrm[3 * j], ny = L->nrm[3 * j + 1], nz = L->nrm[3 * j + 2];
            if (nx == 0.0f && ny == 0.0f && nz == 0.0f) continue; /* no usable normal: match rejected */
            J[0] = wy * nz - wz * ny; J[1] = wz * nx - wx * nz; J[2] = wx * ny - wy * nx;
            J[3] = nx; J[4] = ny; J[5] = nz;
            const float r = nx * ex + ny * ey + nz * ez;
            for (int k = 0; k < 6; k++) for (int l = k; l < 6; l++) {
                int cls = (l < 3) ? 0 : (k < 3 ? 1 : 2);
                sums[hslot21(k, l)] += quant(J[k] * J[l], S[cls]);
            }
            for (int k = 0; k < 3; k++) sums[21 + k] += quant(J[k] * r, S[3]);
            for (int k = 3; k < 6; k++) sums[21 + k] += quant(J[k] * r, S[4]);
            sums[27] += quant(r * r, S[5]);
        } else {
            Hrr[0] = wy * wy + wz * wz; Hrr[1] = -(wx * wy); Hrr[2] = -(wx * wz);
            Hrr[3] = wx * wx + wz * wz; Hrr[4] = -(wy * wz); Hrr[5] = wx * wx + wy * wy;
            Hrt[0] = 0.0f; Hrt[1] = -wz; Hrt[2] = wy; Hrt[3] = wz; Hrt[4] = 0.0f; Hrt[5] = -wx; Hrt[6] = -wy; Hrt[7] = wx; Hrt[8] = 0.0f;
            gg[0] = wy * ez - wz * ey; gg[1] = wz * ex - wx * ez; gg[2] = wx * ey - wy * ex;
            gg[3] = ex; gg[4] = ey; gg[5] = ez;
            ssr = d2;
            int q = 0;
            for (int k = 0; k < 3; k++) for (int l = k; l < 3; l++) sums[hslot21(k, l)] += quant(Hrr[q++], S[0]);
            for (int k = 0; k < 3; k++) for (int m = 0; m < 3; m++) sums[hslot21(k, 3 + m)] += quant(Hrt[3 * k + m], S[1]);
            for (int k = 3; k < 6; k++) for (int l = k; l < 6; l++) sums[hslot21(k, l)] += quant(k == l ? 1.0f : 0.0f, S[2]);
            for (int k = 0; k < 3; k++) sums[21 + k] += quant(gg[k], S[3]);
            for (int k = 3; k < 6; k++) sums[21 + k] += quant(gg[k], S[4]);
            sums[27] += quant(ssr, S[5]);
        }
        sums[28] += 1;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* a8: 6x6 LDL^T solve and SE(3) update about the centre c (all double, fixed operation order)  */
/* ------------------------------------------------------------------------------------------- */
static int solve_update(const int64_t sums[ORC_NSUMS], const int32_t exps[6], const float center[3], double pivot_rel_tol,
                        double T[16], double* th2_out, double* tr2_out) {
    double A[6][6], b[6];
    for (int k = 0; k < 6; k++) for (int l = k; l < 6; l++) {
        int cls = (l < 3) ? 0 : (k < 3 ? 1 : 2);
        double v = ldexp((double)sums[hslot21(k, l)], -exps[cls]);
        A[k][l] = v; A[l][k] = v;
    }
    for (int k = 0; k < 6; k++) b[k] = -ldexp((double)sums[21 + k], -exps[k < 3 ? 3 : 4]);
    double dmax = 0.0;
    for (int k = 0; k < 6; k++) if (A[k][k] > dmax) dmax = A[k][k];
    const double tol = pivot_rel_tol * dmax;
    double Lm[6][6], D[6], Dinv[6];   /* spec v2: ONE division per pivot (its reciprocal), every other quotient is a product with it */
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
        for (int k = 0; k < j; k++) d = d - Lm[j][k] * Lm[j][k] * D[k];
        if (!(d > tol)) return ORC_RANK_DEFICIENT;
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
    *th2_out = th2; *tr2_out = tr2;
    if (!(th2 <= 4.0) || !(tr2 < 1e300)) return ORC_DIVERGED;
    /* nested series in th2: A = sin(th)/th, B = (1-cos th)/th^2, C = (th - sin th)/th^3 */
    double sa = 1.0, sb = 1.0, sc = 1.0;
    /* spec v2: the constant divisors are multiplied in as their (correctly rounded) reciprocals — 59 double divisions per solve were
     * 4 of the 5 us this step takes on one GPU lane */
    for (int k = 12; k >= 1; k--) {
        sa = 1.0 - th2 * sa * (1.0 / (double)((2 * k) * (2 * k + 1)));
        sb = 1.0 - th2 * sb * (1.0 / (double)((2 * k + 1) * (2 * k + 2)));
        sc = 1.0 - th2 * sc * (1.0 / (double)((2 * k + 2) * (2 * k + 3)));
    }
    const double Ac = sa, Bc = sb * 0.5, Cc = sc * (1.0 / 6.0);
    /* W = [w]x, W2 = W*W */
    const double W[9] = { 0, -w2, w1, w2, 0, -w0, -w1, w0, 0 };
    const double W2[9] = { -(w1 * w1 + w2 * w2), w0 * w1, w0 * w2, w0 * w1, -(w0 * w0 + w2 * w2), w1 * w2, w0 * w2, w1 * w2, -(w0 * w0 + w1 * w1) };
    double Re[9], Ve[9];
    for (int i = 0; i < 9; i++) {
        double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        Re[i] = id + Ac * W[i] + Bc * W2[i];
        Ve[i] = id + Bc * W[i] + Cc * W2[i];
    }
    const double te[3] = { Ve[0] * v0 + Ve[1] * v1 + Ve[2] * v2, Ve[3] * v0 + Ve[4] * v1 + Ve[5] * v2, Ve[6] * v0 + Ve[7] * v1 + Ve[8] * v2 };
    const double c[3] = { (double)center[0], (double)center[1], (double)center[2] };
    /* T <- Tinc * T, Tinc(x) = Re (x - c) + c + te */
    double Rn[9], tn[3];
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++)
        Rn[3 * r + cc] = Re[3 * r] * T[cc * 4 + 0] + Re[3 * r + 1] * T[cc * 4 + 1] + Re[3 * r + 2] * T[cc * 4 + 2];
    const double d0 = T[12] - c[0], d1 = T[13] - c[1], d2 = T[14] - c[2];
    for (int r = 0; r < 3; r++) tn[r] = (Re[3 * r] * d0 + Re[3 * r + 1] * d1 + Re[3 * r + 2] * d2) + c[r] + te[r];
    for (int r = 0; r < 3; r++) { for (int cc = 0; cc < 3; cc++) T[cc * 4 + r] = Rn[3 * r + cc]; T[12 + r] = tn[r]; }
    T[3] = 0.0; T[7] = 0.0; T[11] = 0.0; T[15] = 1.0;
    return -1; /* "keep iterating" */
}

/* ------------------------------------------------------------------------------------------- */
/* exported C entry points (ctypes)                                                             */
/* ------------------------------------------------------------------------------------------- */
#ifdef _OPENMP
#include <omp.h>
#endif
/* thread count of the OpenMP build (bench.py cpu_baseline); a no-op in the serial build */
int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

int orc_default_params(orc_params* p) {
    if (!p) return ORC_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->n_levels = 2; /* == m3dreg_default_params (tests/test_abi.py): coarse to fine */
    p->leaf[0] = 0.4f; p->iterations[0] = 20; p->max_corr_dist[0] = 1.5f;
    p->leaf[1] = 0.1f; p->iterations[1] = 20; p->max_corr_dist[1] = 0.5f;
    p->metric = ORC_PT2PLANE; p->min_correspondences = 10;
    p->eps_rot = 1e-5; p->eps_trans = 1e-5; p->pivot_rel_tol = 1e-9;
    p->plane_ratio = 0.25f; p->normal_min_pts = 5; p->normal_leaf = 0.4f; p->normal_min_spread = 0.25f;
    return ORC_OK;
}

void orc_cloud_destroy(orc_cloud* c) {
    if (!c) return;
    for (int l = 0; l < ORC_MAX_LEVELS; l++) level_free(&c->lv[l]);
    level_free(&c->ng);
    free(c->nrm_in); free(c->xyz); free(c);
}

/* source_only: what m3dreg_cloud_desc.source_only asks of the HIP side — the cloud is sorted along every level's curve (its points
 * are the queries of a registration, streamed in that order) but gets no normal-estimation grid and no normals; it cannot be a
 * target of a point-to-plane registration. bench.py's cpu_baseline uses it so that both legs do the same work. */
static int cloud_create(const orc_params* p, const void* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, int source_only, orc_cloud** out) {
    if (!p || !data || !out || n == 0 || n > 0x7FFFFFFFu || p->n_levels < 1 || p->n_levels > ORC_MAX_LEVELS) return ORC_ERR_INVALID_ARG;
    if (ox + 4 > step || oy + 4 > step || oz + 4 > step) return ORC_ERR_INVALID_ARG;
    orc_cloud* c = (orc_cloud*)calloc(1, sizeof(orc_cloud));
    c->n = (int32_t)n; c->n_levels = p->n_levels;
    c->xyz = (float*)malloc(sizeof(float) * 3 * n);
    decode_xyz((const uint8_t*)data, n, step, ox, oy, oz, c->xyz);
    if (p->metric == ORC_PT2PLANE && !source_only) {
        int rc = level_build(&c->ng, c->xyz, c->n, p->normal_leaf);
        if (rc != ORC_OK) { orc_cloud_destroy(c); return rc; }
        c->nrm_in = (float*)calloc(3 * n, sizeof(float));
        grid_normals(&c->ng, p->plane_ratio, p->normal_min_pts, p->normal_min_spread, c->nrm_in);
    }
    for (int l = 0; l < p->n_levels; l++) {
        int rc = level_build(&c->lv[l], c->xyz, c->n, p->leaf[l]);
        if (rc != ORC_OK) { orc_cloud_destroy(c); return rc; }
        if (p->metric == ORC_PT2PLANE && !source_only) { /* per-level copy of the normals in that level's sorted order */
            orc_level* L = &c->lv[l];
            L->nrm = (float*)malloc(sizeof(float) * 3 * n);
            for (size_t j = 0; j < n; j++) memcpy(&L->nrm[3 * j], &c->nrm_in[3 * (size_t)L->perm[j]], 12);
            L->g.has_normals = 1;
        }
    }
    *out = c;
    return ORC_OK;
}
int orc_cloud_create(const orc_params* p, const void* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, orc_cloud** out) {
    return cloud_create(p, data, n, step, ox, oy, oz, 0, out);
}
int orc_cloud_create_source(const orc_params* p, const void* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, orc_cloud** out) {
    return cloud_create(p, data, n, step, ox, oy, oz, 1, out);
}

int orc_cloud_grid_info(const orc_cloud* c, int level, orc_grid_info* out) {
    if (!c || !out || level < 0 || level >= c->n_levels) return ORC_ERR_INVALID_ARG;
    *out = c->lv[level].g;
    return ORC_OK;
}

int orc_cloud_export(const orc_cloud* c, int level, uint32_t* keys, uint32_t* skeys, int32_t* perm, float* sxyz, float* nrm,
                     uint32_t* cell_key, int32_t* cell_start) {
    if (!c || level < 0 || level >= c->n_levels) return ORC_ERR_INVALID_ARG;
    const orc_level* L = &c->lv[level];
    size_t n = (size_t)c->n;
    if (keys) memcpy(keys, L->key, 4 * n);
    if (skeys) memcpy(skeys, L->skey, 4 * n);
    if (perm) memcpy(perm, L->perm, 4 * n);
    if (sxyz) memcpy(sxyz, L->sxyz, 12 * n);
    if (nrm) { if (!L->nrm) return ORC_ERR_INVALID_ARG; memcpy(nrm, L->nrm, 12 * n); }
    if (cell_key) memcpy(cell_key, L->cell_key, 4 * (size_t)L->g.n_cells);
    if (cell_start) memcpy(cell_start, L->cell_start, 4 * ((size_t)L->g.n_cells + 1));
    return ORC_OK;
}

int orc_debug_nn(const orc_cloud* tgt, int level, const float* q, size_t nq, float max_corr_dist, int32_t* out_idx, float* out_d2) {
    if (!tgt || level < 0 || level >= tgt->n_levels || !q) return ORC_ERR_INVALID_ARG;
    const orc_level* L = &tgt->lv[level];
    const float dmax2 = max_corr_dist * max_corr_dist;
    for (size_t i = 0; i < nq; i++) {
        float d2 = 0.0f;
        int32_t j = nn27(L, q[3 * i], q[3 * i + 1], q[3 * i + 2], dmax2, &d2);
        if (out_idx) out_idx[i] = j < 0 ? -1 : L->perm[j];
        if (out_d2) out_d2[i] = j < 0 ? 0.0f : d2;
    }
    return ORC_OK;
}

static void T_from_float(const float Tf[16], double T[16]) { for (int i = 0; i < 16; i++) T[i] = (double)Tf[i]; }

int orc_debug_accumulate(const orc_params* p, const orc_cloud* src, const orc_cloud* tgt, int level, const float Tf[16],
                         int64_t sums[ORC_NSUMS], int32_t exps[6], int32_t* nn_out, float* d2_out) {
    if (!p || !src || !tgt || level < 0 || level >= tgt->n_levels) return ORC_ERR_INVALID_ARG;
    double T[16];
    T_from_float(Tf, T);
    accumulate(src->xyz, src->n, &tgt->lv[level], T, p->metric, p->max_corr_dist[level], sums, exps, nn_out, d2_out);
    return ORC_OK;
}

/* Full registration. trace (optional): column-major double[16] per executed iteration. */
int orc_align_clouds(const orc_params* p, const orc_cloud* src, const orc_cloud* tgt, const float init_T[16], float out_T[16],
                     orc_stats* st, double* trace, size_t trace_cap, size_t* trace_n) {
    if (!p || !src || !tgt || !init_T || !out_T || tgt->n_levels != p->n_levels) return ORC_ERR_INVALID_ARG;
    double T[16];
    T_from_float(init_T, T);
    orc_stats s; memset(&s, 0, sizeof(s));
    s.status = ORC_MAX_ITERATIONS;
    size_t tn = 0;
    int stop = 0;
    for (int l = 0; l < p->n_levels && !stop; l++) {
        const orc_level* L = &tgt->lv[l];
        for (int it = 0; it < p->iterations[l]; it++) {
            int64_t sums[ORC_NSUMS]; int32_t exps[6];
            double th2 = 0.0, tr2 = 0.0;
            accumulate(src->xyz, src->n, L, T, p->metric, p->max_corr_dist[l], sums, exps, NULL, NULL);
            s.iterations++;
            s.n_corr = sums[28];
            s.rms = sums[28] > 0 ? sqrt(ldexp((double)sums[27], -exps[5]) / (double)sums[28]) : 0.0;
            if (sums[28] < (int64_t)p->min_correspondences) { s.status = ORC_TOO_FEW_CORR; stop = 1; }
            else {
                int rc = solve_update(sums, exps, L->g.center, p->pivot_rel_tol, T, &th2, &tr2);
                s.last_rot = sqrt(th2); s.last_trans = sqrt(tr2);
                if (rc >= 0) { s.status = rc; stop = 1; }
            }
            if (trace && tn < trace_cap) memcpy(&trace[16 * tn], T, sizeof(double) * 16);
            tn++;
            if (stop) break;
            if (th2 < p->eps_rot * p->eps_rot && tr2 < p->eps_trans * p->eps_trans) {
                if (l == p->n_levels - 1) { s.status = ORC_CONVERGED; stop = 1; }
                break; /* level converged: move to the next finer level */
            }
        }
    }
    for (int i = 0; i < 16; i++) out_T[i] = (float)T[i];
    if (st) *st = s;
    if (trace_n) *trace_n = tn;
    return ORC_OK;
}
