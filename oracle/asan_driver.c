/*
 * TEST INFRASTRUCTURE (like everything under oracle/): a driver that runs every entry point of the CPU oracle and of the k-d tree
 * baseline on small synthetic inputs under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan && oracle/asan_driver`;
 * tests/test_oracle_sanitized.py does exactly that). GPU sanitizers are not available on the MI355X pool, so the oracle — the checker
 * every parity claim rests on — is the part that gets them. Exits 0 and prints "asan_driver: ok" when every call returned what it should.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- the oracle's C interface (oracle/m3d_oracle.c, m3d_agg_oracle.c, m3d_cal_oracle.c, m3d_map_oracle.c, m3d_loop_oracle.c, m3d_kdtree_icp.c) ---- */
#define ORC_MAX_LEVELS 4
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
typedef struct { int32_t status, iterations; int64_t n_corr; double rms, last_rot, last_trans; } orc_stats;
typedef struct {
    int32_t n, n_valid, n_cells;
    int32_t dims[3], bits[3];
    float mn[3], mx[3], center[3];
    float leaf, inv_leaf, lbound;
    int32_t has_normals;
} orc_grid_info;
typedef struct orc_cloud orc_cloud;
typedef struct orc_agg orc_agg;
typedef struct orc_map orc_map;
typedef struct orc_loop orc_loop;
typedef struct { float sig_leaf; int32_t sig_log2_bits; float radius; int32_t min_gap; int32_t top_k; float min_overlap; int32_t max_keyframes; int32_t reserved; } orc_loop_params;
typedef struct { int32_t source, target; uint32_t overlap, pop_source, pop_target; float dist2; float init_T[16]; } orc_loop_candidate;
orc_loop* orc_loop_create(const orc_loop_params* P);
void orc_loop_destroy(orc_loop* l);
int orc_loop_size(const orc_loop* l);
int orc_loop_add(orc_loop* l, const float* xyz, size_t n, const float T[16]);
void orc_loop_update(orc_loop* l, int k, const float* xyz, size_t n, const float T[16]);
void orc_loop_signature(const orc_loop* l, int k, uint32_t* words, uint32_t* pop);
size_t orc_loop_candidates(const orc_loop* l, int first, int count, orc_loop_candidate* out, size_t cap);
void orc_sincosf(float x, float* s, float* c);
int orc_default_params(orc_params* p);
int orc_set_threads(int n);
int orc_cloud_create(const orc_params* p, const void* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, orc_cloud** out);
int orc_cloud_create_source(const orc_params* p, const void* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, orc_cloud** out);
void orc_cloud_destroy(orc_cloud* c);
int orc_cloud_grid_info(const orc_cloud* c, int level, orc_grid_info* out);
int orc_debug_nn(const orc_cloud* tgt, int level, const float* q, size_t nq, float max_corr_dist, int32_t* out_idx, float* out_d2);
int orc_align_clouds(const orc_params* p, const orc_cloud* src, const orc_cloud* tgt, const float init_T[16], float out_T[16], orc_stats* st,
                     double* trace, size_t trace_cap, size_t* trace_n);
orc_agg* orc_agg_create(const double bb[6]);
void orc_agg_destroy(orc_agg* a);
void orc_agg_restart(orc_agg* a);
void orc_agg_add_cloud(orc_agg* a, const uint8_t* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, const double tf[7]);
void orc_agg_add_scan(orc_agg* a, const float* ranges, size_t n, float angle_min, float angle_increment, const double tf[7]);
size_t orc_agg_count(const orc_agg* a);
void orc_agg_points(const orc_agg* a, float* out);
double orc_agg_angle(const orc_agg* a);
double orc_agg_progress(const orc_agg* a);
int orc_agg_ready(const orc_agg* a);
void orc_cal_offset_matrix(const float p[6], float out_l[9], float out_t[3]);
void orc_cal_compose(const float al[9], const float at[3], const float bl[9], const float bt[3], float ol[9], float ot[3]);
int64_t orc_cal_test_data(const float* seg_xyz, const int32_t* seg_n, const float* seg_T, int32_t n_seg, int32_t laser_up_axis, const float params[6],
                          int64_t out_sizes[4]);
int64_t orc_cal_test_data_bruteforce(const float* seg_xyz, const int32_t* seg_n, const float* seg_T, int32_t n_seg, int32_t laser_up_axis,
                                     const float params[6]);
orc_map* orc_map_create(float leaf, size_t cap);
void orc_map_destroy(orc_map* m);
size_t orc_map_size(const orc_map* m);
void orc_map_points(const orc_map* m, float* out);
size_t orc_map_insert(orc_map* m, const float* xyz, size_t n, const float T[16]);
long long kdicp_align(const float* src, int n_src, const float* tgt, int n_tgt, int metric, float max_corr_dist, int iterations, int normal_k,
                      int threads, double T[16], double ms[3]);

/* ---- a room seen from two poses: walls, floor, a box; deterministic ---- */
static uint32_t lcg_state = 12345u;
static float frand(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) * (1.0f / 16777216.0f); }

static size_t make_room(float* xyz, size_t cap, float yaw, float tx, float ty) {
    size_t n = 0;
    const float c = cosf(yaw), s = sinf(yaw);
    while (n < cap) {
        float p[3];
        const int which = (int)(frand() * 6.0f);
        const float u = frand() * 8.0f - 4.0f, v = frand() * 8.0f - 4.0f, w = frand() * 2.5f;
        switch (which) {
            case 0: p[0] = u; p[1] = v; p[2] = 0.f; break;                /* floor */
            case 1: p[0] = 4.f; p[1] = v; p[2] = w; break;                 /* walls */
            case 2: p[0] = -4.f; p[1] = v; p[2] = w; break;
            case 3: p[0] = u; p[1] = 4.f; p[2] = w; break;
            case 4: p[0] = u; p[1] = -4.f; p[2] = w; break;
            default: p[0] = 1.f + 0.125f * u; p[1] = -1.5f; p[2] = 0.4f * w; break;   /* a box face */
        }
        for (int a = 0; a < 3; a++) p[a] += (frand() - 0.5f) * 0.01f;
        xyz[3 * n] = c * p[0] - s * p[1] + tx;
        xyz[3 * n + 1] = s * p[0] + c * p[1] + ty;
        xyz[3 * n + 2] = p[2];
        n++;
    }
    return n;
}

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "asan_driver: %s failed at line %d\n", #cond, __LINE__); return 1; } } while (0)

int main(void) {
    enum { N = 6000 };
    float* src = (float*)malloc(sizeof(float) * 3 * N);
    float* tgt = (float*)malloc(sizeof(float) * 3 * N);
    CHECK(src && tgt);
    lcg_state = 1u; make_room(tgt, N, 0.f, 0.f, 0.f);
    lcg_state = 2u; make_room(src, N, -0.02f, -0.05f, 0.03f);
    src[3 * 17] = NAN; tgt[3 * 5 + 1] = INFINITY;   /* non-finite points are skipped */

    /* ---- registration oracle: both metrics, one and two levels, source-only source, NN introspection, error paths ---- */
    for (int metric = 0; metric < 2; metric++) {
        for (int levels = 1; levels <= 2; levels++) {
            orc_params p;
            CHECK(orc_default_params(&p) == 0);
            p.n_levels = levels;
            p.leaf[0] = levels == 2 ? 0.5f : 0.25f; p.leaf[1] = 0.25f;
            p.iterations[0] = 6; p.iterations[1] = 6;
            p.max_corr_dist[0] = levels == 2 ? 1.0f : 0.5f; p.max_corr_dist[1] = 0.5f;
            p.metric = metric; p.normal_leaf = 0.5f;
            orc_cloud *cs = NULL, *cs2 = NULL, *ct = NULL;
            CHECK(orc_cloud_create(&p, tgt, N, 12, 0, 4, 8, &ct) == 0);
            CHECK(orc_cloud_create(&p, src, N, 12, 0, 4, 8, &cs) == 0);
            CHECK(orc_cloud_create_source(&p, src, N, 12, 0, 4, 8, &cs2) == 0);
            orc_grid_info gi;
            CHECK(orc_cloud_grid_info(ct, 0, &gi) == 0 && gi.n == N && gi.n_valid == N - 1 && gi.n_cells > 0);
            CHECK(orc_cloud_grid_info(ct, levels, &gi) != 0);   /* no such level */
            const float I[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
            float T1[16], T2[16];
            orc_stats s1, s2;
            double trace[16 * 16]; size_t tn = 0;
            CHECK(orc_align_clouds(&p, cs, ct, I, T1, &s1, trace, 16, &tn) == 0);
            CHECK(orc_align_clouds(&p, cs2, ct, I, T2, &s2, NULL, 0, NULL) == 0);
            CHECK(memcmp(T1, T2, sizeof(T1)) == 0 && s1.n_corr == s2.n_corr);   /* what a source has beyond its points is never read */
            CHECK(tn == (size_t)s1.iterations && s1.n_corr > N / 2);
            CHECK(fabsf(T1[12] - 0.05f) < 0.02f && fabsf(T1[13] + 0.03f) < 0.02f);   /* column-major translation ~ the inverse of the offset */
            int32_t idx[8]; float d2[8];
            CHECK(orc_debug_nn(ct, levels - 1, src + 3 * 100, 8, p.max_corr_dist[levels - 1], idx, d2) == 0);
            orc_cloud_destroy(cs); orc_cloud_destroy(cs2); orc_cloud_destroy(ct);
        }
    }
    {   /* error paths: empty cloud, absurd extent */
        orc_params p; orc_default_params(&p);
        orc_cloud* c = NULL;
        const float nanpt[3] = { NAN, NAN, NAN };
        CHECK(orc_cloud_create(&p, nanpt, 1, 12, 0, 4, 8, &c) != 0 && c == NULL);
        const float far2[6] = { 0.f, 0.f, 0.f, 1.0e7f, 1.0e7f, 1.0e7f };
        p.leaf[0] = 0.01f;
        CHECK(orc_cloud_create(&p, far2, 2, 12, 0, 4, 8, &c) != 0 && c == NULL);
    }
    orc_set_threads(2);

    /* ---- k-d tree baseline ---- */
    for (int metric = 0; metric < 2; metric++) {
        double T[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 }, ms[3];
        CHECK(kdicp_align(src, N, tgt, N, metric, 0.5f, 8, 8, 2, T, ms) > N / 2);
        CHECK(fabs(T[12] - 0.05) < 0.02 && fabs(T[13] + 0.03) < 0.02);
        CHECK(kdicp_align(NULL, 0, tgt, N, metric, 0.5f, 8, 8, 1, T, ms) < 0);
    }

    /* ---- sweep aggregator: PointCloud2-shaped and LaserScan-shaped messages ---- */
    {
        const double bb[6] = { -1.0, 1.0, -1.0, 1.0, -1.0, 1.0 };
        orc_agg* a = orc_agg_create(bb);
        CHECK(a != NULL);
        float ranges[360];
        for (int i = 0; i < 360; i++) ranges[i] = 2.0f + 0.01f * (float)i;
        ranges[7] = INFINITY; ranges[9] = NAN;
        for (int k = 0; k < 40 && !orc_agg_ready(a); k++) {
            const double ang = 0.2 * (double)k;
            const double tf[7] = { 0.0, 0.0, 0.5, 0.0, 0.0, sin(0.5 * ang), cos(0.5 * ang) };   /* {t, quaternion xyzw} */
            if (k & 1) orc_agg_add_scan(a, ranges, 360, -1.57f, 0.00873f, tf);
            else orc_agg_add_cloud(a, (const uint8_t*)tgt, 500, 12, 0, 4, 8, tf);
        }
        CHECK(orc_agg_count(a) > 0 && orc_agg_angle(a) > 0.0);
        (void)orc_agg_progress(a);
        float* out = (float*)malloc(16 * orc_agg_count(a));
        CHECK(out != NULL);
        orc_agg_points(a, out);
        free(out);
        orc_agg_restart(a);
        CHECK(orc_agg_count(a) == 0);
        orc_agg_destroy(a);
    }

    /* ---- calibration cost: voxel version against the brute-force restatement ---- */
    {
        enum { SEG = 4, PER = 300 };
        float* seg = (float*)malloc(sizeof(float) * 3 * SEG * PER);
        int32_t seg_n[SEG]; float seg_T[12 * SEG];
        CHECK(seg != NULL);
        lcg_state = 3u; make_room(seg, SEG * PER, 0.f, 0.f, 0.f);
        for (int s = 0; s < SEG; s++) {
            seg_n[s] = PER;
            const float ang = 1.5707963f * (float)s;
            const float Tm[12] = { cosf(ang), -sinf(ang), 0.f, sinf(ang), cosf(ang), 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.1f };
            memcpy(seg_T + 12 * s, Tm, sizeof(Tm));
        }
        const float prm[6] = { 0.01f, -0.02f, 0.005f, 0.01f, 0.f, -0.01f };
        int64_t sa[4];
        const int64_t ca = orc_cal_test_data(seg, seg_n, seg_T, SEG, 2, prm, sa);
        const int64_t cb = orc_cal_test_data_bruteforce(seg, seg_n, seg_T, SEG, 2, prm);
        CHECK(ca >= 0 && ca == cb && sa[0] + sa[1] == SEG * PER);
        float l[9], t[3], l2[9], t2[3];
        orc_cal_offset_matrix(prm, l, t);
        orc_cal_compose(l, t, l, t, l2, t2);
        free(seg);
    }

    /* ---- voxel map ---- */
    {
        orc_map* m = orc_map_create(0.2f, 4096);
        CHECK(m != NULL);
        const float I[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        const size_t a1 = orc_map_insert(m, tgt, N, I);
        const size_t a2 = orc_map_insert(m, tgt, N, I);   /* the same scan again adds nothing */
        CHECK(a1 > 0 && a2 == 0 && orc_map_size(m) == a1);
        float* out = (float*)malloc(sizeof(float) * 3 * orc_map_size(m));
        CHECK(out != NULL);
        orc_map_points(m, out);
        free(out);
        orc_map_destroy(m);
    }
    /* ---- loop-closure candidates (m3d_loop_oracle.c) and the specified float sine / cosine (m3d_agg_oracle.c) ---- */
    {
        orc_loop_params lp = { 1.0f, 10, 6.0f, 2, 3, 0.3f, 6, 0 };   /* 1024-bit signatures, capacity 6: the seventh keyframe is refused */
        orc_loop* L = orc_loop_create(&lp);
        CHECK(L != NULL);
        for (int k = 0; k < 7; k++) {
            const float Tk[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, (k == 3) ? 100.f : 0.5f * (float)k, 0, 0, 1 };
            const int idx = orc_loop_add(L, (k & 1) ? src : tgt, N, Tk);
            CHECK(idx == (k < 6 ? k : -1));
        }
        CHECK(orc_loop_size(L) == 6);
        orc_loop_candidate cand[18];
        const size_t n_all = orc_loop_candidates(L, 0, -1, cand, 18);
        const size_t n_cut = orc_loop_candidates(L, 0, -1, cand, 2);            /* a short output array */
        CHECK(n_all == n_cut && n_all <= 18);
        for (size_t i = 0; i < (n_all < 2 ? n_all : 2); i++) CHECK(cand[i].source - cand[i].target >= 2 && cand[i].overlap <= cand[i].pop_source);
        uint32_t words[32], pop = 0;
        orc_loop_signature(L, 5, words, &pop);
        const float T5[16] = { 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1, 0, 1.f, 2.f, 0.f, 1 };
        orc_loop_update(L, 5, src, N, T5);
        CHECK(orc_loop_candidates(L, 5, 1, cand, 18) <= 3);
        orc_loop_destroy(L);
        float sn, cs;
        orc_sincosf(0.5f, &sn, &cs);
        CHECK(fabsf(sn - sinf(0.5f)) < 1e-6f && fabsf(cs - cosf(0.5f)) < 1e-6f);
        orc_sincosf(-123456.f, &sn, &cs);
        CHECK(fabsf(sn * sn + cs * cs - 1.f) < 1e-5f);
        orc_sincosf(INFINITY, &sn, &cs);
        CHECK(sn != sn && cs != cs);
    }
    free(src); free(tgt);
    printf("asan_driver: ok\n");
    return 0;
}
