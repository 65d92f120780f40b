/*
 * m3d_kdtree_icp.c — TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg "cpu-kdtree"; never linked into libm3dreg.so, never
 * imported by the product package).
 *
 * The CPU algorithm a ROS/PCL user would run in the gpu_6dslam slot: the reference includes <pcl/registration/icp.h> but never
 * calls it (/root/reference/m3d/m3d_calibration/src/m3d_calibration_sa.cpp:22, m3d_calibration_twiddle.cpp:22), and the only
 * neighbour-search machinery in the tree is pcl::KdTreeFLANN (m3d_calibration_twiddle.cpp:288-304). PCL and FLANN are not in the
 * tree and not in this image, so this file restates the published algorithm of pcl::IterativeClosestPoint with a point-to-plane
 * estimator from scratch: k-d tree over the target (median split on the widest axis, leaves of up to 12 points), target normals
 * from the PCA of every point's k nearest neighbours (pcl::NormalEstimation, k = 10), per iteration a 1-NN query per source point
 * with the max-distance reject, the linearised point-to-plane (or closed-form-free point-to-point Gauss-Newton) 6x6 system in
 * double, Cholesky solve, SE(3) update. OpenMP over the queries. It is NOT bit-comparable with anything — it is the second CPU
 * figure beside the port of the voxel algorithm (m3d_oracle.c), timed on the same host, same clouds, same iteration count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { int lo, hi; int left, right; int axis; float split; float bmin[3], bmax[3]; } kd_node;
typedef struct { const float* p; int n; int* idx; kd_node* nodes; int n_nodes, cap_nodes; } kd_tree;

#define KD_LEAF 12

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec; }

static void select_nth(const float* p, int* idx, int lo, int hi, int nth, int axis) {   /* Hoare quickselect on idx[lo..hi) */
    while (hi - lo > 1) {
        const float pivot = p[3 * idx[lo + (hi - lo) / 2] + axis];
        int i = lo, j = hi - 1;
        while (i <= j) {
            while (p[3 * idx[i] + axis] < pivot) i++;
            while (p[3 * idx[j] + axis] > pivot) j--;
            if (i <= j) { const int t = idx[i]; idx[i] = idx[j]; idx[j] = t; i++; j--; }
        }
        if (nth <= j) hi = j + 1; else if (nth >= i) lo = i; else return;
    }
}

static int kd_build_rec(kd_tree* T, int lo, int hi) {
    const int me = T->n_nodes++;
    kd_node* N = &T->nodes[me];
    N->lo = lo; N->hi = hi; N->left = N->right = -1;
    for (int a = 0; a < 3; a++) { N->bmin[a] = INFINITY; N->bmax[a] = -INFINITY; }
    for (int i = lo; i < hi; i++)
        for (int a = 0; a < 3; a++) { const float v = T->p[3 * T->idx[i] + a]; if (v < N->bmin[a]) N->bmin[a] = v; if (v > N->bmax[a]) N->bmax[a] = v; }
    if (hi - lo <= KD_LEAF) return me;
    int ax = 0; float ext = N->bmax[0] - N->bmin[0];
    for (int a = 1; a < 3; a++) if (N->bmax[a] - N->bmin[a] > ext) { ext = N->bmax[a] - N->bmin[a]; ax = a; }
    const int mid = lo + (hi - lo) / 2;
    select_nth(T->p, T->idx, lo, hi, mid, ax);
    N->axis = ax; N->split = T->p[3 * T->idx[mid] + ax];
    const int l = kd_build_rec(T, lo, mid);
    const int r = kd_build_rec(T, mid, hi);
    T->nodes[me].left = l; T->nodes[me].right = r;   /* (T->nodes does not move: allocated up front) */
    return me;
}

static int kd_build(kd_tree* T, const float* p, int n) {
    T->p = p; T->n = n; T->n_nodes = 0;
    T->idx = (int*)malloc(sizeof(int) * (size_t)n);
    T->cap_nodes = 2 * (n / (KD_LEAF / 2) + 2);
    T->nodes = (kd_node*)malloc(sizeof(kd_node) * (size_t)T->cap_nodes);
    if (!T->idx || !T->nodes) return -1;
    int m = 0;
    for (int i = 0; i < n; i++) if (isfinite(p[3 * i]) && isfinite(p[3 * i + 1]) && isfinite(p[3 * i + 2])) T->idx[m++] = i;
    T->n = m;
    if (m == 0) return -2;
    kd_build_rec(T, 0, m);
    return 0;
}
static void kd_free(kd_tree* T) { free(T->idx); free(T->nodes); T->idx = NULL; T->nodes = NULL; }

static inline float box_d2(const kd_node* N, const float q[3]) {
    float d2 = 0.f;
    for (int a = 0; a < 3; a++) { const float v = q[a] < N->bmin[a] ? N->bmin[a] - q[a] : (q[a] > N->bmax[a] ? q[a] - N->bmax[a] : 0.f); d2 += v * v; }
    return d2;
}

/* k nearest neighbours (k <= 16): best[] is a max-heap-free sorted insertion list; returns how many were found within r2 */
typedef struct { float d2[16]; int id[16]; int k, n; } kd_knn;
static void kd_search(const kd_tree* T, int node, const float q[3], kd_knn* R) {
    const kd_node* N = &T->nodes[node];
    const float worst = R->n < R->k ? INFINITY : R->d2[R->n - 1];
    if (box_d2(N, q) > worst) return;
    if (N->left < 0) {
        for (int i = N->lo; i < N->hi; i++) {
            const int j = T->idx[i];
            const float dx = q[0] - T->p[3 * j], dy = q[1] - T->p[3 * j + 1], dz = q[2] - T->p[3 * j + 2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (R->n == R->k && d2 >= R->d2[R->n - 1]) continue;
            int pos = R->n < R->k ? R->n++ : R->n - 1;
            while (pos > 0 && R->d2[pos - 1] > d2) { R->d2[pos] = R->d2[pos - 1]; R->id[pos] = R->id[pos - 1]; pos--; }
            R->d2[pos] = d2; R->id[pos] = j;
        }
        return;
    }
    const int first = q[N->axis] < N->split ? N->left : N->right, second = first == N->left ? N->right : N->left;
    kd_search(T, first, q, R);
    kd_search(T, second, q, R);
}

/* smallest eigenvector of a symmetric 3x3 (cyclic Jacobi, double) */
static void sym3_min_eigvec(double A[3][3], double v[3]) {
    double V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int sweep = 0; sweep < 12; sweep++) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-18) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    for (int k = 0; k < 3; k++) v[k] = V[k][m];
}

static int chol6_solve(double H[6][6], double g[6], double x[6]) {
    double L[6][6] = { { 0 } };
    for (int j = 0; j < 6; j++) {
        double d = H[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        if (!(d > 1e-12)) return -1;
        L[j][j] = sqrt(d);
        for (int i = j + 1; i < 6; i++) { double v = H[i][j]; for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k]; L[i][j] = v / L[j][j]; }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double v = -g[i]; for (int k = 0; k < i; k++) v -= L[i][k] * y[k]; y[i] = v / L[i][i]; }
    for (int i = 5; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 6; k++) v -= L[k][i] * x[k]; x[i] = v / L[i][i]; }
    return 0;
}

/*
 * One registration. src / tgt: xyz interleaved float32 (non-finite points are skipped). metric 0 = point-to-point, 1 = point-to-plane.
 * T (column-major double[16]) is the initial guess on entry and the result on return. ms[3] = {tree build, normals, iterations}.
 * Returns the number of correspondences of the last iteration, or < 0 on failure.
 */
long long kdicp_align(const float* src, int n_src, const float* tgt, int n_tgt, int metric, float max_corr_dist, int iterations, int normal_k,
                      int threads, double T[16], double ms[3]) {
    if (!src || !tgt || !T || n_src <= 0 || n_tgt <= 0 || iterations < 0) return -1;
    if (normal_k < 3) normal_k = 3;
    if (normal_k > 16) normal_k = 16;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    double t0 = now_ms();
    kd_tree K;
    if (kd_build(&K, tgt, n_tgt) != 0) return -2;
    double t1 = now_ms();
    float* nrm = NULL;
    if (metric == 1) {
        nrm = (float*)calloc((size_t)n_tgt * 3, sizeof(float));
        if (!nrm) { kd_free(&K); return -3; }
#pragma omp parallel for schedule(dynamic, 256)
        for (int ii = 0; ii < K.n; ii++) {
            const int i = K.idx[ii];
            kd_knn R; R.k = normal_k; R.n = 0;
            kd_search(&K, 0, &tgt[3 * i], &R);
            if (R.n < 3) continue;
            double m[3] = { 0, 0, 0 };
            for (int k = 0; k < R.n; k++) for (int a = 0; a < 3; a++) m[a] += tgt[3 * R.id[k] + a];
            for (int a = 0; a < 3; a++) m[a] /= R.n;
            double C[3][3] = { { 0 } };
            for (int k = 0; k < R.n; k++) {
                const double d[3] = { tgt[3 * R.id[k]] - m[0], tgt[3 * R.id[k] + 1] - m[1], tgt[3 * R.id[k] + 2] - m[2] };
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) C[a][b] += d[a] * d[b];
            }
            double v[3];
            sym3_min_eigvec(C, v);
            for (int a = 0; a < 3; a++) nrm[3 * i + a] = (float)v[a];
        }
    }
    double t2 = now_ms();
    const float dmax2 = max_corr_dist * max_corr_dist;
    long long n_corr = 0;
    for (int it = 0; it < iterations; it++) {
        double H[6][6] = { { 0 } }, g[6] = { 0 };
        double h21[21] = { 0 }, g6[6] = { 0 };
        long long cnt = 0;
        const double R0 = T[0], R1 = T[4], R2 = T[8], R3 = T[1], R4 = T[5], R5 = T[9], R6 = T[2], R7 = T[6], R8 = T[10], tx = T[12], ty = T[13], tz = T[14];
#pragma omp parallel for schedule(static) reduction(+ : h21[:21], g6[:6], cnt)
        for (int i = 0; i < n_src; i++) {
            const float px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
            if (!(isfinite(px) && isfinite(py) && isfinite(pz))) continue;
            const float q[3] = { (float)(R0 * px + R1 * py + R2 * pz + tx), (float)(R3 * px + R4 * py + R5 * pz + ty), (float)(R6 * px + R7 * py + R8 * pz + tz) };
            kd_knn R; R.k = 1; R.n = 0;
            kd_search(&K, 0, q, &R);
            if (R.n == 0 || R.d2[0] > dmax2) continue;
            const int j = R.id[0];
            const double e[3] = { q[0] - tgt[3 * j], q[1] - tgt[3 * j + 1], q[2] - tgt[3 * j + 2] };
            if (metric == 1) {
                const double n[3] = { nrm[3 * j], nrm[3 * j + 1], nrm[3 * j + 2] };
                if (n[0] == 0.0 && n[1] == 0.0 && n[2] == 0.0) continue;
                const double J[6] = { q[1] * n[2] - q[2] * n[1], q[2] * n[0] - q[0] * n[2], q[0] * n[1] - q[1] * n[0], n[0], n[1], n[2] };
                const double r = n[0] * e[0] + n[1] * e[1] + n[2] * e[2];
                int s = 0;
                for (int a = 0; a < 6; a++) { for (int b = a; b < 6; b++) h21[s++] += J[a] * J[b]; g6[a] += J[a] * r; }
            } else {
                /* rows of J = [-[q]x | I] */
                const double Jr[3][6] = { { 0, q[2], -q[1], 1, 0, 0 }, { -q[2], 0, q[0], 0, 1, 0 }, { q[1], -q[0], 0, 0, 0, 1 } };
                for (int rr = 0; rr < 3; rr++) {
                    int s = 0;
                    for (int a = 0; a < 6; a++) { for (int b = a; b < 6; b++) h21[s++] += Jr[rr][a] * Jr[rr][b]; g6[a] += Jr[rr][a] * e[rr]; }
                }
            }
            cnt++;
        }
        n_corr = cnt;
        if (cnt < 6) break;
        { int s = 0; for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { H[a][b] = H[b][a] = h21[s++]; } for (int a = 0; a < 6; a++) g[a] = g6[a]; }
        double x[6];
        if (chol6_solve(H, g, x) != 0) break;
        /* T <- exp([w, v]) T (Rodrigues; translation taken as v: first order, as the linearisation) */
        const double th = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const double A = th > 1e-12 ? sin(th) / th : 1.0, B = th > 1e-12 ? (1.0 - cos(th)) / (th * th) : 0.5;
        const double W[9] = { 0, -x[2], x[1], x[2], 0, -x[0], -x[1], x[0], 0 };
        double W2[9], Re[9];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { double v = 0; for (int k = 0; k < 3; k++) v += W[3 * a + k] * W[3 * k + b]; W2[3 * a + b] = v; }
        for (int k = 0; k < 9; k++) Re[k] = (k % 4 == 0 ? 1.0 : 0.0) + A * W[k] + B * W2[k];
        double Tn[16];
        for (int c = 0; c < 4; c++) {
            for (int r = 0; r < 3; r++) Tn[4 * c + r] = Re[3 * r] * T[4 * c] + Re[3 * r + 1] * T[4 * c + 1] + Re[3 * r + 2] * T[4 * c + 2] + (c == 3 ? x[3 + r] : 0.0);
            Tn[4 * c + 3] = c == 3 ? 1.0 : 0.0;
        }
        memcpy(T, Tn, sizeof(Tn));
    }
    double t3 = now_ms();
    if (ms) { ms[0] = t1 - t0; ms[1] = t2 - t1; ms[2] = t3 - t2; }
    free(nrm);
    kd_free(&K);
    return n_corr;
}
