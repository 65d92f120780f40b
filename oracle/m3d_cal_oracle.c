/* m3d_cal_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into libm3dreg.so; see oracle/Makefile).
 *
 * CPU restatement of the calibration cost function of SURVEY.md §8 row f2:
 *   /root/reference/m3d/m3d_calibration/src/m3d_calibration_twiddle.cpp:199-308  (`testData`)
 *   (the same function, line for line, at m3d_calibration_sa.cpp:199-277).
 * The control flow and every constant follow those lines. The arithmetic INSIDE the calls the reference makes
 * into un-vendored dependencies is not in the tree (find_package(PCL 1.5), Eigen 3, FLANN — versions unpinned,
 * m3d/m3d_calibration/CMakeLists.txt:17), so it is restated from their published scalar code paths and the
 * restatement is the spec both sides (this file and csrc/calibrate.hip) follow:  PARITY UNPINNED against PCL.
 *   Eigen::AngleAxisf -> Quaternionf, quaternion product, toRotationMatrix, Transform::rotate / translate,
 *       Affine3f * Affine3f, Affine3f * Vector3f: float, left-to-right sums, no fused multiply-add
 *   pcl::VoxelGrid(0.1): voxel = floor(coordinate * (1.0f / 0.1f)); one output point per occupied voxel = centroid
 *       of its points. PCL adds floats in the order an unstable std::sort leaves them (not reproducible even
 *       against itself); here the coordinates are summed in 2^-16 m fixed point (int64: any order, same bits)
 *   pcl::KdTreeFLANN::radiusSearch(p, 0.05): neighbours with squared float distance < (float)(0.05 * 0.05)
 * Non-finite transformed points are dropped (PCL would index with an undefined voxel).                         */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float l[9]; float t[3]; } cal_affine;   /* row-major linear part + translation */

/* twiddle.cpp:202-220: Affine3f::Identity().rotate(AAx(yaw) * AAy(pitch) * AAz(roll)).translate((x, y, z)) */
void orc_cal_offset_matrix(const float p[6], float out_l[9], float out_t[3]) {
    const float ha0 = 0.5f * p[3], ha1 = 0.5f * p[4], ha2 = 0.5f * p[5];
    /* quaternions (w, x, y, z) of the three axis rotations */
    const float aw = cosf(ha0), ax = sinf(ha0) * 1.0f, ay = sinf(ha0) * 0.0f, az = sinf(ha0) * 0.0f;
    const float bw = cosf(ha1), bx = sinf(ha1) * 0.0f, by = sinf(ha1) * 1.0f, bz = sinf(ha1) * 0.0f;
    const float cw = cosf(ha2), cx = sinf(ha2) * 0.0f, cy = sinf(ha2) * 0.0f, cz = sinf(ha2) * 1.0f;
    /* q = (a * b) * c, Eigen's generic quaternion product */
    const float dw = aw * bw - ax * bx - ay * by - az * bz;
    const float dx = aw * bx + ax * bw + ay * bz - az * by;
    const float dy = aw * by + ay * bw + az * bx - ax * bz;
    const float dz = aw * bz + az * bw + ax * by - ay * bx;
    const float w = dw * cw - dx * cx - dy * cy - dz * cz;
    const float x = dw * cx + dx * cw + dy * cz - dz * cy;
    const float y = dw * cy + dy * cw + dz * cx - dx * cz;
    const float z = dw * cz + dz * cw + dx * cy - dy * cx;
    /* QuaternionBase::toRotationMatrix */
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    float R[9];
    R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
    /* Identity.rotate(R): linear = I * R; .translate(t): translation += linear * t */
    for (int i = 0; i < 9; i++) out_l[i] = R[i];
    for (int r = 0; r < 3; r++) out_t[r] = 0.0f + ((R[3 * r] * p[0] + R[3 * r + 1] * p[1]) + R[3 * r + 2] * p[2]);
}

/* twiddle.cpp:229: mm = original_Transform * laserOffsetMatrix */
void orc_cal_compose(const float al[9], const float at[3], const float bl[9], const float bt[3], float ol[9], float ot[3]) {
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) ol[3 * r + c] = (al[3 * r] * bl[c] + al[3 * r + 1] * bl[3 + c]) + al[3 * r + 2] * bl[6 + c];
        ot[r] = ((al[3 * r] * bt[0] + al[3 * r + 1] * bt[1]) + al[3 * r + 2] * bt[2]) + at[r];
    }
}

typedef struct { uint64_t key; int64_t q[3]; } cal_item;
static int cmp_item(const void* a, const void* b) {
    const uint64_t ka = ((const cal_item*)a)->key, kb = ((const cal_item*)b)->key;
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}
#define CAL_OFF 1048576 /* 2^20: voxel coordinates are stored with this offset in 21 bits each */
static uint64_t cal_key(int set, int ix, int iy, int iz) {
    return ((uint64_t)set << 63) | ((uint64_t)(uint32_t)(ix + CAL_OFF) << 42) | ((uint64_t)(uint32_t)(iy + CAL_OFF) << 21) | (uint64_t)(uint32_t)(iz + CAL_OFF);
}
typedef struct { uint64_t key; float c[3]; } cal_centroid;
static const cal_centroid* find_centroid(const cal_centroid* v, size_t n, uint64_t key) {
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (v[mid].key < key) lo = mid + 1; else hi = mid; }
    return (lo < n && v[lo].key == key) ? &v[lo] : NULL;
}

/* twiddle.cpp:199-308. seg_xyz: all segments' raw points back to back (3 floats each); seg_n[s] points in segment s;
 * seg_T: per segment 12 floats {linear row-major (9), translation (3)} of original_Transform.
 * Returns the count `c` (>= 0) or -1 (voxel coordinate outside +-2^20, allocation failure).
 * out_sizes (optional): {points in firstPc, points in secondPc, voxels of firstPcFilter, voxels of secondPcFilter} */
int64_t orc_cal_test_data(const float* seg_xyz, const int32_t* seg_n, const float* seg_T, int32_t n_seg, int32_t laser_up_axis,
                          const float params[6], int64_t out_sizes[4]) {
    float ol[9], ot[3];
    orc_cal_offset_matrix(params, ol, ot);
    size_t total = 0;
    for (int s = 0; s < n_seg; s++) total += (size_t)seg_n[s];
    cal_item* it = (cal_item*)malloc(sizeof(cal_item) * (total ? total : 1));
    if (!it) return -1;
    const float inv_leaf = 1.0f / 0.1f;   /* VoxelGrid::setLeafSize(0.1, 0.1, 0.1): inverse_leaf_size_ = 1 / leaf_size_ (float) */
    size_t m = 0, n_first = 0, n_second = 0, off = 0;
    for (int s = 0; s < n_seg; s++) {
        float ml[9], mt[3];
        orc_cal_compose(seg_T + 12 * s, seg_T + 12 * s + 9, ol, ot, ml, mt);
        for (int j = 0; j < seg_n[s]; j++, off++) {
            const float* p = seg_xyz + 3 * off;
            float u[3];
            for (int r = 0; r < 3; r++) u[r] = ((ml[3 * r] * p[0] + ml[3 * r + 1] * p[1]) + ml[3 * r + 2] * p[2]) + mt[r];   /* :230 */
            const int set = (p[laser_up_axis] > 0) ? 0 : 1;   /* :234-266: split on the sign of the RAW coordinate */
            if (set == 0) n_first++; else n_second++;
            if (!(isfinite(u[0]) && isfinite(u[1]) && isfinite(u[2]))) continue;
            int v[3];
            for (int r = 0; r < 3; r++) {
                const float f = floorf(u[r] * inv_leaf);
                if (!(f > -(float)CAL_OFF && f < (float)(CAL_OFF - 1))) { free(it); return -1; }
                v[r] = (int)f;
                it[m].q[r] = (int64_t)rintf(u[r] * 65536.0f);
            }
            it[m].key = cal_key(set, v[0], v[1], v[2]);
            m++;
        }
    }
    qsort(it, m, sizeof(cal_item), cmp_item);
    /* :279-286 one centroid per occupied voxel, first set then second (bit 63 of the key) */
    cal_centroid* cen = (cal_centroid*)malloc(sizeof(cal_centroid) * (m ? m : 1));
    if (!cen) { free(it); return -1; }
    size_t nc = 0, n_first_vox = 0;
    for (size_t i = 0; i < m;) {
        size_t j = i; int64_t sx = 0, sy = 0, sz = 0;
        while (j < m && it[j].key == it[i].key) { sx += it[j].q[0]; sy += it[j].q[1]; sz += it[j].q[2]; j++; }
        const double n = (double)(j - i);
        cen[nc].key = it[i].key;
        cen[nc].c[0] = (float)(((double)sx / n) * (1.0 / 65536.0));
        cen[nc].c[1] = (float)(((double)sy / n) * (1.0 / 65536.0));
        cen[nc].c[2] = (float)(((double)sz / n) * (1.0 / 65536.0));
        if (!(it[i].key >> 63)) n_first_vox++;
        nc++; i = j;
    }
    /* :288-304 count the second-half centroids with no first-half centroid closer than 0.05 m */
    const float r2 = (float)(0.05 * 0.05);
    const float rr = 0.0505f;   /* search box half-width: the radius plus a margin far above the rounding of the sums */
    int64_t c = 0;
    for (size_t i = n_first_vox; i < nc; i++) {
        const float* q = cen[i].c;
        int lo[3], hi[3];
        for (int r = 0; r < 3; r++) { lo[r] = (int)floorf((q[r] - rr) * inv_leaf); hi[r] = (int)floorf((q[r] + rr) * inv_leaf); }
        int found = 0;
        for (int ix = lo[0]; ix <= hi[0] && !found; ix++)
            for (int iy = lo[1]; iy <= hi[1] && !found; iy++)
                for (int iz = lo[2]; iz <= hi[2] && !found; iz++) {
                    if (ix <= -CAL_OFF || ix >= CAL_OFF - 1 || iy <= -CAL_OFF || iy >= CAL_OFF - 1 || iz <= -CAL_OFF || iz >= CAL_OFF - 1) continue;
                    const cal_centroid* f = find_centroid(cen, n_first_vox, cal_key(0, ix, iy, iz));
                    if (!f) continue;
                    const float dx = q[0] - f->c[0], dy = q[1] - f->c[1], dz = q[2] - f->c[2];
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < r2) found = 1;
                }
        if (!found) c++;
    }
    if (out_sizes) { out_sizes[0] = (int64_t)n_first; out_sizes[1] = (int64_t)n_second; out_sizes[2] = (int64_t)n_first_vox; out_sizes[3] = (int64_t)(nc - n_first_vox); }
    free(it); free(cen);
    return c;
}

/* brute-force check of the neighbour count (tests only): same centroids, every pair compared */
int64_t orc_cal_test_data_bruteforce(const float* seg_xyz, const int32_t* seg_n, const float* seg_T, int32_t n_seg, int32_t laser_up_axis,
                                     const float params[6]) {
    float ol[9], ot[3];
    orc_cal_offset_matrix(params, ol, ot);
    size_t total = 0;
    for (int s = 0; s < n_seg; s++) total += (size_t)seg_n[s];
    cal_item* it = (cal_item*)malloc(sizeof(cal_item) * (total ? total : 1));
    if (!it) return -1;
    const float inv_leaf = 1.0f / 0.1f;
    size_t m = 0, off = 0;
    for (int s = 0; s < n_seg; s++) {
        float ml[9], mt[3];
        orc_cal_compose(seg_T + 12 * s, seg_T + 12 * s + 9, ol, ot, ml, mt);
        for (int j = 0; j < seg_n[s]; j++, off++) {
            const float* p = seg_xyz + 3 * off;
            float u[3];
            for (int r = 0; r < 3; r++) u[r] = ((ml[3 * r] * p[0] + ml[3 * r + 1] * p[1]) + ml[3 * r + 2] * p[2]) + mt[r];
            if (!(isfinite(u[0]) && isfinite(u[1]) && isfinite(u[2]))) continue;
            const int set = (p[laser_up_axis] > 0) ? 0 : 1;
            int v[3];
            for (int r = 0; r < 3; r++) { v[r] = (int)floorf(u[r] * inv_leaf); it[m].q[r] = (int64_t)rintf(u[r] * 65536.0f); }
            it[m].key = cal_key(set, v[0], v[1], v[2]);
            m++;
        }
    }
    qsort(it, m, sizeof(cal_item), cmp_item);
    cal_centroid* cen = (cal_centroid*)malloc(sizeof(cal_centroid) * (m ? m : 1));
    size_t nc = 0, nf = 0;
    for (size_t i = 0; i < m;) {
        size_t j = i; int64_t sx = 0, sy = 0, sz = 0;
        while (j < m && it[j].key == it[i].key) { sx += it[j].q[0]; sy += it[j].q[1]; sz += it[j].q[2]; j++; }
        const double n = (double)(j - i);
        cen[nc].key = it[i].key;
        cen[nc].c[0] = (float)(((double)sx / n) * (1.0 / 65536.0));
        cen[nc].c[1] = (float)(((double)sy / n) * (1.0 / 65536.0));
        cen[nc].c[2] = (float)(((double)sz / n) * (1.0 / 65536.0));
        if (!(it[i].key >> 63)) nf++;
        nc++; i = j;
    }
    const float r2 = (float)(0.05 * 0.05);
    int64_t c = 0;
    for (size_t i = nf; i < nc; i++) {
        int found = 0;
        for (size_t k = 0; k < nf && !found; k++) {
            const float dx = cen[i].c[0] - cen[k].c[0], dy = cen[i].c[1] - cen[k].c[1], dz = cen[i].c[2] - cen[k].c[2];
            if ((dx * dx + dy * dy) + dz * dz < r2) found = 1;
        }
        if (!found) c++;
    }
    free(it); free(cen);
    return c;
}
