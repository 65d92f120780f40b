/* m3d_map_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into libm3dreg.so).
 * CPU restatement of the persistent map of SURVEY.md §8 row f4 (csrc/map.hip). No reference source exists for this step
 * (north_star's config 5 names it; /root/reference/gpu_6dslam is an empty submodule): PARITY UNPINNED — the spec is
 *   insert(scan, T): for every point of the scan IN INPUT ORDER: skip non-finite points; u = R p + t as the fma chain of
 *   spec row a5 (R, t = T rounded to float); skip non-finite u; voxel = floor(u * (1.0f / leaf)) per axis; a point whose
 *   voxel is already occupied (by an earlier insert or an earlier point of this scan) is dropped, otherwise it occupies
 *   the voxel and is appended. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAP_OFF 1048576
typedef struct { uint64_t* keys; size_t tsize; float* pts; size_t n, cap; float leaf; } orc_map;

orc_map* orc_map_create(float leaf, size_t cap) {
    orc_map* m = (orc_map*)calloc(1, sizeof(orc_map));
    m->leaf = leaf; m->cap = cap;
    m->tsize = 1024; while (m->tsize < 4 * cap) m->tsize <<= 1;
    m->keys = (uint64_t*)malloc(sizeof(uint64_t) * m->tsize);
    memset(m->keys, 0xFF, sizeof(uint64_t) * m->tsize);
    m->pts = (float*)malloc(sizeof(float) * 3 * (cap ? cap : 1));
    return m;
}
void orc_map_destroy(orc_map* m) { if (m) { free(m->keys); free(m->pts); free(m); } }
size_t orc_map_size(const orc_map* m) { return m->n; }
void orc_map_points(const orc_map* m, float* out) { memcpy(out, m->pts, sizeof(float) * 3 * m->n); }

/* xyz: n points, 3 floats each, input order; T column-major float[16]. Returns the number of points added. */
size_t orc_map_insert(orc_map* m, const float* xyz, size_t n, const float T[16]) {
    float R[9], t[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = T[c * 4 + r]; t[r] = T[12 + r]; }
    const float inv = 1.0f / m->leaf;
    const float lim = (float)(MAP_OFF - 1);
    size_t added = 0;
    for (size_t i = 0; i < n; i++) {
        const float* p = xyz + 3 * i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        float u[3];
        for (int r = 0; r < 3; r++) u[r] = fmaf(R[3 * r], p[0], fmaf(R[3 * r + 1], p[1], fmaf(R[3 * r + 2], p[2], t[r])));
        if (!(isfinite(u[0]) && isfinite(u[1]) && isfinite(u[2]))) continue;
        const float fx = floorf(u[0] * inv), fy = floorf(u[1] * inv), fz = floorf(u[2] * inv);
        if (!(fx > -lim && fx < lim && fy > -lim && fy < lim && fz > -lim && fz < lim)) continue;
        const uint64_t key = ((uint64_t)(uint32_t)((int)fx + MAP_OFF) << 42) | ((uint64_t)(uint32_t)((int)fy + MAP_OFF) << 21) | (uint64_t)(uint32_t)((int)fz + MAP_OFF);
        size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (m->tsize - 1);
        while (m->keys[h] != key && m->keys[h] != ~0ull) h = (h + 1) & (m->tsize - 1);
        if (m->keys[h] == key) continue;
        if (m->n >= m->cap) continue;
        m->keys[h] = key;
        memcpy(m->pts + 3 * m->n, u, sizeof(float) * 3);
        m->n++; added++;
    }
    return added;
}
