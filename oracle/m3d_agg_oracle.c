/*
 * m3d_agg_oracle.c — CPU restatement (plain C99) of the aggregation step that precedes the registration
 * path. TEST INFRASTRUCTURE ONLY, same rules as m3d_oracle.c.
 *
 * Unlike the registration path, this step EXISTS in the reference; every function cites the lines it follows:
 *   /root/reference/m3d/m3d_aggregator/src/m3d_aggregator.cpp
 *     :53-88   pointCloudAggregator::addPoints   (rigid transform in double, outside-box filter, angular distance)
 *     :95-103  isPointcloudReady, :119-124 getProgress, :108-114 clearPointCloud, :30 angularDistance = 1.1*pi
 *     :231-254 rotLaserPointCloudCallback (PointCloud2 in), :256-288 rotLaserScanCallback (LaserScan polar -> xyz)
 * The tf::Transform / tf::Quaternion arithmetic those lines call lives in the third-party `tf` package
 * (LinearMath, un-vendored; catkin dependency at m3d/m3d_aggregator/CMakeLists.txt:8-16, version unpinned):
 * its published algorithms (Matrix3x3::setRotation / getRotation, Quaternion::angleShortestPath) are
 * restated below in double precision with the same operation order.
 *
 * Deliberate deviation: the reference never initialises currentAngularDistance /
 * creatingPointCloud in its constructor (m3d_aggregator.cpp:27-40, UB); here both start as after
 * clearPointCloud() + createPointCloud().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double bb[6];            /* x_up, x_down, y_up, y_down, z_up, z_down  (setBBox :42-52) */
    double current_angle;    /* currentAngularDistance */
    double angular_distance; /* 1.1 * M_PI (:30) */
    int creating, first_scan;
    double actual[4];        /* quaternion x,y,z,w of the previous call (:80,:86) */
    float* pts;              /* kept points, pcl::PointXYZ layout: x,y,z,pad (16 bytes) */
    size_t n, cap;
} orc_agg;

/* tf::Matrix3x3::setRotation (LinearMath/Matrix3x3.h), row-major m[9] */
static void set_rotation(const double q[4], double m[9]) {
    const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double s = 2.0 / d;
    const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
    const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
    const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
    const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
    m[0] = 1.0 - (yy + zz); m[1] = xy - wz; m[2] = xz + wy;
    m[3] = xy + wz; m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
    m[6] = xz - wy; m[7] = yz + wx; m[8] = 1.0 - (xx + yy);
}

/* tf::Matrix3x3::getRotation (used by tf::Transform::getRotation, m3d_aggregator.cpp:75) */
static void get_rotation(const double m[9], double q[4]) {
    const double trace = m[0] + m[4] + m[8];
    double temp[4];
    if (trace > 0.0) {
        double s = sqrt(trace + 1.0);
        temp[3] = s * 0.5;
        s = 0.5 / s;
        temp[0] = (m[7] - m[5]) * s;
        temp[1] = (m[2] - m[6]) * s;
        temp[2] = (m[3] - m[1]) * s;
    } else {
        int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
        int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = sqrt(m[3 * i + i] - m[3 * j + j] - m[3 * k + k] + 1.0);
        temp[i] = s * 0.5;
        s = 0.5 / s;
        temp[3] = (m[3 * k + j] - m[3 * j + k]) * s;
        temp[j] = (m[3 * j + i] + m[3 * i + j]) * s;
        temp[k] = (m[3 * k + i] + m[3 * i + k]) * s;
    }
    memcpy(q, temp, sizeof(temp));
}

/* tf::Quaternion::angleShortestPath */
static double angle_shortest_path(const double a[4], const double b[4]) {
    const double la = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    const double lb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    const double s = sqrt(la * lb);
    const double dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    if (dot < 0.0) return acos(-dot / s) * 2.0;   /* dot(-q) == -dot(q) exactly */
    return acos(dot / s) * 2.0;
}

orc_agg* orc_agg_create(const double bb[6]) {
    orc_agg* a = (orc_agg*)calloc(1, sizeof(orc_agg));
    memcpy(a->bb, bb, sizeof(a->bb));
    a->angular_distance = 1.1 * 3.14159265358979323846;   /* 1.1 * M_PI (:30) */
    a->creating = 1; a->first_scan = 1;   /* createPointCloud() :90-94, called from the node ctor :183 */
    return a;
}
void orc_agg_destroy(orc_agg* a) { if (a) { free(a->pts); free(a); } }

/* clearPointCloud :108-114 followed by createPointCloud :90-94 == requestCallback :224-229 */
void orc_agg_restart(orc_agg* a) { a->n = 0; a->current_angle = 0.0; a->first_scan = 1; a->creating = 1; }

/* addPoints :53-88 for one point */
static void add_point(orc_agg* a, float px, float py, float pz, const double m[9], const double o[3], const double q[4]) {
    if (!a->creating) return;
    const double x = (double)px, y = (double)py, z = (double)pz;
    const double p1[3] = { m[0] * x + m[1] * y + m[2] * z + o[0], m[3] * x + m[4] * y + m[5] * z + o[1], m[6] * x + m[7] * y + m[8] * z + o[2] };
    const float pp[3] = { (float)p1[0], (float)p1[1], (float)p1[2] };
    if (((double)pp[0] > a->bb[0]) || ((double)pp[0] < a->bb[1]) || ((double)pp[1] > a->bb[2]) || ((double)pp[1] < a->bb[3]) ||
        ((double)pp[2] > a->bb[4]) || ((double)pp[2] < a->bb[5])) {
        if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 4096; a->pts = (float*)realloc(a->pts, 16 * a->cap); }
        float* d = &a->pts[4 * a->n++];
        d[0] = pp[0]; d[1] = pp[1]; d[2] = pp[2]; d[3] = 0.0f;
    }
    if (a->first_scan) { a->first_scan = 0; memcpy(a->actual, q, 32); }
    else {
        const double dd = angle_shortest_path(q, a->actual);
        if (!isnan(dd)) a->current_angle = a->current_angle + dd;
        memcpy(a->actual, q, 32);
    }
}

static void make_tf(const double tf[7], double m[9], double o[3], double q[4]) {
    const double qin[4] = { tf[3], tf[4], tf[5], tf[6] };
    set_rotation(qin, m);          /* tf::transformMsgToTF -> Transform(Quaternion, Vector3) (:248, :268) */
    o[0] = tf[0]; o[1] = tf[1]; o[2] = tf[2];
    get_rotation(m, q);            /* transform.getRotation() (:75) */
}

/* rotLaserPointCloudCallback :231-254; tf = tx,ty,tz,qx,qy,qz,qw of lookupTransform */
void orc_agg_add_cloud(orc_agg* a, const uint8_t* data, size_t n, size_t step, size_t ox, size_t oy, size_t oz, const double tf[7]) {
    double m[9], o[3], q[4];
    make_tf(tf, m, o, q);
    for (size_t i = 0; i < n; i++) {
        float p[3];
        memcpy(&p[0], data + i * step + ox, 4); memcpy(&p[1], data + i * step + oy, 4); memcpy(&p[2], data + i * step + oz, 4);
        add_point(a, p[0], p[1], p[2], m, o, q);
    }
}

/* rotLaserScanCallback :256-288.
 * WHICH cos / sin: lines 281-282 read `point.x = cos(ang)*dist;` with `float ang, dist`, unqualified, in a file that includes no
 * <cmath> / <math.h> itself (they arrive through the ROS headers) and has no using-directive. Two readings exist, and both are restated:
 *   float_overload == 0 (the DEFAULT, what the toolchains this ROS1 code was written for do — GCC < 6 / libstdc++ whose <cmath> leaves only C's
 *     `double cos(double)` in the global namespace): ang is promoted, cos runs in double, the product cos(ang) * dist is formed in double
 *     (dist promoted) and rounded ONCE when it is stored into the float field;
 *   float_overload == 1 (GCC >= 6 with the C++ <math.h> wrapper in sight: ::cos(float) exists and wins): cosf(ang) * dist in float.
 * The two differ in the last bit of about a third of the coordinates. */
/* Spec §Trig: the float sine / cosine of the float-overload reading, restated: double arithmetic only, two-word reduction by pi/2, degree-13 / -12
 * kernels, one rounding to float. (float_overload == 2 below keeps the C library's cosf / sinf: the side check that the specified functions stay within 1 ulp of them.) */
static void orc_sincosf_spec(float xf, float* s_out, float* c_out) {
    const double x = (double)xf;
    const double fn = rint(x * 6.36619772367581382433e-01);
    const double r = (x - fn * 1.57079632673412561417e+00) - fn * 6.07710050650619224932e-11;
    const double z = r * r;
    const double sp = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sn = r + r * z * (-1.66666666666666324348e-01 + z * sp);
    const double cp = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double cs = 1.0 - (0.5 * z - z * cp);
    const long long q = (long long)fn & 3ll;
    const double sv = (q == 0) ? sn : ((q == 1) ? cs : ((q == 2) ? -sn : -cs));
    const double cv = (q == 0) ? cs : ((q == 1) ? -sn : ((q == 2) ? -cs : sn));
    if ((xf - xf) == 0.0f) { *s_out = (float)sv; *c_out = (float)cv; } else { *s_out = xf - xf; *c_out = xf - xf; }
}
void orc_sincosf(float x, float* s, float* c) { orc_sincosf_spec(x, s, c); }   /* (tests: accuracy against the C library) */

void orc_agg_add_scan2(orc_agg* a, const float* ranges, size_t n, float angle_min, float angle_increment, const double tf[7], int float_overload) {
    double m[9], o[3], q[4];
    make_tf(tf, m, o, q);
    for (size_t i = 0; i < n; i++) {
        const float ang = angle_min + (float)i * angle_increment;   /* :272 (size_t -> float, float arithmetic) */
        const float dist = ranges[i];
        float x, y;                                                 /* :281-283, z = 0 */
        if (float_overload == 1) { float sn, cs; orc_sincosf_spec(ang, &sn, &cs); x = cs * dist; y = sn * dist; }
        else if (float_overload) { x = cosf(ang) * dist; y = sinf(ang) * dist; }
        else { x = (float)(cos((double)ang) * (double)dist); y = (float)(sin((double)ang) * (double)dist); }
        add_point(a, x, y, 0.0f, m, o, q);
    }
}
void orc_agg_add_scan(orc_agg* a, const float* ranges, size_t n, float angle_min, float angle_increment, const double tf[7]) {
    orc_agg_add_scan2(a, ranges, n, angle_min, angle_increment, tf, 0);
}

size_t orc_agg_count(const orc_agg* a) { return a->n; }
void orc_agg_points(const orc_agg* a, float* out) { memcpy(out, a->pts, 16 * a->n); }
double orc_agg_angle(const orc_agg* a) { return a->current_angle; }
/* getProgress :119-124 */
double orc_agg_progress(const orc_agg* a) { return a->creating ? 0.1 * floor(a->current_angle * 1000.0 / a->angular_distance) : -1.0; }
/* isPointcloudReady :95-103 */
int orc_agg_ready(const orc_agg* a) { return a->current_angle > a->angular_distance; }
