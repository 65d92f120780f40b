#pragma once
#include <cmath>
#include <string>
#include <std_msgs/Header.h>
namespace tf {
struct Vector3 { double v[3]; Vector3(double x = 0, double y = 0, double z = 0) : v{ x, y, z } {} };
struct Quaternion { double q[4] = { 0, 0, 0, 1 }; double x() const { return q[0]; } double y() const { return q[1]; } double z() const { return q[2]; } double w() const { return q[3]; } };
struct Matrix3x3 {
    double m[3][3];
    Matrix3x3(double xx, double xy, double xz, double yx, double yy, double yz, double zx, double zy, double zz) : m{ { xx, xy, xz }, { yx, yy, yz }, { zx, zy, zz } } {}
    void getRotation(Quaternion& o) const {   // Shepperd's method, like tf::Matrix3x3::getRotation
        const double tr = m[0][0] + m[1][1] + m[2][2];
        if (tr > 0) { double s = std::sqrt(tr + 1.0); o.q[3] = 0.5 * s; s = 0.5 / s; o.q[0] = (m[2][1] - m[1][2]) * s; o.q[1] = (m[0][2] - m[2][0]) * s; o.q[2] = (m[1][0] - m[0][1]) * s; }
        else { int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0); int j = (i + 1) % 3, k = (i + 2) % 3;
               double s = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0); o.q[i] = 0.5 * s; s = 0.5 / s; o.q[3] = (m[k][j] - m[j][k]) * s; o.q[j] = (m[j][i] + m[i][j]) * s; o.q[k] = (m[k][i] + m[i][k]) * s; }
    }
};
struct Transform { Matrix3x3 basis; Vector3 origin; Transform(const Matrix3x3& b, const Vector3& o) : basis(b), origin(o) {} };
struct StampedTransform : Transform { ros::Time stamp; std::string frame_id, child_frame_id;
    StampedTransform(const Transform& t, const ros::Time& s, const std::string& f, const std::string& c) : Transform(t), stamp(s), frame_id(f), child_frame_id(c) {} };
struct TransformBroadcaster { int sent = 0; void sendTransform(const StampedTransform&) { sent++; } };
}
