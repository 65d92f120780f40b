#pragma once
// see README.md: a stand-in for <ros/ros.h> that exists so that ros/gpu_6dslam_node.cpp can be compiled, linked and driven in tests
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/TransformStamped.h>
#include <sensor_msgs/LaserScan.h>
#include <sensor_msgs/PointCloud2.h>
#include <std_msgs/Bool.h>
#include <std_msgs/Float32.h>
#define ROS_STUB_LOG(level, ...) do { std::fprintf(stderr, "[" level "] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_FATAL(...) ROS_STUB_LOG("FATAL", __VA_ARGS__)
#define ROS_WARN(...) ROS_STUB_LOG("WARN", __VA_ARGS__)
#define ROS_INFO(...) ROS_STUB_LOG("INFO", __VA_ARGS__)
namespace ros {
namespace stub {
inline std::function<void(const sensor_msgs::PointCloud2ConstPtr&)>& callback() { static std::function<void(const sensor_msgs::PointCloud2ConstPtr&)> f; return f; }
inline std::function<void(const sensor_msgs::LaserScanConstPtr&)>& scan_callback() { static std::function<void(const sensor_msgs::LaserScanConstPtr&)> f; return f; }
inline std::function<void(const std_msgs::BoolConstPtr&)>& bool_callback() { static std::function<void(const std_msgs::BoolConstPtr&)> f; return f; }
inline double* current_tf() { static double v[7] = { 0, 0, 0, 0, 0, 0, 1 }; return v; }   // the transform of the message being played (tf2_ros::Buffer::lookupTransform)
inline bool& have_tf() { static bool h = false; return h; }
inline bool& down() { static bool d = false; return d; }
inline std::map<std::string, std::string>& params() { static std::map<std::string, std::string> p; return p; }   // from M3D_STUB_PARAMS="leaf=0.1;iterations=20"
}
inline void init(int&, char**, const std::string&) {
    if (const char* v = std::getenv("M3D_STUB_PARAMS")) { std::stringstream ss(v); std::string kv; while (std::getline(ss, kv, ';')) { const size_t e = kv.find('='); if (e != std::string::npos) stub::params()[kv.substr(0, e)] = kv.substr(e + 1); } }
}
inline void shutdown() { stub::down() = true; }
inline bool ok() { return !stub::down(); }
struct Duration { double s; explicit Duration(double v = 0.0) : s(v) {} };
struct Publisher {
    std::string topic;
    template <class M> void publish(const M&) const {}
    void publish(const std_msgs::Bool& b) const { if (b.data) std::printf("done 1\n"); }
    void publish(const geometry_msgs::TransformStamped& c) const {
        std::printf("closure %s %s %.9g %.9g %.9g  %.9g %.9g %.9g %.9g\n", c.child_frame_id.c_str(), c.header.frame_id.c_str(), c.transform.translation.x, c.transform.translation.y,
                    c.transform.translation.z, c.transform.rotation.x, c.transform.rotation.y, c.transform.rotation.z, c.transform.rotation.w);
    }
    void publish(const geometry_msgs::PoseStamped& p) const {
        std::printf("pose %.9g %.9g %.9g  %.9g %.9g %.9g %.9g\n", p.pose.position.x, p.pose.position.y, p.pose.position.z, p.pose.orientation.x, p.pose.orientation.y, p.pose.orientation.z, p.pose.orientation.w);
    }
};
struct Subscriber {};
struct NodeHandle {
    explicit NodeHandle(const std::string& = std::string()) {}
    template <class T> bool param(const std::string& name, T& var, const T& def) const {
        auto it = stub::params().find(name);
        if (it == stub::params().end()) { var = def; return false; }
        std::stringstream ss(it->second); ss >> var; return true;
    }
    template <class M> Publisher advertise(const std::string& topic, uint32_t) { return Publisher{ topic }; }
    template <class C> Subscriber subscribe(const std::string&, uint32_t, void (C::*fn)(const sensor_msgs::PointCloud2ConstPtr&), C* obj) {
        stub::callback() = [obj, fn](const sensor_msgs::PointCloud2ConstPtr& m) { (obj->*fn)(m); };
        return Subscriber();
    }
    template <class C> Subscriber subscribe(const std::string&, uint32_t, void (C::*fn)(const sensor_msgs::LaserScanConstPtr&), C* obj) {
        stub::scan_callback() = [obj, fn](const sensor_msgs::LaserScanConstPtr& m) { (obj->*fn)(m); };
        return Subscriber();
    }
    template <class C> Subscriber subscribe(const std::string&, uint32_t, void (C::*fn)(const std_msgs::BoolConstPtr&), C* obj) {
        stub::bool_callback() = [obj, fn](const std_msgs::BoolConstPtr& m) { (obj->*fn)(m); };
        return Subscriber();
    }
};
// "spin": play the clouds of M3D_STUB_CLOUDS (colon-separated files of raw float32 x y z triples) into the subscriber, each as the
// PointCloud2 of pcl::PointXYZ that m3d_aggregator.cpp:196-209 publishes: point_step 16, FLOAT32 x@0 y@4 z@8, unorganised
// M3D_STUB_SCANS (one file): LaserScan messages for the ~aggregate_on_device path, each record = uint32 n, float angle_min, float angle_increment, 7 doubles
// {tx ty tz qx qy qz qw} (what tf2_ros::Buffer::lookupTransform answers for this message), n float ranges
inline void spin_scans() {
    const char* v = std::getenv("M3D_STUB_SCANS");
    if (!v || stub::down() || !stub::scan_callback()) return;
    FILE* f = std::fopen(v, "rb");
    if (!f) { ROS_WARN("stub: cannot open %s", v); return; }
    uint32_t n = 0, seq = 0;
    while (std::fread(&n, 4, 1, f) == 1) {
        if (n == 0xFFFFFFFFu) { auto b = std::make_shared<std_msgs::Bool>(); b->data = true; if (stub::bool_callback()) stub::bool_callback()(b); continue; }   // a ~request message
        auto m = std::make_shared<sensor_msgs::LaserScan>();
        if (std::fread(&m->angle_min, 4, 1, f) != 1 || std::fread(&m->angle_increment, 4, 1, f) != 1 || std::fread(stub::current_tf(), 8, 7, f) != 7) break;
        m->ranges.resize(n);
        if (std::fread(m->ranges.data(), 4, n, f) != n) break;
        m->header.seq = seq++; m->header.frame_id = "m3d_test/laser";
        stub::have_tf() = true;
        stub::scan_callback()(m);
    }
    std::fclose(f);
}
inline void spin() {
    spin_scans();
    const char* v = std::getenv("M3D_STUB_CLOUDS");
    if (!v || stub::down() || !stub::callback()) return;
    FILE* tff = std::getenv("M3D_STUB_TF") ? std::fopen(std::getenv("M3D_STUB_TF"), "rb") : nullptr;   // optional: 7 doubles per cloud message
    std::stringstream ss(v); std::string path; uint32_t seq = 0;
    while (std::getline(ss, path, ':')) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) { ROS_WARN("stub: cannot open %s", path.c_str()); continue; }
        std::fseek(f, 0, SEEK_END); const long bytes = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        std::vector<float> xyz(size_t(bytes) / 4);
        if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size()) { std::fclose(f); continue; }
        std::fclose(f);
        auto m = std::make_shared<sensor_msgs::PointCloud2>();
        const uint32_t n = uint32_t(xyz.size() / 3);
        m->header.seq = seq++; m->header.frame_id = "m3d_test/m3d_link"; m->height = 1; m->width = n; m->point_step = 16; m->row_step = 16 * n; m->is_dense = true;
        const char* names[3] = { "x", "y", "z" };
        for (int a = 0; a < 3; a++) { sensor_msgs::PointField pf; pf.name = names[a]; pf.offset = 4 * a; pf.datatype = sensor_msgs::PointField::FLOAT32; pf.count = 1; m->fields.push_back(pf); }
        m->data.assign(size_t(16) * n, 0);
        for (uint32_t i = 0; i < n; i++) std::memcpy(&m->data[size_t(16) * i], &xyz[size_t(3) * i], 12);
        if (tff) stub::have_tf() = std::fread(stub::current_tf(), 8, 7, tff) == 7;
        stub::callback()(m);
    }
    if (tff) std::fclose(tff);
}
}  // namespace ros
