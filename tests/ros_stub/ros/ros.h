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
#include <sensor_msgs/PointCloud2.h>
#define ROS_STUB_LOG(level, ...) do { std::fprintf(stderr, "[" level "] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_FATAL(...) ROS_STUB_LOG("FATAL", __VA_ARGS__)
#define ROS_WARN(...) ROS_STUB_LOG("WARN", __VA_ARGS__)
#define ROS_INFO(...) ROS_STUB_LOG("INFO", __VA_ARGS__)
namespace ros {
namespace stub {
inline std::function<void(const sensor_msgs::PointCloud2ConstPtr&)>& callback() { static std::function<void(const sensor_msgs::PointCloud2ConstPtr&)> f; return f; }
inline bool& down() { static bool d = false; return d; }
inline std::map<std::string, std::string>& params() { static std::map<std::string, std::string> p; return p; }   // from M3D_STUB_PARAMS="leaf=0.1;iterations=20"
}
inline void init(int&, char**, const std::string&) {
    if (const char* v = std::getenv("M3D_STUB_PARAMS")) { std::stringstream ss(v); std::string kv; while (std::getline(ss, kv, ';')) { const size_t e = kv.find('='); if (e != std::string::npos) stub::params()[kv.substr(0, e)] = kv.substr(e + 1); } }
}
inline void shutdown() { stub::down() = true; }
inline bool ok() { return !stub::down(); }
struct Publisher {
    std::string topic;
    template <class M> void publish(const M&) const {}
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
};
// "spin": play the clouds of M3D_STUB_CLOUDS (colon-separated files of raw float32 x y z triples) into the subscriber, each as the
// PointCloud2 of pcl::PointXYZ that m3d_aggregator.cpp:196-209 publishes: point_step 16, FLOAT32 x@0 y@4 z@8, unorganised
inline void spin() {
    const char* v = std::getenv("M3D_STUB_CLOUDS");
    if (!v || stub::down() || !stub::callback()) return;
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
        stub::callback()(m);
    }
}
}  // namespace ros
