// gpu_6dslam_node.cpp — SOURCE-ONLY ROS1 shim (catkin package `gpu_6dslam`, executable `gpu_6dslam_node`).
//
// This is the node the reference launches at
//   /root/reference/m3d/m3d_husky_launch/launch/m3d_husky_bringup.launch:13
//     <node pkg="gpu_6dslam" type="gpu_6dslam_node" name="gpu_6dslam_node" />
// and whose own source is absent from the reference (gpu_6dslam/ is an empty, un-vendored submodule).
// It cannot be compiled or run in this repository's environments (no ROS, no PCL, no Eigen anywhere:
// SURVEY.md §0 F7); it is kept short on purpose — everything of substance happens behind the C ABI of
// include/m3dreg.h — and is reviewed, not executed. Build it inside a catkin workspace with
//   find_package(catkin REQUIRED COMPONENTS roscpp sensor_msgs geometry_msgs tf)
//   target_link_libraries(gpu_6dslam_node ${catkin_LIBRARIES} m3dreg)
//
// Topic contract (what the producer publishes, all under the aggregator's private namespace):
//   ~cloud    sensor_msgs/PointCloud2, queue 1   m3d/m3d_aggregator/src/m3d_aggregator.cpp:174,209
//   ~done     std_msgs/Bool, published right BEFORE each cloud          :206-208
//   ~progress std_msgs/Float32                                          :190-192
// resolved as /m3d_test/aggregator/{cloud,done,progress} (universal.launch:15,45; the joystick remap
// in m3d_husky_bringup.launch:10 confirms the /m3d_test/aggregator prefix). The message is
// pcl::toPCLPointCloud2 of pcl::PointXYZ (:196-201): point_step 16, FLOAT32 x@0 y@4 z@8, unorganised,
// frame_id = pointCloudFrame (default m3d_test/m3d_link, :152,:203).
#include <cstring>
#include <string>

#include <geometry_msgs/PoseStamped.h>
#include <vector>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>

#include "m3dreg.h"

class Gpu6dSlamNode {
public:
    Gpu6dSlamNode() : nh_("~") {
        std::string cloud_topic;
        nh_.param<std::string>("cloud", cloud_topic, "/m3d_test/aggregator/cloud");
        nh_.param<std::string>("odom_frame", odom_frame_, "m3d_test/odom");
        m3dreg_params p;
        m3dreg_default_params(&p);
        // the library's defaults are a two-level pyramid (0.4 m -> 0.1 m); ~leaf / ~max_corr_dist / ~iterations address the FINEST level,
        // ~coarse_leaf / ~coarse_max_corr_dist / ~coarse_iterations the one before it (~coarse_leaf:=0 registers on the fine level alone)
        const int fine = p.n_levels - 1;
        double leaf = p.leaf[fine], dmax = p.max_corr_dist[fine], normal_leaf = p.normal_leaf;
        double coarse_leaf = fine > 0 ? p.leaf[0] : 0.0, coarse_dmax = fine > 0 ? p.max_corr_dist[0] : 0.0;
        int iters = p.iterations[fine], coarse_iters = fine > 0 ? p.iterations[0] : 0, device = 0;
        nh_.param("leaf", leaf, leaf);
        nh_.param("max_corr_dist", dmax, dmax);
        nh_.param("normal_leaf", normal_leaf, normal_leaf);
        nh_.param("iterations", iters, iters);
        nh_.param("coarse_leaf", coarse_leaf, coarse_leaf);
        nh_.param("coarse_max_corr_dist", coarse_dmax, coarse_dmax);
        nh_.param("coarse_iterations", coarse_iters, coarse_iters);
        nh_.param("device", device, device);
        nh_.param<std::string>("mode", mode_, "scan_to_scan");   // or "scan_to_map": register against the HBM map of all earlier sweeps
        nh_.param("map_leaf", map_leaf_, 0.05);
        nh_.param("map_capacity", map_capacity_, 1 << 22);
        if (coarse_leaf > 0.0) {
            p.n_levels = 2;
            p.leaf[0] = float(coarse_leaf); p.max_corr_dist[0] = float(coarse_dmax); p.iterations[0] = coarse_iters;
            p.leaf[1] = float(leaf); p.max_corr_dist[1] = float(dmax); p.iterations[1] = iters;
        } else {
            p.n_levels = 1;
            p.leaf[0] = float(leaf); p.max_corr_dist[0] = float(dmax); p.iterations[0] = iters;
        }
        p.normal_leaf = float(normal_leaf);
        int rc = m3dreg_create(&p, device, nullptr, &h_);
        if (rc != M3DREG_OK) {   // same policy as the reference's drivers: fatal + exit (encoder_node_li.cpp:60-80)
            ROS_FATAL("m3dreg_create failed (%d): no usable MI355X; there is no CPU fallback", rc);
            ros::shutdown();
            return;
        }
        m3dreg_set_latency_mode(h_, 1);   // ONE spin thread, one registration per sweep: this handle's work has the GPU to itself (ABI 7)
        if (mode_ == "scan_to_map" && m3dmap_create(h_, float(map_leaf_), size_t(map_capacity_), &map_) != M3DREG_OK) {
            ROS_FATAL("m3dmap_create: %s", m3dreg_last_error(h_));
            ros::shutdown();
            return;
        }
        for (int i = 0; i < 16; i++) { pose_[i] = (i % 5 == 0) ? 1.f : 0.f; delta_[i] = pose_[i]; }
        pose_pub_ = nh_.advertise<geometry_msgs::PoseStamped>("pose", 1);
        sub_ = nh_.subscribe(cloud_topic, 1, &Gpu6dSlamNode::onCloud, this);   // queue depth 1, like the producer
    }
    ~Gpu6dSlamNode() { if (h_) { if (prev_) m3dreg_cloud_destroy(h_, prev_); if (map_) m3dmap_destroy(map_); m3dreg_destroy(h_); } }

private:
    // The message crosses the C ABI as it is — raw buffer + field table; x/y/z are resolved by name on the other side the
    // way pcl::fromPCLPointCloud2 would (any offsets, FLOAT32/FLOAT64, either byte order, padded rows: SURVEY §8 row f3).
    // Each sweep is bucketed ONCE: it is the source of this registration and the target of the next.
    m3dreg_cloud* bucket(const sensor_msgs::PointCloud2& m) {
        std::vector<m3dreg_point_field> ft(m.fields.size());
        for (size_t i = 0; i < m.fields.size(); i++)
            ft[i] = m3dreg_point_field{ m.fields[i].name.c_str(), m.fields[i].offset, m.fields[i].datatype, m.fields[i].count };
        m3dreg_cloud* c = nullptr;
        const int rc = m3dreg_cloud_create_pc2(h_, m.data.data(), m.data.size(), m.width, m.height, m.point_step, m.row_step, ft.data(), ft.size(),
                                               m.is_bigendian ? 1 : 0, map_ ? M3DREG_CLOUD_SOURCE_ONLY : 0, &c);   // scan-to-map: a sweep is only a source and a map insert
        if (rc != M3DREG_OK) ROS_WARN("m3dreg_cloud_create_pc2: %s", m3dreg_last_error(h_));   // logged and swallowed, m3d_aggregator.cpp:239-241
        return rc == M3DREG_OK ? c : nullptr;
    }

    void onCloud(const sensor_msgs::PointCloud2ConstPtr& msg) {
        m3dreg_cloud* cur = bucket(*msg);
        if (!cur) return;
        if (map_) { onCloudMap(cur, msg->header); return; }
        if (prev_) {
            float T[16]; m3dreg_stats st;
            const int rc = m3dreg_align_clouds(h_, cur, prev_, delta_, T, &st);
            if (rc != M3DREG_OK) { ROS_WARN("m3dreg_align_clouds: %s", m3dreg_last_error(h_)); }
            else if (st.status == M3DREG_CONVERGED || st.status == M3DREG_MAX_ITERATIONS) {
                std::memcpy(delta_, T, sizeof(T));   // constant-velocity prior for the next sweep
                float P[16];                          // pose <- pose * T  (column-major 4x4, Eigen::Matrix4f layout)
                for (int c = 0; c < 4; c++)
                    for (int r = 0; r < 4; r++) {
                        float s = 0.f;
                        for (int k = 0; k < 4; k++) s += pose_[k * 4 + r] * T[c * 4 + k];
                        P[c * 4 + r] = s;
                    }
                std::memcpy(pose_, P, sizeof(P));
                publish(msg->header);
            } else {
                ROS_WARN("registration rejected: status %d after %d iterations", st.status, st.iterations);
            }
            m3dreg_cloud_destroy(h_, prev_);
        }
        prev_ = cur;   // the new sweep becomes the target of the next registration (scan-to-scan odometry)
    }

    static void mul4(const float* A, const float* B, float* out) {   // out = A * B, column-major 4x4 (Eigen::Matrix4f layout)
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) {
                float s = 0.f;
                for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
                out[c * 4 + r] = s;
            }
    }
    static void inv_rigid(const float* T, float* out) {   // [R t]^-1 = [R^T  -R^T t]
        for (int i = 0; i < 16; i++) out[i] = (i % 5 == 0) ? 1.f : 0.f;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) out[c * 4 + r] = T[r * 4 + c];
        for (int r = 0; r < 3; r++) out[12 + r] = -(out[r] * T[12] + out[4 + r] * T[13] + out[8 + r] * T[14]);
    }

    // scan-to-map: the sweep is registered against the voxel-deduplicated map of all earlier sweeps, which
    // never leaves HBM, then inserted with its pose. Prior: constant velocity in the map frame.
    void onCloudMap(m3dreg_cloud* cur, const std_msgs::Header& hdr) {
        size_t n_map = 0, added = 0;
        m3dmap_size(map_, &n_map);
        if (n_map == 0) {
            if (m3dmap_insert(map_, cur, pose_, &added) != M3DREG_OK) ROS_WARN("m3dmap_insert: %s", m3dreg_last_error(h_));
            m3dreg_cloud_destroy(h_, cur);
            return;
        }
        m3dreg_cloud* tgt = nullptr;
        if (m3dmap_as_cloud(map_, &tgt) != M3DREG_OK) { ROS_WARN("m3dmap_as_cloud: %s", m3dreg_last_error(h_)); m3dreg_cloud_destroy(h_, cur); return; }
        float prior[16], T[16]; m3dreg_stats st;
        mul4(pose_, delta_, prior);
        const int rc = m3dreg_align_clouds(h_, cur, tgt, prior, T, &st);
        if (rc != M3DREG_OK) { ROS_WARN("m3dreg_align_clouds: %s", m3dreg_last_error(h_)); }
        else if (st.status == M3DREG_CONVERGED || st.status == M3DREG_MAX_ITERATIONS) {
            float inv[16];
            inv_rigid(pose_, inv);
            mul4(inv, T, delta_);
            std::memcpy(pose_, T, sizeof(T));
            if (m3dmap_insert(map_, cur, T, &added) != M3DREG_OK) ROS_WARN("m3dmap_insert: %s", m3dreg_last_error(h_));   // map full: keep localising
            publish(hdr);
        } else {
            ROS_WARN("registration rejected: status %d after %d iterations", st.status, st.iterations);
        }
        m3dreg_cloud_destroy(h_, tgt);
        m3dreg_cloud_destroy(h_, cur);
    }

    void publish(const std_msgs::Header& hdr) {
        tf::Matrix3x3 R(pose_[0], pose_[4], pose_[8], pose_[1], pose_[5], pose_[9], pose_[2], pose_[6], pose_[10]);
        tf::Transform t(R, tf::Vector3(pose_[12], pose_[13], pose_[14]));
        br_.sendTransform(tf::StampedTransform(t, hdr.stamp, odom_frame_, hdr.frame_id));
        geometry_msgs::PoseStamped ps;
        ps.header.stamp = hdr.stamp; ps.header.frame_id = odom_frame_;
        tf::Quaternion q; R.getRotation(q);
        ps.pose.position.x = pose_[12]; ps.pose.position.y = pose_[13]; ps.pose.position.z = pose_[14];
        ps.pose.orientation.x = q.x(); ps.pose.orientation.y = q.y(); ps.pose.orientation.z = q.z(); ps.pose.orientation.w = q.w();
        pose_pub_.publish(ps);
    }

    ros::NodeHandle nh_;
    ros::Subscriber sub_;
    ros::Publisher pose_pub_;
    tf::TransformBroadcaster br_;
    std::string odom_frame_;
    m3dreg_handle* h_ = nullptr;
    m3dreg_cloud* prev_ = nullptr;
    m3dmap* map_ = nullptr;
    std::string mode_;
    double map_leaf_ = 0.05;
    int map_capacity_ = 1 << 22;
    float pose_[16], delta_[16];
};

int main(int argc, char** argv) {
    ros::init(argc, argv, "gpu_6dslam_node");
    Gpu6dSlamNode node;
    ros::spin();   // single-threaded, callbacks serialised — same threading model as the producer (m3d_aggregator.cpp:185)
    return 0;
}
