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
//
// ~aggregate_on_device:=true: the node takes the AGGREGATOR's inputs instead of its output — the rotating laser's
// sensor_msgs/LaserScan on ~rotLaserScan (default /m3d_test/rot_scan, m3d_aggregator.cpp:153,179-180) and / or its PointCloud2 on ~rotLaserPointCloud
// (:154,181-182), the tf lookup of every message into ~pointCloudFrame (:236-240, :261-265), ~request (:178, :224-229) — and does what
// m3d_aggregator.cpp:53-124,231-288 does on the device (m3dagg_*): a sweep is BORN in HBM and bucketed in place (m3dagg_take_cloud), it never
// crosses PCIe as a cloud. ~progress and ~done are published like the aggregator's (:190-192, :206-208, :215-222).
// ~mode:=slam: scan-to-scan odometry as above, every registered sweep kept as a keyframe (resident cloud + pose); each new keyframe is
// scored on the device against the older ones (m3dloop_candidates: one streaming pass over their occupancy signatures), the candidates are registered
// in ONE m3dreg_align_batch, gated (m3dloop_gate), and the accepted loop closures are published on ~loop_closure as geometry_msgs/TransformStamped
// (header.frame_id = keyframe_<target>, child_frame_id = keyframe_<source>, the transform source -> target) for a pose-graph back end.
#include <cstdio>
#include <cstring>
#include <string>

#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/TransformStamped.h>
#include <vector>
#include <ros/ros.h>
#include <sensor_msgs/LaserScan.h>
#include <sensor_msgs/PointCloud2.h>
#include <std_msgs/Bool.h>
#include <std_msgs/Float32.h>
#include <tf/transform_broadcaster.h>
#include <tf2_ros/transform_listener.h>

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
        nh_.param<std::string>("mode", mode_, "scan_to_scan");   // or "scan_to_map": register against the HBM map of all earlier sweeps; or "slam": odometry + loop closures
        nh_.param("map_leaf", map_leaf_, 0.05);
        nh_.param("map_capacity", map_capacity_, 1 << 22);
        nh_.param("aggregate_on_device", aggregate_on_device_, false);
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
        if (mode_ == "slam") {
            m3dloop_params lp;
            m3dloop_default_params(&lp);
            double radius = lp.radius, sig_leaf = lp.sig_leaf, min_overlap = lp.min_overlap;
            nh_.param("loop_radius", radius, radius);
            nh_.param("loop_sig_leaf", sig_leaf, sig_leaf);
            nh_.param("loop_min_overlap", min_overlap, min_overlap);
            nh_.param("loop_min_gap", lp.min_gap, lp.min_gap);
            nh_.param("loop_top_k", lp.top_k, lp.top_k);
            nh_.param("loop_max_keyframes", lp.max_keyframes, 1024);
            nh_.param("loop_min_corr", loop_min_corr_, 2000);
            nh_.param("loop_max_rms", loop_max_rms_, 0.05);
            lp.radius = float(radius); lp.sig_leaf = float(sig_leaf); lp.min_overlap = float(min_overlap);
            loop_top_k_ = lp.top_k;
            if (m3dloop_create(h_, &lp, &loop_) != M3DREG_OK) {
                ROS_FATAL("m3dloop_create: %s", m3dreg_last_error(h_));
                ros::shutdown();
                return;
            }
            closure_pub_ = nh_.advertise<geometry_msgs::TransformStamped>("loop_closure", 16);
        }
        for (int i = 0; i < 16; i++) { pose_[i] = (i % 5 == 0) ? 1.f : 0.f; delta_[i] = pose_[i]; }
        pose_pub_ = nh_.advertise<geometry_msgs::PoseStamped>("pose", 1);
        if (!aggregate_on_device_) {
            sub_ = nh_.subscribe(cloud_topic, 1, &Gpu6dSlamNode::onCloud, this);   // queue depth 1, like the producer
            return;
        }
        // the aggregator's own inputs and outputs (m3d_aggregator.cpp:149-186), the aggregation itself on the device
        std::string scan_topic, pc_topic;
        nh_.param<std::string>("pointCloudFrame", root_frame_, "m3d_test/m3d_link");
        nh_.param<std::string>("rotLaserScan", scan_topic, "/m3d_test/rot_scan");
        nh_.param<std::string>("rotLaserPointCloud", pc_topic, "");
        double bb[6] = { 1, -1, 1, -1, 1, -1 };   // x_up, x_down, y_up, y_down, z_up, z_down (:164-171)
        nh_.param("bb_x_up", bb[0], bb[0]); nh_.param("bb_x_down", bb[1], bb[1]);
        nh_.param("bb_y_up", bb[2], bb[2]); nh_.param("bb_y_down", bb[3], bb[3]);
        nh_.param("bb_z_up", bb[4], bb[4]); nh_.param("bb_z_down", bb[5], bb[5]);
        int agg_capacity = 1 << 21, float_trig = 0;
        nh_.param("aggregate_capacity", agg_capacity, agg_capacity);
        nh_.param("scan_trig_float_overload", float_trig, float_trig);   // which cos / sin :281-282 resolves to on the toolchain being replaced (m3dagg_set_scan_trig)
        if (m3dagg_create(h_, bb, size_t(agg_capacity), &agg_) != M3DREG_OK || m3dagg_set_scan_trig(agg_, float_trig) != M3DREG_OK || m3dagg_set_rearm(agg_, 0) != M3DREG_OK) {   // (idle after a sweep until ~request, :211)
            ROS_FATAL("m3dagg_create: %s", m3dreg_last_error(h_));
            ros::shutdown();
            return;
        }
        progress_pub_ = nh_.advertise<std_msgs::Float32>("progress", 1);
        done_pub_ = nh_.advertise<std_msgs::Bool>("done", 1);
        rqt_sub_ = nh_.subscribe("request", 1, &Gpu6dSlamNode::onRequest, this);
        if (!scan_topic.empty()) scan_sub_ = nh_.subscribe(scan_topic, 1, &Gpu6dSlamNode::onLaserScan, this);
        if (!pc_topic.empty()) pc_sub_ = nh_.subscribe(pc_topic, 1, &Gpu6dSlamNode::onLaserCloud, this);
    }
    ~Gpu6dSlamNode() {
        if (!h_) return;
        if (loop_) m3dloop_destroy(loop_);
        for (m3dreg_cloud* c : keyframes_) m3dreg_cloud_destroy(h_, c);
        if (prev_ && !loop_) m3dreg_cloud_destroy(h_, prev_);   // (slam: the previous sweep is the newest keyframe)
        if (agg_) m3dagg_destroy(agg_);
        if (map_) m3dmap_destroy(map_);
        m3dreg_destroy(h_);
    }

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
        if (cur) onSweep(cur, msg->header);
    }

    // ---- ~aggregate_on_device: m3d_aggregator.cpp:224-288 with the per-point loops on the device -------------------------------------------------
    bool lookup(const std_msgs::Header& hdr, double tf7[7]) {
        geometry_msgs::TransformStamped t;   // (identity when the lookup fails: the reference logs the exception and goes on with a default transform, :236-241)
        try {
            tf_buffer_.canTransform(root_frame_, hdr.frame_id, hdr.stamp, ros::Duration(0.2));
            t = tf_buffer_.lookupTransform(root_frame_, hdr.frame_id, hdr.stamp);
        } catch (tf2::TransformException& ex) { ROS_WARN("%s", ex.what()); }
        tf7[0] = t.transform.translation.x; tf7[1] = t.transform.translation.y; tf7[2] = t.transform.translation.z;
        tf7[3] = t.transform.rotation.x; tf7[4] = t.transform.rotation.y; tf7[5] = t.transform.rotation.z; tf7[6] = t.transform.rotation.w;
        return true;
    }
    void onRequest(const std_msgs::BoolConstPtr& req) { if (req->data && m3dagg_restart(agg_) != M3DREG_OK) ROS_WARN("m3dagg_restart: %s", m3dreg_last_error(h_)); }   // :224-229
    void onLaserScan(const sensor_msgs::LaserScanConstPtr& scan) {   // :256-288
        double tf7[7];
        lookup(scan->header, tf7);
        if (m3dagg_add_scan(agg_, scan->ranges.data(), scan->ranges.size(), scan->angle_min, scan->angle_increment, tf7) != M3DREG_OK) ROS_WARN("m3dagg_add_scan: %s", m3dreg_last_error(h_));
        afterMessage(scan->header);
    }
    void onLaserCloud(const sensor_msgs::PointCloud2ConstPtr& m) {   // :231-254 (pcl::PointXYZ view of the message: FLOAT32 x / y / z by name)
        double tf7[7];
        lookup(m->header, tf7);
        size_t off[3] = { 0, 4, 8 };
        for (const sensor_msgs::PointField& f : m->fields) { if (f.name == "x") off[0] = f.offset; else if (f.name == "y") off[1] = f.offset; else if (f.name == "z") off[2] = f.offset; }
        if (m3dagg_add_cloud(agg_, m->data.data(), size_t(m->width) * m->height, m->point_step, off[0], off[1], off[2], tf7) != M3DREG_OK) ROS_WARN("m3dagg_add_cloud: %s", m3dreg_last_error(h_));
        afterMessage(m->header);
    }
    void afterMessage(const std_msgs::Header& hdr) {   // publishPointcloud (:188-222) without the publish: the sweep stays in HBM
        double progress = 0.0, angle = 0.0; int ready = 0;
        if (m3dagg_status(agg_, &progress, &ready, &angle, nullptr) != M3DREG_OK) return;   // (host-side bookkeeping only: no device round trip per message)
        std_msgs::Float32 pm; pm.data = float(progress); progress_pub_.publish(pm);       // :190-192
        std_msgs::Bool done; done.data = ready != 0;
        if (!ready) { if (progress >= 0.0) done_pub_.publish(done); return; }              // :215-221: done = false only while a cloud is being created
        done_pub_.publish(done);                                                          // :206-208: done = true right before the cloud
        m3dreg_cloud* cur = nullptr;
        if (m3dagg_take_cloud(agg_, &cur) != M3DREG_OK) { ROS_WARN("m3dagg_take_cloud: %s", m3dreg_last_error(h_)); return; }   // (also clears the aggregate, :211)
        std_msgs::Header h = hdr;
        h.frame_id = root_frame_; h.stamp = ros::Time::now();   // :203-204
        onSweep(cur, h);
    }

    // ---- one sweep, bucketed and resident, however it got there -------------------------------------------------------------------------------------
    void onSweep(m3dreg_cloud* cur, const std_msgs::Header& hdr) {
        if (map_) { onCloudMap(cur, hdr); return; }
        if (!prev_ && loop_) addKeyframe(cur, hdr);   // the first sweep: keyframe 0 at the origin
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
                publish(hdr);
                if (loop_) addKeyframe(cur, hdr);
            } else {
                ROS_WARN("registration rejected: status %d after %d iterations", st.status, st.iterations);
                if (loop_) { m3dreg_cloud_destroy(h_, cur); return; }   // slam: a sweep the odometry could not place is no keyframe; the last good one stays the target
            }
            if (!loop_) m3dreg_cloud_destroy(h_, prev_);
        }
        prev_ = cur;   // the new sweep becomes the target of the next registration (scan-to-scan odometry)
    }

    // ---- ~mode:=slam: the sweep becomes a keyframe; its loop-closure candidates are registered in one batch ---------------------------------------
    void addKeyframe(m3dreg_cloud* cur, const std_msgs::Header& hdr) {
        int32_t k = -1;
        if (m3dloop_add_keyframe(loop_, cur, pose_, nullptr, &k) != M3DREG_OK) { ROS_WARN("m3dloop_add_keyframe: %s", m3dreg_last_error(h_)); return; }
        keyframes_.push_back(cur);
        std::vector<m3dloop_candidate> cand(static_cast<size_t>(loop_top_k_));
        size_t n = 0;
        if (m3dloop_candidates(loop_, k, 1, cand.data(), cand.size(), &n) != M3DREG_OK) { ROS_WARN("m3dloop_candidates: %s", m3dreg_last_error(h_)); return; }
        if (n > cand.size()) n = cand.size();
        if (n == 0) return;
        std::vector<m3dreg_pair> pairs(n);
        std::vector<float> T(16 * n);
        std::vector<m3dreg_stats> st(n);
        std::vector<uint8_t> ok(n);
        if (m3dloop_make_pairs(loop_, cand.data(), n, pairs.data()) != M3DREG_OK || m3dreg_align_batch(h_, pairs.data(), n, T.data(), st.data()) != M3DREG_OK ||
            m3dloop_gate(cand.data(), st.data(), n, loop_min_corr_, loop_max_rms_, ok.data()) != M3DREG_OK) { ROS_WARN("loop closure batch: %s", m3dreg_last_error(h_)); return; }
        for (size_t i = 0; i < n; i++) {
            if (!ok[i]) continue;
            const float* t = &T[16 * i];
            tf::Matrix3x3 R(t[0], t[4], t[8], t[1], t[5], t[9], t[2], t[6], t[10]);
            tf::Quaternion q; R.getRotation(q);
            geometry_msgs::TransformStamped c;
            char name[32];
            c.header.stamp = hdr.stamp;
            std::snprintf(name, sizeof(name), "keyframe_%d", cand[i].target); c.header.frame_id = name;
            std::snprintf(name, sizeof(name), "keyframe_%d", cand[i].source); c.child_frame_id = name;
            c.transform.translation.x = t[12]; c.transform.translation.y = t[13]; c.transform.translation.z = t[14];
            c.transform.rotation.x = q.x(); c.transform.rotation.y = q.y(); c.transform.rotation.z = q.z(); c.transform.rotation.w = q.w();
            closure_pub_.publish(c);
        }
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
    ros::Subscriber sub_, rqt_sub_, scan_sub_, pc_sub_;
    ros::Publisher pose_pub_, progress_pub_, done_pub_, closure_pub_;
    tf::TransformBroadcaster br_;
    tf2_ros::Buffer tf_buffer_;
    tf2_ros::TransformListener tf_listener_{ tf_buffer_ };
    std::string odom_frame_, root_frame_;
    m3dreg_handle* h_ = nullptr;
    m3dreg_cloud* prev_ = nullptr;
    m3dmap* map_ = nullptr;
    m3dagg* agg_ = nullptr;
    m3dloop* loop_ = nullptr;
    std::vector<m3dreg_cloud*> keyframes_;   // slam: the keyframes' clouds stay resident (they are the loop closures' sources and targets)
    int loop_top_k_ = 2, loop_min_corr_ = 2000;
    double loop_max_rms_ = 0.05;
    bool aggregate_on_device_ = false;
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
