#pragma once
// stand-in for tf2_ros::Buffer / TransformListener: lookupTransform returns the transform the stub's ros::spin() attached to the message being played
#include <stdexcept>
#include <string>
#include <geometry_msgs/TransformStamped.h>
#include <ros/ros.h>
namespace tf2 { struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; }; }
namespace tf2_ros {
struct Buffer {
    bool canTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&) const { return true; }
    geometry_msgs::TransformStamped lookupTransform(const std::string& target, const std::string& source, const ros::Time&) const {
        if (!ros::stub::have_tf()) throw tf2::TransformException("stub: no transform from " + source + " to " + target);
        geometry_msgs::TransformStamped t;
        const double* v = ros::stub::current_tf();
        t.header.frame_id = target; t.child_frame_id = source;
        t.transform.translation.x = v[0]; t.transform.translation.y = v[1]; t.transform.translation.z = v[2];
        t.transform.rotation.x = v[3]; t.transform.rotation.y = v[4]; t.transform.rotation.z = v[5]; t.transform.rotation.w = v[6];
        return t;
    }
};
struct TransformListener { explicit TransformListener(Buffer&) {} };
}
