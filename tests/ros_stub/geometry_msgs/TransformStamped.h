#pragma once
#include <string>
#include <std_msgs/Header.h>
#include <geometry_msgs/PoseStamped.h>
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::Header header; std::string child_frame_id; Transform transform; };
}
