#pragma once
#include <memory>
#include <vector>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct LaserScan {
    std_msgs::Header header; float angle_min = 0, angle_max = 0, angle_increment = 0, time_increment = 0, scan_time = 0, range_min = 0, range_max = 0;
    std::vector<float> ranges, intensities;
};
typedef std::shared_ptr<LaserScan> LaserScanPtr;
typedef std::shared_ptr<const LaserScan> LaserScanConstPtr;
}
