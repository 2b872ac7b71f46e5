// TEST INFRASTRUCTURE (oracle/_ref build): field names of the RViz marker message the reference fills for debugging.
#pragma once
#include <string>
#include <vector>
#include "ros/ros.h"
namespace std_msgs { struct Header { std::string frame_id; ros::Time stamp; }; struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; }
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
}
namespace visualization_msgs {
struct Marker {
    enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4 };
    enum { ADD = 0, MODIFY = 0, DELETE = 2 };
    std_msgs::Header header; std::string ns; int id = 0, type = 0, action = 0;
    geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color; ros::Duration lifetime;
    std::vector<geometry_msgs::Point> points; std::vector<std_msgs::ColorRGBA> colors;
};
}
