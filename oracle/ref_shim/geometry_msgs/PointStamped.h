// TEST INFRASTRUCTURE (oracle/_ref build)
#pragma once
#include "visualization_msgs/Marker.h"
namespace geometry_msgs { struct PointStamped { std_msgs::Header header; Point point; }; }
