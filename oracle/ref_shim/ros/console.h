// TEST INFRASTRUCTURE (oracle/_ref build)
#pragma once
#include "ros/ros.h"
