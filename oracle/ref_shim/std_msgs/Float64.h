// TEST INFRASTRUCTURE (oracle/_ref build)
#pragma once
namespace std_msgs { struct Float64 { double data = 0; }; }
