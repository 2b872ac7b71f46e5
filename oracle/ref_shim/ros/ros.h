// TEST INFRASTRUCTURE (oracle/_ref build): the few ROS names the reference's controller sources mention, as no-ops.
// ROS is not installed here; nothing on the hot path goes through it (parameters keep their coded defaults, publishers drop messages).
#pragma once
#include <math.h>    // the real ROS headers pull in the C headers: unqualified isnan() / abs(double) in the
#include <stdlib.h>  // reference (S/A1RobotControl.cpp:315,559; S/utils/Utils.cpp:57) resolve through them
#include <string>
namespace ros {
struct Duration {};
struct Time { static Time now() { return Time(); } };
struct Publisher { template <class M> void publish(const M &) const {} };
struct NodeHandle {
    // bool param(name, out, default): nothing is on the parameter server => the default is taken
    template <class T> bool param(const std::string &, T &out, const T &def) const { out = def; return false; }
    // T param(name, default): the returning overload (the reference discards its result, S/A1RobotControl.cpp:63)
    template <class T> T param(const std::string &, const T &def) const { return def; }
    template <class M> Publisher advertise(const std::string &, int) { return Publisher(); }
};
}  // namespace ros
