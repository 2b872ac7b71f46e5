// tests/cpp/dropin_tree/ConvexMpc.h -- TEST INFRASTRUCTURE: stands where S/ConvexMpc.h stands relative to S/test/test_mpc.cpp, so that the
// reference's only driver of the hot path compiles, UNMODIFIED (fed to the compiler through stdin from /root/reference), against the drop-in
// class instead of the reference's own: same includes as S/ConvexMpc.h:10-20, then ConvexMpc := a1mpc::ConvexMpcGpu<PLAN_HORIZON>.
#pragma once
#define EIGEN_STACK_ALLOCATION_LIMIT 0
#include <vector>
#include <chrono>
#include "OsqpEigen/OsqpEigen.h"
#include <Eigen/Dense>
#include "A1CtrlStates.h"
#include "A1Params.h"
#include "utils/Utils.h"
#include "a1mpc_dropin.hpp"
typedef a1mpc::ConvexMpcGpu<PLAN_HORIZON> ConvexMpc;
