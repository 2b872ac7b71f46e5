// TEST INFRASTRUCTURE: forwards "../A1CtrlStates.h" of S/test/test_mpc.cpp to the reference's header (A1MPC_REF = its directory)
#pragma once
#define A1MPC_STR2(x) #x
#define A1MPC_STR(x) A1MPC_STR2(x)
#include A1MPC_STR(A1MPC_REF/A1CtrlStates.h)
