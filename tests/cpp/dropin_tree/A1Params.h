// TEST INFRASTRUCTURE: forwards "../A1Params.h" of S/test/test_mpc.cpp to the reference's header (A1MPC_REF = its directory)
#pragma once
#ifndef A1MPC_STR
#define A1MPC_STR2(x) #x
#define A1MPC_STR(x) A1MPC_STR2(x)
#endif
#include A1MPC_STR(A1MPC_REF/A1Params.h)
