// a1mpc_build_id.cpp -- the identity of a liba1mpc.so: the hash of the sources it was compiled from (build.py source_hash(), passed as A1MPC_SOURCE_HASH).
// a1mpc_build_info() returns this string; build.py reads it from the file's bytes to decide whether the shipped library is the compilation of the sources beside it.
#ifndef A1MPC_SOURCE_HASH
#define A1MPC_SOURCE_HASH "unknown"
#endif
extern "C" const char a1mpc_build_id_[] = "sources " A1MPC_SOURCE_HASH " arch gfx950";
