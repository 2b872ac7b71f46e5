"""The N2b kernel (contact logic, recent-contact filters, plane fit, terrain pitch) compiled FOR THE HOST from the product's own source text: the section of
csrc/a1mpc_hip.hip between its N2b banner and the launch helper is cut out, the HIP keywords are defined away and one call of the per-robot body (contact_terrain_robot) runs one robot on its record.
Test infrastructure (the kernel is plain C++ without intrinsics): lets the CPU suite step the shipped arithmetic and state layout against the oracle."""
import ctypes as C, os, subprocess, hashlib, tempfile
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_SRC = os.path.join(_ROOT, "a1-qp-mpc-controller_amd", "csrc", "a1mpc_hip.hip")
_PRE = r'''
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <cstring>
#define __global__
#define __device__
#define __launch_bounds__(x)
#define __forceinline__ inline
struct Dim { unsigned x; };
static thread_local Dim blockIdx, threadIdx;
using std::fabs; using std::sqrt; using std::acos;
'''
_POST = r'''
extern "C" size_t n2b_bytes_per_robot() { return kCtBytesPerRobot; }
extern "C" void n2b_tick(int n, int max_batch, void* state, double counter_per_swing, double foot_force_low, int use_terrain_adapt, const double* gc, const uint8_t* plan,
                         const double* ff, const double* foot, const double* z, double* pitch, uint8_t* contacts, double* recent_out, double* terrain_out, const double* recent_in) {
    ContactArgs a;
    a.n = n; a.counter_per_swing = counter_per_swing; a.foot_force_low = foot_force_low; a.use_terrain_adapt = use_terrain_adapt;
    a.rec = reinterpret_cast<CtRecord*>(state); a.leg_ring = reinterpret_cast<double*>(a.rec + max_batch); a.terrain_ring = a.leg_ring + static_cast<size_t>(max_batch) * kCtLegRing; a.stride = max_batch;
    a.gait_counter = gc; a.foot_force = ff; a.foot_pos_abs = foot; a.root_pos_z = z; a.plan_contacts = plan; a.pitch_d = pitch; a.contacts = contacts;
    a.recent_out = recent_out; a.terrain_out = terrain_out; a.recent_in = recent_in; a.z_stride = 1; a.pitch_stride = 1; a.pk_tick = nullptr;
    for (int b = 0; b < n; ++b) contact_terrain_robot(a, b, a.rec + b);   // (the kernel proper adds the wavefront's LDS staging of the records around this)
}
'''


def section():
    s = open(_SRC).read()
    i0 = s.index("// ---- N2b: contact logic"); i1 = s.index("// One wavefront = 64 robots.")
    return s[i0:i1]


def load():
    src = _PRE + section() + _POST
    tag = hashlib.sha256(src.encode()).hexdigest()[:12]
    d = os.path.join(tempfile.gettempdir(), "a1mpc_n2b_host"); os.makedirs(d, exist_ok=True)
    so = os.path.join(d, f"n2b_{tag}.so")
    if not os.path.exists(so):
        cpp = os.path.join(d, f"n2b_{tag}.cpp"); open(cpp, "w").write(src)
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-fPIC", "-shared", cpp, "-o", so], check=True)
    lib = C.CDLL(so)
    lib.n2b_bytes_per_robot.restype = C.c_size_t
    return lib


class HostN2b:
    def __init__(self, max_batch):
        self.lib = load(); self.max_batch = max_batch
        self.state = np.zeros(self.lib.n2b_bytes_per_robot() * max_batch // 8 + 16)   # 128-byte aligned view below
        off = (-self.state.ctypes.data) % 128 // 8
        self.base = self.state[off:]
        assert self.base.ctypes.data % 128 == 0

    def tick(self, gc, plan, ff, foot, z, pitch, counter_per_swing=120.0, foot_force_low=30.0, use_terrain_adapt=1, recent_in=None):
        n = len(z)
        f = lambda v: np.ascontiguousarray(v, dtype=np.float64)
        gc, ff, foot, z = f(gc), f(ff), f(foot), f(z); pitch = f(pitch).copy(); plan = np.ascontiguousarray(plan, dtype=np.uint8)
        ct = np.zeros((n, 4), np.uint8); rec = np.zeros((n, 12)); ta = np.zeros(n)
        p = lambda v: v.ctypes.data_as(C.c_void_p)
        ri = f(recent_in) if recent_in is not None else None
        self.lib.n2b_tick(C.c_int(n), C.c_int(self.max_batch), p(self.base), C.c_double(counter_per_swing), C.c_double(foot_force_low), C.c_int(use_terrain_adapt), p(gc), p(plan), p(ff),
                          p(foot), p(z), p(pitch), p(ct), p(rec) if ri is None else None, p(ta), p(ri) if ri is not None else None)
        return dict(contacts=ct, foot_pos_recent_contact=rec, terrain_angle=ta, root_euler_d_pitch=pitch)
