"""ctypes wrapper around oracle/_ref/liba1ref*.so  --  TEST INFRASTRUCTURE ONLY.

oracle/_ref holds the REFERENCE's own sources (S/ConvexMpc.cpp, S/A1RobotControl.cpp, S/A1BasicEKF.cpp, S/utils/Utils.cpp,
S/utils/filter.hpp, S/legKinematics/A1Kinematics.cpp, S/test/test_mpc.cpp) compiled verbatim from /root/reference against the
stand-in headers of oracle/ref_shim/ (see oracle/Makefile `ref`, oracle/ref_harness.cpp).  It exists to PIN oracle/a1mpc_oracle.c;
only tests/ may import this module.  /root/reference is needed to (re)build, not to load a prebuilt library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_SRC = "/root/reference/src/a1_cpp/src"
_libs = {}


def lib_path(horizon=10):
    return os.path.join(_HERE, "_ref", "liba1ref.so" if horizon == 10 else f"liba1ref_h{horizon}.so")


def build():
    """(Re)build oracle/_ref from the reference sources when they are present; returns True when the libraries exist."""
    if os.path.isdir(REFERENCE_SRC):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return all(os.path.exists(lib_path(h)) for h in (10, 16, 20))


def lib(horizon=10):
    if horizon not in _libs:
        L = C.CDLL(lib_path(horizon))
        L.ref_ctx_new.restype = C.c_void_p
        L.ref_dihedral_angle.restype = C.c_double
        assert L.ref_plan_horizon() == horizon
        _libs[horizon] = L
    return _libs[horizon]


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _a(v):
    return np.ascontiguousarray(v, dtype=np.float64)


def colmajor(M):
    """(r,c) array -> Eigen storage order"""
    return np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).ravel()


def convex_mpc_form(horizon, q, r, euler, mass, inertia_rowmajor, R_rowmajor, foot, contacts, x0, xref, dt, foot_stride=0, want_AB=False):
    """ConvexMpc driven as S/A1RobotControl.cpp:447-518 does; foot is 3x4 column-major (x foot_stride per step)."""
    L = lib(horizon)
    n, m = 12 * horizon, 20 * horizon
    P = np.zeros((n, n)); g = np.zeros(n); A = np.zeros((m, n)); l = np.zeros(m); u = np.zeros(m)
    Aqp = np.zeros((13 * horizon, 13)) if want_AB else None
    Bqp = np.zeros((13 * horizon, n)) if want_AB else None
    I = colmajor(np.asarray(inertia_rowmajor, float).reshape(3, 3)); R = colmajor(np.asarray(R_rowmajor, float).reshape(3, 3))
    L.ref_convex_mpc_form(_p(_a(q)), _p(_a(r)), _p(_a(euler)), C.c_double(mass), _p(I), _p(R), _p(_a(foot)), C.c_int(foot_stride),
                          _p(np.ascontiguousarray(contacts, dtype=np.uint8), C.c_uint8), _p(_a(x0)), _p(_a(xref)), C.c_double(dt),
                          _p(P), _p(g), _p(A), _p(l), _p(u), _p(Aqp), _p(Bqp))
    return dict(P=P, g=g, A=A, l=l, u=u, A_qp=Aqp, B_qp=Bqp)


def last_qp(horizon=10):
    L = lib(horizon)
    n, m = C.c_int(), C.c_int()
    solves = L.ref_last_qp_dims(C.byref(n), C.byref(m))
    n, m = n.value, m.value
    P = np.zeros((n, n)); q = np.zeros(n); A = np.zeros((m, n)); l = np.zeros(m); u = np.zeros(m); x = np.zeros(n); y = np.zeros(m)
    L.ref_last_qp(_p(P), _p(q), _p(A), _p(l), _p(u), _p(x), _p(y))
    it, st, nf = C.c_int(), C.c_int(), C.c_int(); rho = C.c_double()
    L.ref_last_info(C.byref(it), C.byref(st), C.byref(nf), C.byref(rho))
    re = C.c_int(); reinits = L.ref_last_reinit(C.byref(re))
    return dict(P=P, q=q, A=A, l=l, u=u, x=x, y=y, iters=it.value, status=st.value, nfact=nf.value, rho=rho.value, solves=solves, reinit=re.value, reinits=reinits)


def run_test_mpc(horizon=10):
    """S/test/test_mpc.cpp's main(), as written; returns (printed text, the QP it solved)."""
    buf = C.create_string_buffer(1 << 14)
    rc = lib(horizon).ref_run_test_mpc(buf, len(buf))
    assert rc == 0
    return buf.value.decode(), last_qp(horizon)


def set_base_settings(settings, horizon=10):
    """OSQP settings every OsqpEigen::Solver constructed afterwards starts from (oracle.Settings layout)."""
    lib(horizon).ref_set_base_settings(C.byref(settings))


class Controller:
    """One A1CtrlStates + A1RobotControl (+ optional A1BasicEKF) of the reference."""

    def __init__(self, horizon=10):
        self.L = lib(horizon)
        self.c = C.c_void_p(self.L.ref_ctx_new())

    def close(self):
        if self.c:
            self.L.ref_ctx_free(self.c); self.c = None

    def __del__(self):
        self.close()

    def set(self, name, value):
        v = _a(value).ravel()
        rc = self.L.ref_state_set(self.c, name.encode(), _p(v), C.c_int(v.size))
        assert rc == 0, f"no field {name} of {v.size} numbers"

    def set_mat(self, name, M):
        self.set(name, colmajor(M))

    def get(self, name, n):
        v = np.zeros(n)
        rc = self.L.ref_state_get(self.c, name.encode(), _p(v), C.c_int(n))
        assert rc == 0, f"no field {name} of {n} numbers"
        return v

    def get_mat(self, name, r, c):
        return self.get(name, r * c).reshape(c, r).T.copy()

    def compute_grf(self, dt):
        g = np.zeros(12)
        self.L.ref_compute_grf(self.c, C.c_double(dt), _p(g))
        return g    # 3x4 column-major == the C ABI's grf layout

    def update_plan(self, dt):
        self.L.ref_update_plan(self.c, C.c_double(dt))

    def generate_swing_legs_ctrl(self, dt):
        self.L.ref_generate_swing_legs_ctrl(self.c, C.c_double(dt))

    def compute_joint_torques(self):
        self.L.ref_compute_joint_torques(self.c)

    def compute_walking_surface(self):
        v = np.zeros(3); self.L.ref_compute_walking_surface(self.c, _p(v)); return v

    def ekf_new(self, assume_flat_ground=True):
        self.L.ref_ekf_new(self.c, C.c_int(int(assume_flat_ground)))

    def ekf_init_state(self):
        self.L.ref_ekf_init_state(self.c)

    def ekf_update(self, dt):
        self.L.ref_ekf_update(self.c, C.c_double(dt))


def filter_run(window, x):
    x = _a(x); out = np.zeros_like(x)
    lib().ref_filter_run(C.c_int(window), C.c_int(x.size), _p(x), _p(out))
    return out


def leg_fk(q, rho_opt, rho_fix):
    p = np.zeros(3); lib().ref_leg_fk(_p(_a(q)), _p(_a(rho_opt)), _p(_a(rho_fix)), _p(p)); return p


def leg_jac(q, rho_opt, rho_fix):
    J = np.zeros(9); lib().ref_leg_jac(_p(_a(q)), _p(_a(rho_opt)), _p(_a(rho_fix)), _p(J)); return J   # column-major


def bezier_foot_curve(t, start, fin, pitch=0.0):
    o = np.zeros(3); lib().ref_bezier_foot_curve(C.c_float(t), _p(_a(start)), _p(_a(fin)), C.c_double(pitch), _p(o)); return o


def skew(v):
    S = np.zeros(9); lib().ref_skew(_p(_a(v)), _p(S)); return S.reshape(3, 3).T.copy()


def quat_to_euler(w, x, y, z):
    e = np.zeros(3); lib().ref_quat_to_euler(C.c_double(w), C.c_double(x), C.c_double(y), C.c_double(z), _p(e)); return e


def pseudo_inverse(M):
    o = np.zeros(9); lib().ref_pseudo_inverse(_p(colmajor(M)), _p(o)); return o.reshape(3, 3).T.copy()


def dihedral_angle(a, b):
    return lib().ref_dihedral_angle(_p(_a(a)), _p(_a(b)))
