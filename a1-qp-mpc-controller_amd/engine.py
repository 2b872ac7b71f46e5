"""ctypes binding of liba1mpc.so -- the host-side mirror of the reference interface for the hot path.

The reference is C++ (ConvexMpc + A1RobotControl::compute_grf); the C++ adapter with the reference's own
class interface is include/a1mpc_convex_mpc.hpp.  This module is the Python plumbing the tests and
bench.py use: numpy arrays in the C-ABI layouts in, GRFs out.  There is no fallback: if the HIP library is
missing, or there is no GPU, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

NS, NU, NC = 13, 12, 20
STATUS_NAMES = {1: "solved", 2: "solved_inaccurate", -2: "max_iter_reached", -7: "non_cvx", -10: "unsolved"}
SUPPORTED_HORIZONS = (1, 4, 6, 8, 10, 12, 14, 16, 20)   # (1, 10, 16, 20: tuned; the others: the kernel families as they instantiate)


class Config(C.Structure):  # a1mpc_config, include/a1mpc.h
    _fields_ = [("horizon", C.c_int32), ("dt", C.c_double), ("mu", C.c_double), ("fz_min", C.c_double),
                ("fz_max", C.c_double), ("q", C.c_double * NS), ("r", C.c_double * NU), ("mass", C.c_double),
                ("inertia_body", C.c_double * 9), ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("adaptive_rho_tolerance", C.c_double),
                ("max_iter", C.c_int32), ("check_termination", C.c_int32), ("adaptive_rho", C.c_int32),
                ("adaptive_rho_interval", C.c_int32), ("scaling", C.c_int32), ("warm_start", C.c_int32)]


class BalanceConfig(C.Structure):  # a1mpc_balance_config
    _fields_ = [("Q", C.c_double * 6), ("R", C.c_double), ("mu", C.c_double), ("F_min", C.c_double), ("F_max", C.c_double)]


class GaitConfig(C.Structure):  # a1mpc_gait_config
    _fields_ = [("counter_per_gait", C.c_double), ("counter_per_swing", C.c_double), ("control_dt", C.c_double),
                ("foot_delta_x_limit", C.c_double), ("foot_delta_y_limit", C.c_double), ("default_foot_pos", C.c_double * 12),
                ("gait_counter_reset", C.c_double * 4)]


class ContactConfig(C.Structure):  # a1mpc_contact_config
    _fields_ = [("counter_per_swing", C.c_double), ("foot_force_low", C.c_double), ("use_terrain_adapt", C.c_int32)]


class TickParams(C.Structure):  # a1mpc_tick_params
    _fields_ = [("gait", GaitConfig), ("contact", ContactConfig), ("control_dt", C.c_double), ("assume_flat_ground", C.c_int32),
                ("kp_foot", C.c_double * 3), ("kd_foot", C.c_double * 3), ("km_foot", C.c_double * 3), ("rho_fix", C.c_double * 20), ("rho_opt", C.c_double * 12)]


TICK_BUFFER_FIELDS = ("joint_pos", "joint_vel", "R_world", "R_z", "root_euler", "root_ang_vel", "imu_acc", "imu_ang_vel", "foot_force", "movement_mode", "mpc_active",
                      "root_lin_vel_d", "root_ang_vel_d", "root_pos_d_z", "gait_counter_speed", "torques_gravity", "gait_counter", "foot_pos_start", "foot_pos_rel_last_time",
                      "foot_pos_target_last_time", "root_euler_d", "joint_torques", "root_pos", "root_lin_vel", "estimated_contacts", "plan_contacts", "contacts", "foot_pos_rel",
                      "j_foot_blocks", "foot_vel_rel", "foot_pos_abs", "foot_vel_abs", "foot_pos_world", "foot_vel_world", "foot_pos_target_rel", "foot_pos_target_abs",
                      "foot_pos_target_world", "foot_pos_cur", "foot_forces_kin", "foot_pos_recent_contact", "terrain_angle", "grf", "iters", "status")


class TickBuffers(C.Structure):  # a1mpc_tick_buffers: device pointers, in the header's order
    _fields_ = [(k, C.c_void_p) for k in TICK_BUFFER_FIELDS]


EXPORTS = ["a1mpc_set_timing", "a1mpc_default_tick_params", "a1mpc_control_tick_device", "a1mpc_last_control_tick_ms", "a1mpc_last_stage_ms", "a1mpc_sharded_create", "a1mpc_sharded_solve_batch", "a1mpc_sharded_solve_batch_ticks", "a1mpc_sharded_solve_batch_device", "a1mpc_sharded_solve_batch_ticks_device", "a1mpc_sharded_handle",
           "a1mpc_sharded_last_transfer", "a1mpc_sharded_info", "a1mpc_sharded_destroy", "a1mpc_terrain_batch", "a1mpc_form_qp_batch", "a1mpc_solve_batch_strided", "a1mpc_solve_batch_strided_device", "a1mpc_update_config", "a1mpc_warm_start", "a1mpc_get_warm_start", "a1mpc_get_workspace_z", "a1mpc_get_workspace_scaling", "a1mpc_last_warm_start_mode", "a1mpc_set_profiling", "a1mpc_last_stage_cycles", "a1mpc_last_tick_stage_cycles", "a1mpc_update_plan_batch_device", "a1mpc_swing_legs_batch_device", "a1mpc_contact_terrain_batch_device", "a1mpc_leg_state_batch_device",
           "a1mpc_ekf_update_batch_device", "a1mpc_joint_torques_batch_device", "a1mpc_ekf_update_batch", "a1mpc_reset_ekf_state", "a1mpc_leg_state_batch", "a1mpc_swing_legs_batch", "a1mpc_default_contact_config", "a1mpc_contact_terrain_batch", "a1mpc_reset_contact_state", "a1mpc_default_gait_config", "a1mpc_update_plan_batch", "a1mpc_joint_torques_batch", "a1mpc_set_schedule", "a1mpc_default_config", "a1mpc_default_balance_config", "a1mpc_create", "a1mpc_destroy", "a1mpc_solve_batch",
           "a1mpc_solve_batch_device", "a1mpc_solve_batch_ticks", "a1mpc_solve_batch_ticks_device", "a1mpc_balance_solve_batch", "a1mpc_reset_warm_start", "a1mpc_last_kernel_ms",
           "a1mpc_kernel_info", "a1mpc_last_nfact", "a1mpc_status_string", "a1mpc_last_error", "a1mpc_build_info", "a1mpc_pipeline_create", "a1mpc_pipeline_submit_device", "a1mpc_pipeline_submit",
           "a1mpc_pipeline_submit_strided_device", "a1mpc_pipeline_submit_strided", "a1mpc_pipeline_submit_ticks_device", "a1mpc_pipeline_wait", "a1mpc_pipeline_join", "a1mpc_pipeline_handle", "a1mpc_pipeline_depth", "a1mpc_pipeline_destroy"]

_lib = None


class A1MpcError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen liba1mpc.so (no compute).  Raises if the library has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _build.LIB_PATH
    if not os.path.exists(path):
        raise A1MpcError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback for the solver)")
    lib = C.CDLL(path)
    vp, i32, dp, u8p, i32p = C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    lib.a1mpc_default_config.argtypes = [C.POINTER(Config)]; lib.a1mpc_default_config.restype = None
    lib.a1mpc_default_balance_config.argtypes = [C.POINTER(BalanceConfig)]; lib.a1mpc_default_balance_config.restype = None
    lib.a1mpc_create.argtypes = [C.POINTER(Config), i32, i32, C.POINTER(vp)]; lib.a1mpc_create.restype = C.c_int
    lib.a1mpc_destroy.argtypes = [vp]; lib.a1mpc_destroy.restype = None
    lib.a1mpc_solve_batch.argtypes = [vp, i32, dp, dp, dp, dp, u8p, dp, dp, i32p, i32p]; lib.a1mpc_solve_batch.restype = C.c_int
    lib.a1mpc_solve_batch_device.argtypes = [vp, i32] + [vp] * 9 + [vp]; lib.a1mpc_solve_batch_device.restype = C.c_int
    lib.a1mpc_solve_batch_ticks.argtypes = [vp, i32, dp, dp, dp, u8p, dp, dp, i32p, i32p]; lib.a1mpc_solve_batch_ticks.restype = C.c_int
    lib.a1mpc_solve_batch_ticks_device.argtypes = [vp, i32] + [vp] * 8 + [vp]; lib.a1mpc_solve_batch_ticks_device.restype = C.c_int
    lib.a1mpc_balance_solve_batch.argtypes = [vp, C.POINTER(BalanceConfig), i32, dp, dp, dp, dp, u8p, dp, dp, i32p, i32p]
    lib.a1mpc_balance_solve_batch.restype = C.c_int
    lib.a1mpc_default_gait_config.argtypes = [C.POINTER(GaitConfig)]; lib.a1mpc_default_gait_config.restype = None
    lib.a1mpc_update_plan_batch.argtypes = [vp, C.POINTER(GaitConfig), i32, u8p, dp, dp, dp, dp, dp, dp, dp, u8p, dp, dp, dp]
    lib.a1mpc_update_plan_batch.restype = C.c_int
    lib.a1mpc_joint_torques_batch.argtypes = [vp, i32, u8p, u8p, dp, dp, dp, dp, dp, dp]; lib.a1mpc_joint_torques_batch.restype = C.c_int
    lib.a1mpc_default_contact_config.argtypes = [C.POINTER(ContactConfig)]; lib.a1mpc_default_contact_config.restype = None
    lib.a1mpc_contact_terrain_batch.argtypes = [vp, C.POINTER(ContactConfig), i32, dp, u8p, dp, dp, dp, dp, u8p, dp, dp]
    lib.a1mpc_contact_terrain_batch.restype = C.c_int
    lib.a1mpc_reset_contact_state.argtypes = [vp]; lib.a1mpc_reset_contact_state.restype = C.c_int
    lib.a1mpc_swing_legs_batch.argtypes = [vp, i32, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp]
    lib.a1mpc_swing_legs_batch.restype = C.c_int
    lib.a1mpc_leg_state_batch.argtypes = [vp, i32] + [dp] * 14; lib.a1mpc_leg_state_batch.restype = C.c_int
    lib.a1mpc_ekf_update_batch.argtypes = [vp, i32, C.c_double, i32, u8p, dp, dp, dp, dp, dp, dp, dp, dp, u8p]; lib.a1mpc_ekf_update_batch.restype = C.c_int
    lib.a1mpc_reset_ekf_state.argtypes = [vp]; lib.a1mpc_reset_ekf_state.restype = C.c_int
    vpp = C.c_void_p
    lib.a1mpc_update_plan_batch_device.argtypes = [vp, C.POINTER(GaitConfig), i32] + [vpp] * 12 + [vpp]; lib.a1mpc_update_plan_batch_device.restype = C.c_int
    lib.a1mpc_swing_legs_batch_device.argtypes = [vp, i32, C.c_double, C.c_double] + [vpp] * 4 + [dp, dp] + [vpp] * 5 + [vpp]; lib.a1mpc_swing_legs_batch_device.restype = C.c_int
    lib.a1mpc_contact_terrain_batch_device.argtypes = [vp, C.POINTER(ContactConfig), i32] + [vpp] * 9 + [vpp]; lib.a1mpc_contact_terrain_batch_device.restype = C.c_int
    lib.a1mpc_leg_state_batch_device.argtypes = [vp, i32] + [vpp] * 5 + [dp, dp] + [vpp] * 7 + [vpp]; lib.a1mpc_leg_state_batch_device.restype = C.c_int
    lib.a1mpc_ekf_update_batch_device.argtypes = [vp, i32, C.c_double, i32] + [vpp] * 10 + [vpp]; lib.a1mpc_ekf_update_batch_device.restype = C.c_int
    lib.a1mpc_joint_torques_batch_device.argtypes = [vp, i32] + [vpp] * 5 + [dp] + [vpp] * 2 + [vpp]; lib.a1mpc_joint_torques_batch_device.restype = C.c_int
    lib.a1mpc_solve_batch_strided.argtypes = [vp, i32, dp, dp, dp, dp, i32, u8p, i32, dp, dp, dp, i32p, i32p]; lib.a1mpc_solve_batch_strided.restype = C.c_int
    lib.a1mpc_solve_batch_strided_device.argtypes = [vp, i32] + [vp] * 4 + [i32, vp, i32] + [vp] * 5 + [vp]; lib.a1mpc_solve_batch_strided_device.restype = C.c_int
    lib.a1mpc_sharded_create.argtypes = [C.POINTER(Config), i32, i32p, i32, i32, C.POINTER(vp)]; lib.a1mpc_sharded_create.restype = C.c_int
    lib.a1mpc_sharded_solve_batch.argtypes = [vp, i32, dp, dp, dp, dp, u8p, dp, i32p, i32p]; lib.a1mpc_sharded_solve_batch.restype = C.c_int
    lib.a1mpc_sharded_info.argtypes = [vp, i32p, i32p, i32p]; lib.a1mpc_sharded_info.restype = C.c_int
    if path == _build.LIB_PATH or hasattr(lib, "a1mpc_sharded_solve_batch_ticks"):   # (round 6)
        lib.a1mpc_sharded_solve_batch_ticks.argtypes = [vp, i32, dp, dp, dp, u8p, dp, i32p, i32p]; lib.a1mpc_sharded_solve_batch_ticks.restype = C.c_int
        lib.a1mpc_sharded_solve_batch_device.argtypes = [vp, i32] + [vp] * 9; lib.a1mpc_sharded_solve_batch_device.restype = C.c_int
        lib.a1mpc_sharded_solve_batch_ticks_device.argtypes = [vp, i32] + [vp] * 8; lib.a1mpc_sharded_solve_batch_ticks_device.restype = C.c_int
        lib.a1mpc_sharded_handle.argtypes = [vp, i32, C.POINTER(vp)]; lib.a1mpc_sharded_handle.restype = C.c_int
        lib.a1mpc_sharded_last_transfer.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]; lib.a1mpc_sharded_last_transfer.restype = C.c_int
    lib.a1mpc_sharded_destroy.argtypes = [vp]; lib.a1mpc_sharded_destroy.restype = None
    lib.a1mpc_pipeline_create.argtypes = [C.POINTER(Config), i32, i32, i32, C.POINTER(vp)]; lib.a1mpc_pipeline_create.restype = C.c_int
    lib.a1mpc_pipeline_submit_device.argtypes = [vp, i32, i32, i32] + [vp] * 9 + [vp, i32p]; lib.a1mpc_pipeline_submit_device.restype = C.c_int
    if path == _build.LIB_PATH or hasattr(lib, "a1mpc_pipeline_submit"):  # (an older build bound by hand for an A/B, tools/ab_probe.py, may lack the round-3 entries)
        lib.a1mpc_pipeline_submit.argtypes = [vp, i32, i32, i32, dp, dp, dp, dp, u8p, dp, dp, i32p, i32p, i32p]; lib.a1mpc_pipeline_submit.restype = C.c_int
    if path == _build.LIB_PATH or hasattr(lib, "a1mpc_pipeline_submit_strided_device"):   # (round 6)
        lib.a1mpc_pipeline_submit_strided_device.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32p]; lib.a1mpc_pipeline_submit_strided_device.restype = C.c_int
        lib.a1mpc_pipeline_submit_strided.argtypes = [vp, i32, i32, i32, dp, dp, dp, dp, i32, u8p, i32, dp, dp, dp, i32p, i32p, i32p]; lib.a1mpc_pipeline_submit_strided.restype = C.c_int
        lib.a1mpc_pipeline_submit_ticks_device.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32p]; lib.a1mpc_pipeline_submit_ticks_device.restype = C.c_int
    lib.a1mpc_pipeline_wait.argtypes = [vp, i32]; lib.a1mpc_pipeline_wait.restype = C.c_int
    lib.a1mpc_pipeline_join.argtypes = [vp, i32, vp]; lib.a1mpc_pipeline_join.restype = C.c_int
    lib.a1mpc_pipeline_handle.argtypes = [vp, i32, C.POINTER(vp)]; lib.a1mpc_pipeline_handle.restype = C.c_int
    lib.a1mpc_pipeline_depth.argtypes = [vp, i32p]; lib.a1mpc_pipeline_depth.restype = C.c_int
    lib.a1mpc_pipeline_destroy.argtypes = [vp]; lib.a1mpc_pipeline_destroy.restype = None
    lib.a1mpc_terrain_batch.argtypes = [vp, i32, i32, dp, dp, dp, dp]; lib.a1mpc_terrain_batch.restype = C.c_int
    lib.a1mpc_form_qp_batch.argtypes = [vp, i32, dp, dp, dp, dp, i32, u8p, i32, dp, dp, dp, dp, dp]; lib.a1mpc_form_qp_batch.restype = C.c_int
    lib.a1mpc_update_config.argtypes = [vp, C.POINTER(Config)]; lib.a1mpc_update_config.restype = C.c_int
    lib.a1mpc_warm_start.argtypes = [vp, i32, dp, dp, dp]; lib.a1mpc_warm_start.restype = C.c_int
    lib.a1mpc_get_warm_start.argtypes = [vp, i32, dp, dp, dp]; lib.a1mpc_get_warm_start.restype = C.c_int
    lib.a1mpc_get_workspace_z.argtypes = [vp, i32, dp]; lib.a1mpc_get_workspace_z.restype = C.c_int
    lib.a1mpc_get_workspace_scaling.argtypes = [vp, i32, dp, dp, dp]; lib.a1mpc_get_workspace_scaling.restype = C.c_int
    lib.a1mpc_set_profiling.argtypes = [vp, i32]; lib.a1mpc_set_profiling.restype = C.c_int
    lib.a1mpc_last_stage_cycles.argtypes = [vp, dp, C.POINTER(C.c_int32)]; lib.a1mpc_last_stage_cycles.restype = C.c_int
    lib.a1mpc_last_warm_start_mode.argtypes = [vp, C.POINTER(C.c_int32)]; lib.a1mpc_last_warm_start_mode.restype = C.c_int
    if path == _build.LIB_PATH or hasattr(lib, "a1mpc_set_timing"):   # (round 5)
        lib.a1mpc_set_timing.argtypes = [vp, i32]; lib.a1mpc_set_timing.restype = C.c_int
    if path == _build.LIB_PATH or hasattr(lib, "a1mpc_control_tick_device"):   # (round 5)
        lib.a1mpc_default_tick_params.argtypes = [C.POINTER(TickParams)]; lib.a1mpc_default_tick_params.restype = None
        lib.a1mpc_control_tick_device.argtypes = [vp, C.POINTER(TickParams), C.POINTER(TickBuffers), i32, vp]; lib.a1mpc_control_tick_device.restype = C.c_int
        lib.a1mpc_last_control_tick_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]; lib.a1mpc_last_control_tick_ms.restype = C.c_int
    if path == _build.LIB_PATH or hasattr(lib, "a1mpc_last_tick_stage_cycles"):  # (round 5; an older build bound by hand for an A/B may lack it)
        lib.a1mpc_last_tick_stage_cycles.argtypes = [vp, dp, C.POINTER(C.c_int32)]; lib.a1mpc_last_tick_stage_cycles.restype = C.c_int
    lib.a1mpc_set_schedule.argtypes = [vp, i32]; lib.a1mpc_set_schedule.restype = C.c_int
    lib.a1mpc_reset_warm_start.argtypes = [vp]; lib.a1mpc_reset_warm_start.restype = C.c_int
    lib.a1mpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]; lib.a1mpc_last_kernel_ms.restype = C.c_int
    lib.a1mpc_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]; lib.a1mpc_last_stage_ms.restype = C.c_int
    lib.a1mpc_last_nfact.argtypes = [vp, i32, i32p]; lib.a1mpc_last_nfact.restype = C.c_int
    lib.a1mpc_kernel_info.argtypes = [vp, i32p, i32p, i32p]; lib.a1mpc_kernel_info.restype = C.c_int
    lib.a1mpc_status_string.argtypes = [C.c_int]; lib.a1mpc_status_string.restype = C.c_char_p
    lib.a1mpc_last_error.argtypes = []; lib.a1mpc_last_error.restype = C.c_char_p
    if path == _build.LIB_PATH:
        _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        raise A1MpcError(f"{what}: {lib.a1mpc_status_string(rc).decode()} ({lib.a1mpc_last_error().decode()})")


def make_config(params, horizon, **osqp):
    """params: dict with dt, mu, fz_min, fz_max, q(13), r(12), mass, inertia(9) -- e.g. scenarios.*()['params'].
    osqp: overrides of the OSQP settings fields (rho, eps_abs, eps_rel, max_iter, warm_start, ...)."""
    lib = load_library()
    cfg = Config()
    lib.a1mpc_default_config(C.byref(cfg))
    cfg.horizon = int(horizon)
    cfg.dt, cfg.mu, cfg.fz_min, cfg.fz_max, cfg.mass = params["dt"], params["mu"], params["fz_min"], params["fz_max"], params["mass"]
    cfg.q[:] = [float(v) for v in params["q"]]
    cfg.r[:] = [float(v) for v in params["r"]]
    cfg.inertia_body[:] = [float(v) for v in np.asarray(params["inertia"], dtype=float).reshape(9)]
    for k, v in osqp.items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    return cfg


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape)


class Engine:
    """One a1mpc handle: a fixed (robot constants, horizon, OSQP settings) configuration on one GPU."""

    def __init__(self, cfg, max_batch, device=0):
        self.lib = load_library()
        self.cfg = cfg
        self.horizon = int(cfg.horizon)
        self.max_batch = int(max_batch)
        self.device = int(device)
        self._h = C.c_void_p()
        _check(self.lib, self.lib.a1mpc_create(C.byref(cfg), self.max_batch, self.device, C.byref(self._h)), "a1mpc_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.a1mpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- host arrays (the drop-in path: S/A1RobotControl.cpp:446-562 for n ticks) ----
    def solve(self, x0, xref, R, foot, contact, want_u=False):
        h = self.horizon
        x0 = _f64(x0, (-1, NS)); n = x0.shape[0]
        xref = _f64(xref, (n, NS * h)); R = _f64(R, (n, 9)); foot = _f64(foot, (n, 12))
        contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(n, 4)
        grf = np.zeros((n, 12)); u = np.zeros((n, NU * h)) if want_u else None
        iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
        rc = self.lib.a1mpc_solve_batch(self._h, n, _dp(x0), _dp(xref), _dp(R), _dp(foot),
                                        contact.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(grf), _dp(u), _ip(iters), _ip(status))
        _check(self.lib, rc, "a1mpc_solve_batch")
        return dict(grf=grf, u=u, iters=iters, status=status)

    # ---- the general case of the ConvexMpc interface: per-step feet (B_mat_d_list) and / or a per-step contact schedule ----
    def solve_strided(self, x0, xref, R, foot, foot_stride, contact, contact_stride, want_u=False, yaw_A=None):
        h = self.horizon
        x0 = _f64(x0, (-1, NS)); n = x0.shape[0]
        xref = _f64(xref, (n, NS * h)); R = _f64(R, (n, 9)); foot = _f64(foot, (n, 12 * h if foot_stride else 12))
        contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(n, 4 * h if contact_stride else 4)
        grf = np.zeros((n, 12)); u = np.zeros((n, NU * h)) if want_u else None
        iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
        rc = self.lib.a1mpc_solve_batch_strided(self._h, n, _dp(x0), _dp(xref), _dp(R), _dp(foot), int(foot_stride),
                                                contact.ctypes.data_as(C.POINTER(C.c_uint8)), int(contact_stride),
                                                None if yaw_A is None else _dp(np.ascontiguousarray(yaw_A, dtype=np.float64)), _dp(grf), _dp(u), _ip(iters), _ip(status))
        _check(self.lib, rc, "a1mpc_solve_batch_strided")
        return dict(grf=grf, u=u, iters=iters, status=status)

    def form_qp(self, x0, xref, R, foot, contact, foot_stride=0, contact_stride=0, yaw_A=None):
        """the dense (P, g, l, u) the reference's ConvexMpc members hold, formed on the GPU (debug / verification)"""
        h = self.horizon
        x0 = _f64(x0, (-1, NS)); n = x0.shape[0]
        xref = _f64(xref, (n, NS * h)); R = _f64(R, (n, 9)); foot = _f64(foot, (n, 12 * h if foot_stride else 12))
        contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(n, 4 * h if contact_stride else 4)
        P = np.zeros((n, 12 * h, 12 * h)); g = np.zeros((n, 12 * h)); l = np.zeros((n, 20 * h)); u = np.zeros((n, 20 * h))
        rc = self.lib.a1mpc_form_qp_batch(self._h, n, _dp(x0), _dp(xref), _dp(R), _dp(foot), int(foot_stride), contact.ctypes.data_as(C.POINTER(C.c_uint8)),
                                          int(contact_stride), None if yaw_A is None else _dp(np.ascontiguousarray(yaw_A, dtype=np.float64)), _dp(P), _dp(g), _dp(l), _dp(u))
        _check(self.lib, rc, "a1mpc_form_qp_batch")
        return dict(P=P, g=g, l=l, u=u)

    def terrain(self, foot_pos_recent_contact, root_pos_z, root_euler_d_pitch, use_terrain_adapt=1):
        rec = _f64(foot_pos_recent_contact, (-1, 12)); n = rec.shape[0]
        z = _f64(root_pos_z, (n,)); pd = _f64(root_euler_d_pitch, (n,)).copy(); ta = np.zeros(n)
        _check(self.lib, self.lib.a1mpc_terrain_batch(self._h, int(use_terrain_adapt), n, _dp(rec), _dp(z), _dp(pd), _dp(ta)), "a1mpc_terrain_batch")
        return pd, ta

    def update_config(self, cfg):
        """replace everything but the horizon (dt, weights, mass / inertia, limits, OSQP settings); the warm start is kept"""
        _check(self.lib, self.lib.a1mpc_update_config(self._h, C.byref(cfg)), "a1mpc_update_config")
        self.cfg = cfg

    def set_warm_start(self, x=None, y=None, rho=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, rho)]
        n = next(a for a in arrs if a is not None).shape[0]
        _check(self.lib, self.lib.a1mpc_warm_start(self._h, n, _dp(arrs[0]), _dp(arrs[1]), _dp(arrs[2])), "a1mpc_warm_start")

    def get_warm_start(self, n):
        h = self.horizon
        x = np.zeros((n, NU * h)); y = np.zeros((n, 20 * h)); rho = np.zeros(n)
        _check(self.lib, self.lib.a1mpc_get_warm_start(self._h, int(n), _dp(x), _dp(y), _dp(rho)), "a1mpc_get_warm_start")
        return x, y, rho

    def set_timing(self, on=True):
        """a1mpc_set_timing: the handle's HIP timing events on / off (off: last_kernel_ms & co. are unavailable, every tick is a few event records lighter)"""
        _check(self.lib, self.lib.a1mpc_set_timing(self._h, 1 if on else 0), "a1mpc_set_timing")

    def set_profiling(self, on=True):
        _check(self.lib, self.lib.a1mpc_set_profiling(self._h, 1 if on else 0), "a1mpc_set_profiling")

    def last_stage_cycles(self):
        """dict(factor, iterate, check: shader-clock cycles summed over the QPs of the last profiled solve; qps) -- a1mpc_last_stage_cycles"""
        c = np.zeros(3); q = C.c_int32(0)
        _check(self.lib, self.lib.a1mpc_last_stage_cycles(self._h, _dp(c), C.byref(q)), "a1mpc_last_stage_cycles")
        return dict(factor=float(c[0]), iterate=float(c[1]), check=float(c[2]), qps=int(q.value))

    def control_tick_device(self, params, buffers, n, stream=None):
        """a1mpc_control_tick_device: one whole control tick of n robots (device pointers in a TickBuffers), asynchronous on `stream`"""
        _check(self.lib, self.lib.a1mpc_control_tick_device(self._h, C.byref(params), C.byref(buffers), int(n), C.c_void_p(stream) if stream else None), "a1mpc_control_tick_device")

    def last_control_tick_ms(self):
        ms = C.c_float(0); fused = C.c_int32(0)
        _check(self.lib, self.lib.a1mpc_last_control_tick_ms(self._h, C.byref(ms), C.byref(fused)), "a1mpc_last_control_tick_ms")
        return float(ms.value), bool(fused.value)

    TICK_STAGES = ("formation", "ruiz", "handoff", "factor", "iterate", "check", "outputs", "total")

    def last_tick_stage_cycles(self):
        """dict(stage -> shader-clock cycles summed over the QPs of the last profiled fused / latency tick; qps) -- a1mpc_last_tick_stage_cycles"""
        c = np.zeros(8); q = C.c_int32(0)
        _check(self.lib, self.lib.a1mpc_last_tick_stage_cycles(self._h, _dp(c), C.byref(q)), "a1mpc_last_tick_stage_cycles")
        return dict(zip(self.TICK_STAGES, map(float, c)), qps=int(q.value))

    def last_warm_start_mode(self):
        m = C.c_int32(-9)
        _check(self.lib, self.lib.a1mpc_last_warm_start_mode(self._h, C.byref(m)), "a1mpc_last_warm_start_mode")
        return int(m.value)

    def get_workspace_scaling(self, n):
        """warm_start = 2: (D (n, 12 h), E (n, 20 h), c (n)) of the last update-path tick (a1mpc_get_workspace_scaling)"""
        h = self.horizon
        D = np.zeros((n, NU * h)); E = np.zeros((n, 20 * h)); c = np.zeros(n)
        _check(self.lib, self.lib.a1mpc_get_workspace_scaling(self._h, int(n), _dp(D), _dp(E), _dp(c)), "a1mpc_get_workspace_scaling")
        return D, E, c

    def get_workspace_z(self, n):
        """warm_start = 2: the unscaled z the update path keeps beside (x, y, rho), (n, 20 h) in the reference's row order (a1mpc_get_workspace_z)"""
        z = np.zeros((n, 20 * self.horizon))
        _check(self.lib, self.lib.a1mpc_get_workspace_z(self._h, int(n), _dp(z)), "a1mpc_get_workspace_z")
        return z

    # ---- N1: compact tick records (x0 / x_ref built on the device, S/A1RobotControl.cpp:452-488) ----
    def solve_ticks(self, tick, R, foot, contact, want_u=False):
        h = self.horizon
        tick = _f64(tick, (-1, 22)); n = tick.shape[0]
        R = _f64(R, (n, 9)); foot = _f64(foot, (n, 12))
        contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(n, 4)
        grf = np.zeros((n, 12)); u = np.zeros((n, NU * h)) if want_u else None
        iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
        rc = self.lib.a1mpc_solve_batch_ticks(self._h, n, _dp(tick), _dp(R), _dp(foot), contact.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(grf),
                                              _dp(u), _ip(iters), _ip(status))
        _check(self.lib, rc, "a1mpc_solve_batch_ticks")
        return dict(grf=grf, u=u, iters=iters, status=status)

    # ---- device pointers (torch tensors already resident in HBM), asynchronous on `stream` ----
    def solve_device(self, n, d_x0, d_xref, d_R, d_foot, d_contact, d_grf, d_u=None, d_iters=None, d_status=None, stream=None):
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        rc = self.lib.a1mpc_solve_batch_device(self._h, int(n), ptr(d_x0), ptr(d_xref), ptr(d_R), ptr(d_foot), ptr(d_contact),
                                               ptr(d_grf), ptr(d_u), ptr(d_iters), ptr(d_status),
                                               C.c_void_p(int(stream)) if stream else None)
        _check(self.lib, rc, "a1mpc_solve_batch_device")

    def balance_solve(self, root_acc, R, Rz, foot, contact, qp=None):
        if qp is None:
            qp = BalanceConfig(); self.lib.a1mpc_default_balance_config(C.byref(qp))
        root_acc = _f64(root_acc, (-1, 6)); n = root_acc.shape[0]
        R = _f64(R, (n, 9)); Rz = _f64(Rz, (n, 9)); foot = _f64(foot, (n, 12))
        contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(n, 4)
        grf = np.zeros((n, 12)); f = np.zeros((n, 12)); iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
        rc = self.lib.a1mpc_balance_solve_batch(self._h, C.byref(qp), n, _dp(root_acc), _dp(R), _dp(Rz), _dp(foot),
                                                contact.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(grf), _dp(f), _ip(iters), _ip(status))
        _check(self.lib, rc, "a1mpc_balance_solve_batch")
        return dict(grf=grf, f_world=f, iters=iters, status=status)

    # ---- N2a: gait plan + Raibert foothold (S/A1RobotControl.cpp:148-202) ----
    def update_plan(self, movement_mode, gait_counter, gait_counter_speed, root_lin_vel, Rz, R, root_pos, root_lin_vel_d, gait=None):
        if gait is None:
            gait = GaitConfig(); self.lib.a1mpc_default_gait_config(C.byref(gait))
        gc = np.array(gait_counter, dtype=np.float64).reshape(-1, 4); n = gc.shape[0]
        mm = np.ascontiguousarray(movement_mode, dtype=np.uint8).reshape(n)
        spd = _f64(gait_counter_speed, (n, 4)); v = _f64(root_lin_vel, (n, 3)); Rz = _f64(Rz, (n, 9)); R = _f64(R, (n, 9))
        pos = _f64(root_pos, (n, 3)); vd = _f64(root_lin_vel_d, (n, 3))
        pc = np.zeros((n, 4), np.uint8); rel = np.zeros((n, 12)); ab = np.zeros((n, 12)); wo = np.zeros((n, 12))
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        rc = self.lib.a1mpc_update_plan_batch(self._h, C.byref(gait), n, u8(mm), _dp(gc), _dp(spd), _dp(v), _dp(Rz), _dp(R), _dp(pos), _dp(vd), u8(pc),
                                              _dp(rel), _dp(ab), _dp(wo))
        _check(self.lib, rc, "a1mpc_update_plan_batch")
        return dict(gait_counter=gc, plan_contacts=pc, foot_pos_target_rel=rel, foot_pos_target_abs=ab, foot_pos_target_world=wo)

    # ---- N3: GRF -> joint torques (S/A1RobotControl.cpp:289-319) ----
    def joint_torques(self, active, contacts, j_foot_blocks, grf, f_kin, km_foot, torques_gravity, joint_torques_prev):
        tau = np.array(joint_torques_prev, dtype=np.float64).reshape(-1, 12); n = tau.shape[0]
        act = np.ascontiguousarray(active, dtype=np.uint8).reshape(n); c = np.ascontiguousarray(contacts, dtype=np.uint8).reshape(n, 4)
        Jb = _f64(j_foot_blocks, (n, 36)); g = _f64(grf, (n, 12)); fk = _f64(f_kin, (n, 12)); km = _f64(km_foot, (3,)); tg = _f64(torques_gravity, (n, 12))
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        rc = self.lib.a1mpc_joint_torques_batch(self._h, n, u8(act), u8(c), _dp(Jb), _dp(g), _dp(fk), _dp(km), _dp(tg), _dp(tau))
        _check(self.lib, rc, "a1mpc_joint_torques_batch")
        return tau

    # ---- N2b: contacts, recent-contact filters, terrain pitch (S/A1RobotControl.cpp:256-282, 566-582, 335-376); state lives on the device ----
    def contact_terrain(self, gait_counter, plan_contacts, foot_force, foot_pos_abs, root_pos_z, root_euler_d_pitch, cfg=None):
        if cfg is None:
            cfg = ContactConfig(); self.lib.a1mpc_default_contact_config(C.byref(cfg))
        gc = np.ascontiguousarray(gait_counter, dtype=np.float64).reshape(-1, 4); n = gc.shape[0]; pc = np.ascontiguousarray(plan_contacts, dtype=np.uint8).reshape(n, 4)
        ff = _f64(foot_force, (n, 4)); fp = _f64(foot_pos_abs, (n, 12)); z = _f64(root_pos_z, (n,))
        pd = np.array(root_euler_d_pitch, dtype=np.float64).reshape(n)
        ct = np.zeros((n, 4), np.uint8); rec = np.zeros((n, 12)); ta = np.zeros(n)
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        rc = self.lib.a1mpc_contact_terrain_batch(self._h, C.byref(cfg), n, _dp(gc), u8(pc), _dp(ff), _dp(fp), _dp(z), _dp(pd), u8(ct), _dp(rec), _dp(ta))
        _check(self.lib, rc, "a1mpc_contact_terrain_batch")
        return dict(contacts=ct, foot_pos_recent_contact=rec, terrain_angle=ta, root_euler_d_pitch=pd)

    # ---- N4a: swing-leg targets + foot PD force (S/A1RobotControl.cpp:204-254); the three state arrays are updated in place ----
    def swing_legs(self, Rz, foot_pos_abs, gait_counter, foot_pos_target_rel, foot_pos_start, rel_last, target_last, kp=(300.0, 400.0, 400.0),
                   kd=(8.0, 8.0, 8.0), counter_per_swing=120.0, dt=0.0025):
        n = foot_pos_start.shape[0]
        for st in (foot_pos_start, rel_last, target_last):
            assert st.dtype == np.float64 and st.flags["C_CONTIGUOUS"] and st.shape == (n, 12)
        Rz = _f64(Rz, (n, 9)); fa = _f64(foot_pos_abs, (n, 12)); gc = _f64(gait_counter, (n, 4)); tr = _f64(foot_pos_target_rel, (n, 12))
        kp = _f64(kp, (3,)); kd = _f64(kd, (3,)); cur = np.zeros((n, 12)); kin = np.zeros((n, 12))
        rc = self.lib.a1mpc_swing_legs_batch(self._h, n, counter_per_swing, dt, _dp(Rz), _dp(fa), _dp(gc), _dp(tr), _dp(kp), _dp(kd), _dp(foot_pos_start),
                                             _dp(rel_last), _dp(target_last), _dp(cur), _dp(kin))
        _check(self.lib, rc, "a1mpc_swing_legs_batch")
        return cur, kin

    # ---- N4b: leg kinematics (S/GazeboA1ROS.cpp:264-279) ----
    A1_RHO_FIX = np.array([[0.1805, 0.047, 0.0838, 0.21, 0.21], [0.1805, -0.047, -0.0838, 0.21, 0.21], [-0.1805, 0.047, 0.0838, 0.21, 0.21],
                           [-0.1805, -0.047, -0.0838, 0.21, 0.21]])

    def leg_state(self, joint_pos, joint_vel, R, root_pos, root_lin_vel, rho_fix=None, rho_opt=None):
        q = _f64(joint_pos, (-1, 12)); n = q.shape[0]
        qd = _f64(joint_vel, (n, 12)); R = _f64(R, (n, 9)); pos = _f64(root_pos, (n, 3)); vel = _f64(root_lin_vel, (n, 3))
        fix = _f64(self.A1_RHO_FIX if rho_fix is None else rho_fix, (4, 5)); opt = _f64(np.zeros((4, 3)) if rho_opt is None else rho_opt, (4, 3))
        names = ("foot_pos_rel", "Jb", "foot_vel_rel", "foot_pos_abs", "foot_vel_abs", "foot_pos_world", "foot_vel_world")
        out = {k: np.zeros((n, 36 if k == "Jb" else 12)) for k in names}
        rc = self.lib.a1mpc_leg_state_batch(self._h, n, _dp(q), _dp(qd), _dp(R), _dp(pos), _dp(vel), _dp(fix), _dp(opt), *[_dp(out[k]) for k in names])
        _check(self.lib, rc, "a1mpc_leg_state_batch")
        return out

    # ---- N4c: A1BasicEKF (S/A1BasicEKF.cpp); filter state per robot on the device, first call = init_state ----
    def ekf_update(self, dt, movement_mode, foot_force, R, imu_acc, imu_ang_vel, foot_pos_rel, foot_vel_rel, assume_flat_ground=1):
        ff = _f64(foot_force, (-1, 4)); n = ff.shape[0]
        mm = np.ascontiguousarray(movement_mode, dtype=np.uint8).reshape(n)
        R = _f64(R, (n, 9)); acc = _f64(imu_acc, (n, 3)); w = _f64(imu_ang_vel, (n, 3)); fk = _f64(foot_pos_rel, (n, 12)); fv = _f64(foot_vel_rel, (n, 12))
        pos = np.zeros((n, 3)); vel = np.zeros((n, 3)); ec = np.zeros((n, 4), np.uint8)
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        rc = self.lib.a1mpc_ekf_update_batch(self._h, n, float(dt), int(assume_flat_ground), u8(mm), _dp(ff), _dp(R), _dp(acc), _dp(w), _dp(fk), _dp(fv),
                                             _dp(pos), _dp(vel), u8(ec))
        _check(self.lib, rc, "a1mpc_ekf_update_batch")
        return pos, vel, ec

    def reset_ekf_state(self):
        _check(self.lib, self.lib.a1mpc_reset_ekf_state(self._h), "a1mpc_reset_ekf_state")

    def reset_contact_state(self):
        _check(self.lib, self.lib.a1mpc_reset_contact_state(self._h), "a1mpc_reset_contact_state")

    def set_schedule(self, history=True):
        """queue order of batches beyond the resident set: previous solve's longest-first (default) or index order"""
        _check(self.lib, self.lib.a1mpc_set_schedule(self._h, 1 if history else 0), "a1mpc_set_schedule")

    def reset_warm_start(self):
        _check(self.lib, self.lib.a1mpc_reset_warm_start(self._h), "a1mpc_reset_warm_start")

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(self.lib, self.lib.a1mpc_last_kernel_ms(self._h, C.byref(ms)), "a1mpc_last_kernel_ms")
        return float(ms.value)

    def last_stage_ms(self):
        a = C.c_float(); b = C.c_float()
        _check(self.lib, self.lib.a1mpc_last_stage_ms(self._h, C.byref(a), C.byref(b)), "a1mpc_last_stage_ms")
        return float(a.value), float(b.value)

    def last_nfact(self, n):
        out = np.zeros(int(n), np.int32)
        _check(self.lib, self.lib.a1mpc_last_nfact(self._h, int(n), _ip(out)), "a1mpc_last_nfact")
        return out

    def kernel_info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(self.lib, self.lib.a1mpc_kernel_info(self._h, C.byref(a), C.byref(b), C.byref(c)), "a1mpc_kernel_info")
        return dict(lds_bytes_per_workgroup=a.value, qps_per_workgroup=b.value, threads_per_workgroup=c.value)


# ---- work model used for roofline.achieved (DESIGN.md "Measurement") --------------
class Pipeline:
    """a1mpc_pipeline: `depth` engine handles on `depth` HIP streams of one GPU, batches submitted round-robin so that the next batch's
    set-up kernel and persistent rows fill the tail of the one before (include/a1mpc.h).  Device pointers (torch tensors) or host arrays."""

    def __init__(self, cfg, max_batch, device=0, depth=2):
        self.lib = load_library()
        self.cfg = cfg; self.horizon = int(cfg.horizon); self.max_batch = int(max_batch); self.device = int(device)
        self._p = C.c_void_p()
        _check(self.lib, self.lib.a1mpc_pipeline_create(C.byref(cfg), self.max_batch, self.device, int(depth), C.byref(self._p)), "a1mpc_pipeline_create")
        d = C.c_int32(0)
        _check(self.lib, self.lib.a1mpc_pipeline_depth(self._p, C.byref(d)), "a1mpc_pipeline_depth")
        self.depth = int(d.value)

    def submit_device(self, n, d_x0, d_xref, d_R, d_foot, d_contact, d_grf, d_u=None, d_iters=None, d_status=None, slot=-1, fresh=True, after_stream=None):
        """returns the slot the batch went to; outputs are valid after wait(slot) / join(slot, stream)"""
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        k = C.c_int32(-1)
        rc = self.lib.a1mpc_pipeline_submit_device(self._p, int(slot), 1 if fresh else 0, int(n), ptr(d_x0), ptr(d_xref), ptr(d_R), ptr(d_foot), ptr(d_contact),
                                                   ptr(d_grf), ptr(d_u), ptr(d_iters), ptr(d_status), C.c_void_p(int(after_stream)) if after_stream else None, C.byref(k))
        _check(self.lib, rc, "a1mpc_pipeline_submit_device")
        return int(k.value)

    def submit(self, x0, xref, R, foot, contact, out, slot=-1, fresh=True):
        """host arrays in (a1mpc_solve_batch's layouts; snapshotted before the call returns), host arrays out: `out` = dict(grf=, [u=], [iters=], [status=]) of
        preallocated C-contiguous numpy arrays that wait(slot) -- or the next submit to the slot -- fills.  Returns the slot."""
        n = int(x0.shape[0])
        x0 = np.ascontiguousarray(x0, np.float64); xref = np.ascontiguousarray(xref, np.float64); R = np.ascontiguousarray(R, np.float64)
        foot = np.ascontiguousarray(foot, np.float64); contact = np.ascontiguousarray(contact, np.uint8)
        # the C side memcpys n*12 (grf) / n*12*H (u) doubles and n int32 (iters, status) into these arrays LATER (wait / the next submit): a wrong dtype or a short
        # array would be a deferred heap overflow, so dtype and total size are checked here, before the call
        need = {"grf": (np.float64, n * NU), "u": (np.float64, n * NU * self.horizon), "iters": (np.int32, n), "status": (np.int32, n)}
        for k_, a_ in out.items():
            if a_ is None:
                continue
            if k_ not in need:
                raise ValueError(f"unknown output array {k_!r}")
            dt_, size_ = need[k_]
            if not (isinstance(a_, np.ndarray) and a_.flags["C_CONTIGUOUS"] and a_.dtype == dt_ and a_.size >= size_):
                raise ValueError(f"output array {k_!r} must be a C-contiguous numpy array of {np.dtype(dt_).name} with at least {size_} elements")
        if out.get("grf") is None:
            raise ValueError("output array 'grf' is required")
        k = C.c_int32(-1)
        rc = self.lib.a1mpc_pipeline_submit(self._p, int(slot), 1 if fresh else 0, n, _dp(x0), _dp(xref), _dp(R), _dp(foot), _u8p(contact), _dp(out["grf"]),
                                            _dp(out["u"]) if out.get("u") is not None else None, _ip(out["iters"]) if out.get("iters") is not None else None,
                                            _ip(out["status"]) if out.get("status") is not None else None, C.byref(k))
        _check(self.lib, rc, "a1mpc_pipeline_submit")
        self._keep = getattr(self, "_keep", {}); self._keep[int(k.value)] = out   # the output arrays must outlive the slot's batch
        return int(k.value)

    def submit_strided_device(self, n, d_x0, d_xref, d_R, d_foot, foot_stride, d_contact, contact_stride, d_grf, d_u=None, d_iters=None, d_status=None, d_yaw_A=None,
                              slot=-1, fresh=True, after_stream=None):
        """a1mpc_pipeline_submit_strided_device: per-step feet / contact schedules / an A_c yaw of its own (the general path) with batches in flight together"""
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        k = C.c_int32(-1)
        rc = self.lib.a1mpc_pipeline_submit_strided_device(self._p, int(slot), 1 if fresh else 0, int(n), ptr(d_x0), ptr(d_xref), ptr(d_R), ptr(d_foot), int(foot_stride),
                                                           ptr(d_contact), int(contact_stride), ptr(d_yaw_A), ptr(d_grf), ptr(d_u), ptr(d_iters), ptr(d_status),
                                                           C.c_void_p(int(after_stream)) if after_stream else None, C.byref(k))
        _check(self.lib, rc, "a1mpc_pipeline_submit_strided_device")
        return int(k.value)

    def submit_ticks_device(self, n, d_tick, d_R, d_foot, d_contact, d_grf, d_u=None, d_iters=None, d_status=None, slot=-1, fresh=True, after_stream=None):
        """a1mpc_pipeline_submit_ticks_device: the compact 22-number tick records (x0 / x_ref are built on the device) with batches in flight together"""
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        k = C.c_int32(-1)
        rc = self.lib.a1mpc_pipeline_submit_ticks_device(self._p, int(slot), 1 if fresh else 0, int(n), ptr(d_tick), ptr(d_R), ptr(d_foot), ptr(d_contact), ptr(d_grf), ptr(d_u),
                                                         ptr(d_iters), ptr(d_status), C.c_void_p(int(after_stream)) if after_stream else None, C.byref(k))
        _check(self.lib, rc, "a1mpc_pipeline_submit_ticks_device")
        return int(k.value)

    def submit_strided(self, x0, xref, R, foot, foot_stride, contact, contact_stride, out, yaw_A=None, slot=-1, fresh=True):
        """a1mpc_pipeline_submit_strided: host arrays in (a1mpc_solve_batch_strided's layouts; snapshotted before the call returns), host arrays out like submit()"""
        n = int(x0.shape[0])
        x0 = np.ascontiguousarray(x0, np.float64); xref = np.ascontiguousarray(xref, np.float64); R = np.ascontiguousarray(R, np.float64)
        foot = np.ascontiguousarray(foot, np.float64); contact = np.ascontiguousarray(contact, np.uint8)
        ya = None if yaw_A is None else np.ascontiguousarray(yaw_A, np.float64)
        assert foot.size == n * (NU * self.horizon if foot_stride else NU) and contact.size == n * (4 * self.horizon if contact_stride else 4)
        need = {"grf": (np.float64, n * NU), "u": (np.float64, n * NU * self.horizon), "iters": (np.int32, n), "status": (np.int32, n)}
        for k_, a_ in out.items():
            if a_ is None:
                continue
            if k_ not in need:
                raise ValueError(f"unknown output array {k_!r}")
            dt_, size_ = need[k_]
            if not (isinstance(a_, np.ndarray) and a_.flags["C_CONTIGUOUS"] and a_.dtype == dt_ and a_.size >= size_):
                raise ValueError(f"output array {k_!r} must be a C-contiguous numpy array of {np.dtype(dt_).name} with at least {size_} elements")
        if out.get("grf") is None:
            raise ValueError("output array 'grf' is required")
        k = C.c_int32(-1)
        rc = self.lib.a1mpc_pipeline_submit_strided(self._p, int(slot), 1 if fresh else 0, n, _dp(x0), _dp(xref), _dp(R), _dp(foot), int(foot_stride), _u8p(contact), int(contact_stride),
                                                    _dp(ya), _dp(out["grf"]), _dp(out["u"]) if out.get("u") is not None else None,
                                                    _ip(out["iters"]) if out.get("iters") is not None else None, _ip(out["status"]) if out.get("status") is not None else None, C.byref(k))
        _check(self.lib, rc, "a1mpc_pipeline_submit_strided")
        self._keep = getattr(self, "_keep", {}); self._keep[int(k.value)] = out   # the output arrays must outlive the slot's batch
        return int(k.value)

    def wait(self, slot=-1):
        _check(self.lib, self.lib.a1mpc_pipeline_wait(self._p, int(slot)), "a1mpc_pipeline_wait")

    def join(self, stream, slot=-1):
        _check(self.lib, self.lib.a1mpc_pipeline_join(self._p, int(slot), C.c_void_p(int(stream)) if stream else None), "a1mpc_pipeline_join")

    def handle(self, slot):
        h = C.c_void_p()
        _check(self.lib, self.lib.a1mpc_pipeline_handle(self._p, int(slot), C.byref(h)), "a1mpc_pipeline_handle")
        return h

    def last_nfact(self, slot, n):
        out = np.zeros(int(n), np.int32)
        _check(self.lib, self.lib.a1mpc_last_nfact(self.handle(slot), int(n), _ip(out)), "a1mpc_last_nfact")
        return out

    def close(self):
        if getattr(self, "_p", None) is not None and self._p:
            self.lib.a1mpc_pipeline_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class ShardedEngine:
    """a1mpc_sharded_*: one handle, the batch cut into contiguous shards over `devices` (None = all visible), transport 0 = pinned copies, 1 = RCCL"""

    def __init__(self, cfg, max_batch, devices=None, transport=0):
        self.lib = load_library(); self.horizon = int(cfg.horizon); self._h = C.c_void_p()
        dv = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        _check(self.lib, self.lib.a1mpc_sharded_create(C.byref(cfg), int(max_batch), _ip(dv), 0 if dv is None else len(dv), int(transport), C.byref(self._h)), "a1mpc_sharded_create")

    def info(self):
        n = C.c_int32(); t = C.c_int32(); d = np.zeros(64, np.int32)
        _check(self.lib, self.lib.a1mpc_sharded_info(self._h, C.byref(n), _ip(d), C.byref(t)), "a1mpc_sharded_info")
        return dict(n_shards=n.value, devices=d[:n.value].tolist(), transport=t.value)

    def solve(self, x0, xref, R, foot, contact):
        h = self.horizon
        x0 = _f64(x0, (-1, NS)); n = x0.shape[0]
        xref = _f64(xref, (n, NS * h)); R = _f64(R, (n, 9)); foot = _f64(foot, (n, 12)); contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(n, 4)
        grf = np.zeros((n, 12)); iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
        rc = self.lib.a1mpc_sharded_solve_batch(self._h, n, _dp(x0), _dp(xref), _dp(R), _dp(foot), contact.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(grf), _ip(iters), _ip(status))
        _check(self.lib, rc, "a1mpc_sharded_solve_batch")
        return dict(grf=grf, iters=iters, status=status)

    def solve_ticks(self, tick, R, foot, contact):
        """a1mpc_sharded_solve_batch_ticks: the compact 22-number tick records, host arrays"""
        tick = _f64(tick, (-1, 22)); n = tick.shape[0]
        R = _f64(R, (n, 9)); foot = _f64(foot, (n, 12)); contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(n, 4)
        grf = np.zeros((n, 12)); iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
        rc = self.lib.a1mpc_sharded_solve_batch_ticks(self._h, n, _dp(tick), _dp(R), _dp(foot), contact.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(grf), _ip(iters), _ip(status))
        _check(self.lib, rc, "a1mpc_sharded_solve_batch_ticks")
        return dict(grf=grf, iters=iters, status=status)

    def solve_device(self, n, d_x0, d_xref, d_R, d_foot, d_contact, d_grf, d_iters=None, d_status=None, stream=None):
        """a1mpc_sharded_solve_batch_device: the batch resident on shard 0's GPU (torch tensors of that device)"""
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        rc = self.lib.a1mpc_sharded_solve_batch_device(self._h, int(n), ptr(d_x0), ptr(d_xref), ptr(d_R), ptr(d_foot), ptr(d_contact), ptr(d_grf), ptr(d_iters), ptr(d_status),
                                                       C.c_void_p(int(stream)) if stream else None)
        _check(self.lib, rc, "a1mpc_sharded_solve_batch_device")

    def solve_ticks_device(self, n, d_tick, d_R, d_foot, d_contact, d_grf, d_iters=None, d_status=None, stream=None):
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        rc = self.lib.a1mpc_sharded_solve_batch_ticks_device(self._h, int(n), ptr(d_tick), ptr(d_R), ptr(d_foot), ptr(d_contact), ptr(d_grf), ptr(d_iters), ptr(d_status),
                                                             C.c_void_p(int(stream)) if stream else None)
        _check(self.lib, rc, "a1mpc_sharded_solve_batch_ticks_device")

    def last_transfer(self):
        a = C.c_int64(); b = C.c_int64()
        _check(self.lib, self.lib.a1mpc_sharded_last_transfer(self._h, C.byref(a), C.byref(b)), "a1mpc_sharded_last_transfer")
        return dict(scatter_bytes=int(a.value), gather_bytes=int(b.value))

    def shard_handle(self, g):
        h = C.c_void_p()
        _check(self.lib, self.lib.a1mpc_sharded_handle(self._h, int(g), C.byref(h)), "a1mpc_sharded_handle")
        return h

    def reset_warm_start(self):
        n = self.info()["n_shards"]
        for g in range(n):
            _check(self.lib, self.lib.a1mpc_reset_warm_start(self.shard_handle(g)), "a1mpc_reset_warm_start")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.a1mpc_sharded_destroy(self._h); self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def algorithmic_flops(h, iters, nfact):
    """F(h, iters, nfact) of the structured dense-condensed algorithm, per solve (numpy-broadcastable)."""
    iters = np.asarray(iters, dtype=np.float64); nfact = np.asarray(nfact, dtype=np.float64)
    f_cond = 3744.0 * h * (h + 1) * (h + 2) / 6.0 + 4056.0 * (h - 1) + 312.0 * h * h + 364.0 * h
    f_fact = 576.0 * h ** 3
    f_iter = 288.0 * h * h + 464.0 * h
    f_chk = 288.0 * h * h + 336.0 * h
    return f_cond + nfact * f_fact + iters * f_iter + np.ceil(iters / 25.0) * f_chk


def algorithmic_bytes(h):
    """Compulsory HBM bytes per solve: input record + 12 GRF doubles."""
    return 8 * (38 + 13 * h) + 96
