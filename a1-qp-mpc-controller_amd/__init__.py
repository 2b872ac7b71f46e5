"""a1-qp-mpc-controller_amd -- MI355X-native batched convex-MPC QP engine (hot path of
ShuoYangRobotics/A1-QP-MPC-Controller: ConvexMpc formation + the OSQP solve inside
A1RobotControl::compute_grf).  The directory name is not a Python identifier; load it with
`importlib` under the module name `a1_qp_mpc_controller_amd` (see __graft_entry__.load_package)."""
from . import build, engine, scenarios, sharding  # noqa: F401
from .engine import (A1MpcError, BalanceConfig, Config, Engine, GaitConfig, Pipeline, ShardedEngine, algorithmic_bytes, algorithmic_flops, load_library,  # noqa: F401
                     make_config)
