"""mpc_local_planner_amd -- MI355X-native batched receding-horizon NLP solve behind the
Controller::step() surface of rst-tu-dortmund/mpc_local_planner.

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of include/mpc_hip.h) and the
host-side mirror of the reference's controller interface.  See DESIGN.md.
"""
from ._abi import (MpcConfig, make_config, config_carlike_min_time, config_unicycle_quadratic,  # noqa: F401
                   config_bicycle_min_time, STATUS_NAMES, OBJ_MIN_TIME, OBJ_QUADRATIC, OBJ_MIN_TIME_VIA_POINTS)
from .solver import BatchSolver, BatchResult, MpcError  # noqa: F401
from . import workloads  # noqa: F401
from . import params  # noqa: F401  (the reference's parameter set -> mpc_config)
from .params import config_from_params, config_from_yaml  # noqa: F401
