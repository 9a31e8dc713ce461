"""scpp_amd -- MI355X (gfx950) batched Successive-Convexification engine.

Python here is PLUMBING only (ctypes over the C ABI in include/scpp_hip.h, config loading, torch.distributed
for the multi-GPU gather).  All compute runs in hand-written HIP kernels inside libscpp_hip.so; the package
raises if that library is missing -- there is no CPU fallback.
"""
from ._lib import (  # noqa: F401
    MODEL_LANDER3DOF,
    MODEL_ROCKET2D,
    MODEL_ROCKETQUAT,
    MODE_FOH,
    MODE_VT,
    STATUS_REJECTION_CAP,
    SCVX_SOLVE_CAP,
    RocketQuatParams,
    SCOpts,
    SocpOpts,
    Context,
    ScppHipError,
    load_library,
)
from .parameter_server import ParameterServer  # noqa: F401
from .models import Lander3dof, Rocket2D, RocketQuat, counter_uniform  # noqa: F401
from .sc_algorithm import SCAlgorithm, load_sc_opts  # noqa: F401
from .sc_sim import SCSim, interpolated_input  # noqa: F401
from .scvx_algorithm import SCvxAlgorithm, load_scvx_opts  # noqa: F401
from .mpc_algorithm import MPCAlgorithm, MPCSim  # noqa: F401
from ._lib import MpcOpts  # noqa: F401
