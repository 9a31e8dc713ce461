"""Host-side mirror of scpp::MPCAlgorithm (scpp_core/include/MPCAlgorithm.hpp:17-64, src/MPCAlgorithm.cpp:11-139) and of the
MPC_sim driver (scpp/src/MPC_sim.cpp:16-86) over the C ABI, batched: every call handles B independent controllers / closed
loops.  All arithmetic runs in the HIP library (scpp_hip_mpc_*); there is no CPU fallback."""
import math
import os

import numpy as np

from ._lib import MODEL_ROCKET2D, Context, MpcOpts
from .parameter_server import ParameterServer


class MPCAlgorithm:
    def __init__(self, model, batch_max=256, device=0, library=None):
        self.model = model
        self.batch_max, self.device, self.library = batch_max, device, library
        self.ctx = None
        self.initialized = False
        self.x_init = None
        self.x_final = None
        self.loadParameters()

    def loadParameters(self):
        """MPCAlgorithm.cpp:17-32"""
        ps = ParameterServer(os.path.join(self.model.getParameterFolder(), "MPC.info"))
        self.K = int(ps.load_scalar("K"))
        self.nondimensionalize = ps.load_scalar("nondimensionalize", bool)
        self.constant_dynamics = ps.load_scalar("constant_dynamics", bool)
        self.intermediate_cost_active = ps.load_scalar("intermediate_cost_active", bool)
        self.time_horizon = ps.load_scalar("time_horizon")
        self.setStateWeights(ps.load_vector("state_weights_intermediate", 6), ps.load_vector("state_weights_terminal", 6))
        self.setInputWeights(ps.load_vector("input_weights", 2))

    def setStateWeights(self, intermediate, terminal):
        self.state_weights_intermediate = np.asarray(intermediate, dtype=np.float64)
        self.state_weights_terminal = np.asarray(terminal, dtype=np.float64)

    def setInputWeights(self, intermediate):
        self.input_weights = np.asarray(intermediate, dtype=np.float64)

    def initialize(self, feastol=1e-8, abstol=1e-8, reltol=1e-8, maxit=50):
        """MPCAlgorithm.cpp:34-69: linearise at the operating point, discretise exactly, build the problem (once)."""
        p = self.model.p
        if p.constrain_initial_final:
            raise RuntimeError("constrain_initial_final must be disabled for MPC (config/Rocket2D/model.info: "
                               "'enable for SC and disable for MPC/LQR')")
        o = MpcOpts()
        o.K = self.K
        o.nondimensionalize = int(self.nondimensionalize)
        o.constant_dynamics = int(self.constant_dynamics)
        o.intermediate_cost_active = int(self.intermediate_cost_active)
        o.time_horizon = self.time_horizon
        o.state_weights_intermediate[:] = list(self.state_weights_intermediate)
        o.state_weights_terminal[:] = list(self.state_weights_terminal)
        o.input_weights[:] = list(self.input_weights)
        x_eq, u_eq = self.model.getOperatingPoint()
        o.x_eq[:] = list(x_eq)
        o.u_eq[:] = list(u_eq)
        o.tan_gamma_gs, o.theta_max, o.w_B_max = p.tan_gamma_gs, p.theta_max, p.w_B_max
        o.gimbal_max, o.T_min, o.T_max = p.gimbal_max, p.T_min, p.T_max
        o.x_scale_ref = math.hypot(p.x_init[0], p.x_init[1])
        o.feastol, o.abstol, o.reltol, o.maxit = feastol, abstol, reltol, maxit
        self.ctx = Context(MODEL_ROCKET2D, K=max(self.K, 3), batch_max=self.batch_max, device=self.device, library=self.library)
        self.ctx.mpc_setup(o, self.model.flow_params())
        self.A, self.B, self.z = self.ctx.mpc_model()
        self.initialized = True
        return self

    def setInitialState(self, x):
        self.x_init = np.asarray(x, dtype=np.float64).reshape(-1, 6)

    def setFinalState(self, x):
        self.x_final = np.asarray(x, dtype=np.float64).reshape(-1, 6)

    def solve(self):
        """MPCAlgorithm.cpp:95-121 for every row of x_init; returns the number of controllers whose solve succeeded."""
        assert self.initialized
        return self.ctx.mpc_solve(self.x_init, self.x_final)

    def getSolution(self):
        """X [B][K][6], U [B][K-1][2] plus cost [B][2] = (input_cost, error_cost), status, iters"""
        return self.ctx.mpc_download()


class MPCSim:
    """MPC_sim.cpp:16-86 for B closed loops on the device (deterministic plant step, see include/scpp_hip.h)."""

    def __init__(self, algorithm, sim_time=15.0, min_timestep=0.010, stop_tol=0.02, max_steps=0):
        self.alg = algorithm
        self.sim_time, self.min_timestep, self.stop_tol, self.max_steps = sim_time, min_timestep, stop_tol, max_steps

    def run(self, x_start, x_final=None):
        x_final = self.alg.model.p.x_final if x_final is None else x_final
        return self.alg.ctx.mpc_sim(x_start, x_final, self.min_timestep, self.sim_time, self.stop_tol, self.max_steps)
