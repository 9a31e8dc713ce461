"""Batched mirror of the reference's SCAlgorithm front end (scpp_core/include/SCAlgorithm.hpp:17-45):
same call order  loadParameters -> initialize -> solve(warm_start) -> getSolution, with a batch of initial states."""
import os

import numpy as np

from ._lib import Context, SCOpts
from .parameter_server import ParameterServer


def load_sc_opts(param_folder, K=None):
    """SCAlgorithm::loadParameters (SCAlgorithm.cpp:22-46)."""
    ps = ParameterServer(os.path.join(param_folder, "SC.info"))
    o = SCOpts()
    o.K = ps.load_scalar("K", int) if K is None else int(K)
    o.free_final_time = int(ps.load_scalar("free_final_time", bool))
    o.nondimensionalize = int(ps.load_scalar("nondimensionalize", bool))
    o.delta_tol = ps.load_scalar("delta_tol")
    o.max_iterations = ps.load_scalar("max_iterations", int)
    o.nu_tol = ps.load_scalar("nu_tol")
    o.weight_time = ps.load_scalar("weight_time")
    o.weight_virtual_control = ps.load_scalar("weight_virtual_control")
    o.weight_trust_region_trajectory = ps.load_scalar("weight_trust_region_trajectory")
    o.interpolate_input = int(ps.load_scalar("interpolate_input", bool))
    o.weight_trust_region_time = ps.load_scalar("weight_trust_region_time") if o.free_final_time else 0.0
    return o


class SCAlgorithm:
    def __init__(self, model, K=None, batch_max=1, device=0, library=None):
        self.model = model
        self.opts = load_sc_opts(model.getParameterFolder(), K)
        self.batch_max = batch_max
        self.device = device
        self.library = library
        self.ctx = None
        self._warm = False

    def initialize(self):
        """SCAlgorithm::initialize (SCAlgorithm.cpp:48-64): allocates the device context."""
        self.ctx = Context(self.model.model_id, self.opts.K, self.batch_max, self.device, self.library)
        return self

    def solve(self, x_init=None, warm_start=False):
        """SCAlgorithm::solve for every row of x_init [B][state_dim] (dimensional). Returns #converged."""
        if x_init is None:
            x_init = self.model.x_init[None, :]
        x_init = np.atleast_2d(np.asarray(x_init, dtype=np.float64))
        if not warm_start:
            self.opts = load_sc_opts(self.model.getParameterFolder(), self.opts.K)  # loadParameters() on cold start
        self.ctx.sc_setup(self.model.sc_params(), self.opts, x_init, warm_start=warm_start)
        return self.ctx.sc_solve()

    def getSolution(self):
        return self.ctx.download()
