"""Host-side mirror of the RocketQuat model plugin's configuration half
(scpp_models/src/rocketQuat.cpp:203-289, scpp_models/include/common.hpp:30-38).  The flow map itself is
device code (scpp_amd/csrc/model_rocketquat.h)."""
import math
import os

import numpy as np

from ._lib import MODEL_LANDER3DOF, MODEL_ROCKET2D, MODEL_ROCKETQUAT, Lander3dofParams, Rocket2dParams, RocketQuatParams, ScppHipError
from .parameter_server import ParameterServer

CONFIG_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")

_MASK = (1 << 64) - 1


def counter_uniform(seed, instance, draw):
    """Counter-based uniform in [-1,1): SplitMix64 keyed by (seed, instance, draw)."""
    z = (seed + (instance * 8 + draw + 1) * 0x9E3779B97F4A7C15) & _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    z = z ^ (z >> 31)
    return 2.0 * ((z >> 11) * (1.0 / 9007199254740992.0)) - 1.0


def euler_to_quaternion_xyz(eta):
    """q = Rx(phi) Ry(theta) Rz(psi) as (w,x,y,z)  (common.hpp:30-38)."""
    cx, sx = math.cos(0.5 * eta[0]), math.sin(0.5 * eta[0])
    cy, sy = math.cos(0.5 * eta[1]), math.sin(0.5 * eta[1])
    cz, sz = math.cos(0.5 * eta[2]), math.sin(0.5 * eta[2])
    aw, ax, ay, az = cx * cy, sx * cy, cx * sy, sx * sy
    return [aw * cz - az * sz, ax * cz + ay * sz, ay * cz - ax * sz, aw * sz + az * cz]


class RocketQuat:
    modelName = "RocketQuat"
    model_id = MODEL_ROCKETQUAT
    state_dim, input_dim, param_dim = 14, 4, 10

    def sc_params(self):
        """the C-ABI parameter struct of this model for scpp_hip_sc_setup"""
        return self.p

    def __init__(self, param_folder=CONFIG_ROOT):
        self.param_folder = param_folder
        self.p = None
        self.x_init = None
        self.rpy_init = None

    def getParameterFolder(self):
        return os.path.join(self.param_folder, self.modelName)

    def loadParameters(self):
        """rocketQuat.cpp:229-289"""
        ps = ParameterServer(os.path.join(self.getParameterFolder(), "model.info"))
        d2r = math.pi / 180.0
        p = RocketQuatParams()
        g_I = ps.load_vector("g_I", 3)
        p.g_I[:] = g_I
        p.J_B[:] = ps.load_vector("J_B", 3)
        p.r_T_B[:] = ps.load_vector("r_T_B", 3)
        m_init = ps.load_scalar("m_init")
        r_init = ps.load_vector("r_init", 3)
        v_init = ps.load_vector("v_init", 3)
        rpy_init = [a * d2r for a in ps.load_vector("rpy_init", 3)]
        w_init = [a * d2r for a in ps.load_vector("w_init", 3)]
        w_final = [a * d2r for a in ps.load_vector("w_final", 3)]
        m_dry = ps.load_scalar("m_dry")
        r_final = ps.load_vector("r_final", 3)
        v_final = ps.load_vector("v_final", 3)
        rpy_final = [a * d2r for a in ps.load_vector("rpy_final", 3)]
        p.T_min = ps.load_scalar("T_min")
        p.T_max = ps.load_scalar("T_max")
        p.t_max = ps.load_scalar("t_max")
        I_sp = ps.load_scalar("I_sp")
        p.gimbal_max = ps.load_scalar("gimbal_max") * d2r
        p.theta_max = ps.load_scalar("theta_max") * d2r
        p.gamma_gs = ps.load_scalar("gamma_gs") * d2r
        p.w_B_max = ps.load_scalar("w_B_max") * d2r
        self.random_initial_state = ps.load_scalar("random_initial_state", bool)
        p.final_time = ps.load_scalar("final_time")
        p.exact_minimum_thrust = int(ps.load_scalar("exact_minimum_thrust", bool))
        p.enable_roll_control = int(ps.load_scalar("enable_roll_control", bool))
        p.alpha_m = 1.0 / (I_sp * abs(g_I[2]))
        q_init = euler_to_quaternion_xyz(rpy_init)
        q_final = euler_to_quaternion_xyz(rpy_final)
        self.x_init = np.array([m_init] + r_init + v_init + q_init + w_init, dtype=np.float64)
        p.x_final[:] = [m_dry] + r_final + v_final + q_final + w_final
        self.rpy_init = rpy_init
        self.p = p
        return self

    def randomized_initial_states(self, batch, seed=20260927, first=0):
        """The reference's (commented-out) randomizeInitialState recipe (rocketQuat.cpp:203-227), made
        deterministic and batched per SURVEY.md §8(d): 7 counter-based uniforms per instance."""
        X = np.tile(self.x_init, (batch, 1))
        for b in range(batch):
            i = first + b
            X[b, 1] *= counter_uniform(seed, i, 0)
            X[b, 2] *= counter_uniform(seed, i, 1)
            X[b, 4] *= counter_uniform(seed, i, 2)
            X[b, 5] *= counter_uniform(seed, i, 3)
            X[b, 6] *= 1.0 + 0.2 * counter_uniform(seed, i, 4)
            euler = [counter_uniform(seed, i, 5) * self.rpy_init[0], counter_uniform(seed, i, 6) * self.rpy_init[1], self.rpy_init[2]]
            X[b, 7:11] = euler_to_quaternion_xyz(euler)
        return X

    def flow_params(self, x_init=None, nondimensionalize=True):
        """getNewModelParameters (rocketQuat.cpp:168-173) after Parameters::nondimensionalize (:291-312)."""
        x = self.x_init if x_init is None else np.asarray(x_init)
        m_s = x[0] if nondimensionalize else 1.0
        r_s = float(np.linalg.norm(x[1:4])) if nondimensionalize else 1.0
        p = self.p
        return np.array([p.alpha_m * r_s] + [g / r_s for g in p.g_I] + [j / (m_s * r_s * r_s) for j in p.J_B] + [r / r_s for r in p.r_T_B])


class Rocket2DParameters:
    """Rocket2d::Parameters after loadFromFile (rocket2d.cpp:150-198): SI units, radians."""


class Rocket2D:
    """Host-side configuration half of the Rocket2d plugin (scpp_models/src/rocket2d.cpp:40-44,143-198); the flow map is
    device code (csrc/model_rocketquat.h: Rocket2dModel)."""
    modelName = "Rocket2D"
    model_id = MODEL_ROCKET2D
    state_dim, input_dim, param_dim = 6, 2, 6

    @property
    def x_init(self):
        return self.p.x_init

    def sc_params(self):
        """scpp_rocket2d_params for scpp_hip_sc_setup_rocket2d.  The device problem always carries the x_init / x_final /
        U(0, K-1) = 0 equalities (rocket2d.cpp:54-59 with constrain_initial_final true, the SC configuration of model.info:55-56);
        a model with the flag off would silently get a different problem, so it is refused (like host/rocket_2d.hpp: scSetup)."""
        if not self.p.constrain_initial_final:
            raise ScppHipError("Rocket2D: constrain_initial_final must be enabled for SC / SCvx (model.info: 'enable for SC and "
                               "disable for MPC/LQR'); the device sub-problem has the initial / final equalities built in")
        p, q = self.p, Rocket2dParams()
        q.g_I[:] = p.g_I
        q.r_T_B[:] = p.r_T_B
        q.m, q.J_B, q.T_min, q.T_max = p.m, p.J_B, p.T_min, p.T_max
        q.gimbal_max, q.theta_max, q.gamma_gs, q.w_B_max = p.gimbal_max, p.theta_max, p.gamma_gs, p.w_B_max
        q.x_final[:] = list(p.x_final)
        q.final_time = p.final_time
        return q

    def __init__(self, param_folder=CONFIG_ROOT):
        self.param_folder = param_folder
        self.p = None

    def getParameterFolder(self):
        return os.path.join(self.param_folder, self.modelName)

    def loadParameters(self):
        ps = ParameterServer(os.path.join(self.getParameterFolder(), "model.info"))
        d2r = math.pi / 180.0
        p = Rocket2DParameters()
        p.g_I = ps.load_vector("g_I", 2)
        p.J_B = ps.load_scalar("J_B")
        p.r_T_B = ps.load_vector("r_T_B", 2)
        p.m = ps.load_scalar("m")
        p.T_min, p.T_max = ps.load_scalar("T_min"), ps.load_scalar("T_max")
        p.gamma_gs = ps.load_scalar("gamma_gs") * d2r
        p.gimbal_max = ps.load_scalar("gimbal_max") * d2r
        p.theta_max = ps.load_scalar("theta_max") * d2r
        p.w_B_max = ps.load_scalar("w_B_max") * d2r
        p.final_time = ps.load_scalar("final_time")
        p.constrain_initial_final = ps.load_scalar("constrain_initial_final", bool)
        p.add_slack_variables = ps.load_scalar("add_slack_variables", bool)
        p.x_init = np.array(ps.load_vector("r_init", 2) + ps.load_vector("v_init", 2)
                            + [ps.load_scalar("eta_init") * d2r, ps.load_scalar("w_init") * d2r])
        p.x_final = np.array(ps.load_vector("r_final", 2) + ps.load_vector("v_final", 2)
                             + [ps.load_scalar("eta_final") * d2r, ps.load_scalar("w_final") * d2r])
        p.tan_gamma_gs = math.tan(p.gamma_gs)
        self.p = p
        return self

    def flow_params(self):
        """getNewModelParameters (rocket2d.cpp:143-148): par = [m, J_B, g_I, r_T_B]"""
        p = self.p
        return np.array([p.m, p.J_B, p.g_I[0], p.g_I[1], p.r_T_B[0], p.r_T_B[1]])

    def getOperatingPoint(self):
        """rocket2d.cpp:40-44.  The reference streams `0, -p.g_I * p.m` (a scalar and a 2-vector) into a 2-vector, which
        asserts / overruns; the evident intent -- hover thrust (0, -g_y m) -- is what is returned."""
        return np.zeros(6), np.array([0.0, -self.p.g_I[1] * self.p.m])

    def randomized_initial_states(self, batch, seed=20260927, first=0, spread=1.0):
        """Synthetic closed-loop start states around the shipped x_init (build-defined; the reference has no recipe for
        Rocket2D): lateral position and velocity scaled by U(-1,1), descent rate by 1 + 0.2 U, tilt by U(-1,1)."""
        out = np.zeros((batch, 6))
        for b in range(batch):
            i = first + b
            x = self.p.x_init.copy()
            x[0] *= spread * counter_uniform(seed, i, 0)
            x[2] = 0.05 * abs(x[3]) * counter_uniform(seed, i, 1)
            x[3] *= 1.0 + 0.2 * counter_uniform(seed, i, 2)
            x[4] *= counter_uniform(seed, i, 3)
            out[b] = x
        return out


class Lander3dofParameters:
    """Parameters of the Lander3dof model after loadFromFile: SI units, radians."""


class Lander3dof:
    """Host-side configuration half of the Lander3dof plugin -- NOT a model of the reference: this repository's third model (a point-mass
    powered-descent vehicle, states [m, r, v], input T), added in round 6 through the model-plugin path; the flow map, the constraint table
    and the per-instance set-up are device code (csrc/model_lander3dof.h, constraint_table.h: Lander3dofSC, sc_kernels.h: Lander3dofPlugin).
    Same call order as the reference's models: loadParameters -> (algorithm).  config/Lander3dof/{model,SC,SCvx}.info."""
    modelName = "Lander3dof"
    model_id = MODEL_LANDER3DOF
    state_dim, input_dim, param_dim = 7, 3, 4

    def __init__(self, param_folder=CONFIG_ROOT):
        self.param_folder = param_folder
        self.p = None

    @property
    def x_init(self):
        return self.p.x_init

    def getParameterFolder(self):
        return os.path.join(self.param_folder, self.modelName)

    def loadParameters(self):
        ps = ParameterServer(os.path.join(self.getParameterFolder(), "model.info"))
        d2r = math.pi / 180.0
        p = Lander3dofParameters()
        p.g_I = ps.load_vector("g_I", 3)
        p.exact_minimum_thrust = ps.load_scalar("exact_minimum_thrust", bool)
        p.alpha_m = 1.0 / (ps.load_scalar("I_sp") * abs(p.g_I[2]))
        p.T_min, p.T_max = ps.load_scalar("T_min"), ps.load_scalar("T_max")
        p.pointing_max = ps.load_scalar("pointing_max") * d2r
        p.gamma_gs = ps.load_scalar("gamma_gs") * d2r
        p.final_time = ps.load_scalar("final_time")
        p.x_init = np.array([ps.load_scalar("m_init")] + ps.load_vector("r_init", 3) + ps.load_vector("v_init", 3))
        p.x_final = np.array([ps.load_scalar("m_dry")] + ps.load_vector("r_final", 3) + ps.load_vector("v_final", 3))
        self.p = p
        return self

    def sc_params(self):
        """scpp_lander3dof_params for scpp_hip_sc_setup_lander3dof / scpp_hip_scvx_setup_lander3dof"""
        p, q = self.p, Lander3dofParams()
        q.exact_minimum_thrust = int(p.exact_minimum_thrust)
        q.g_I[:] = p.g_I
        q.alpha_m, q.T_min, q.T_max, q.pointing_max, q.gamma_gs = p.alpha_m, p.T_min, p.T_max, p.pointing_max, p.gamma_gs
        q.x_final[:] = list(p.x_final)
        q.final_time = p.final_time
        return q

    def flow_params(self, nondimensionalize=False, x_init=None):
        """par = [alpha_m, g_I]; nondimensionalize=True: in the units of the instance (m_scale = m_init, r_scale = |r_init|)"""
        x = self.x_init if x_init is None else np.asarray(x_init)
        r_s = float(np.linalg.norm(x[1:4])) if nondimensionalize else 1.0
        return np.array([self.p.alpha_m * r_s] + [g / r_s for g in self.p.g_I])

    def randomized_initial_states(self, batch, seed=20260927, first=0):
        """Synthetic start states around the shipped x_init, RocketQuat's recipe without the attitude: lateral position and velocity
        scaled by U(-1,1), descent rate by 1 + 0.2 U; mass and altitude unchanged."""
        out = np.zeros((batch, 7))
        for b in range(batch):
            i = first + b
            x = self.p.x_init.copy()
            x[1] *= counter_uniform(seed, i, 0)
            x[2] *= counter_uniform(seed, i, 1)
            x[4] *= counter_uniform(seed, i, 2)
            x[5] *= counter_uniform(seed, i, 3)
            x[6] *= 1.0 + 0.2 * counter_uniform(seed, i, 4)
            out[b] = x
        return out
