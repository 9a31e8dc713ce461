"""ctypes binding of the C ABI declared in include/scpp_hip.h."""
import ctypes as C
import os

import numpy as np

MODEL_ROCKETQUAT, MODEL_ROCKET2D, MODEL_LANDER3DOF = 0, 1, 2
MODE_FOH, MODE_VT = 1, 2
IPM_RESIDENT, IPM_SPLIT, IPM_RESIDENT_WS = 0, 1, 2
STREAM_POOLS, STREAM_PERSISTENT = 0, 1
# include/scpp_hip.h: return codes, and the per-instance status of an SCvx run retired in the reference's exit-less reject loop
E_ARG, E_HIP, E_UNSUPPORTED, E_STATE = -1, -2, -3, -4
# What this binding was written against.  load_library() asks the library for ITS values (scpp_hip_query) and refuses one that disagrees: the
# status moved from -4 to -5 between two builds once, and a constant that is only written down on both sides is how that goes unnoticed.
ABI_REVISION = 7
STATUS_REJECTION_CAP = -5
SCVX_SOLVE_CAP = 64  # csrc/scvx_kernels.h: sub-problem solves per configured iteration before an instance is retired
Q_ABI_REVISION, Q_STATUS_REJECTION_CAP, Q_SCVX_SOLVE_CAP, Q_MAX_K, Q_MPC_MAX_K = 0, 1, 2, 3, 4

_HERE = os.path.dirname(os.path.abspath(__file__))


class ScppHipError(RuntimeError):
    pass


class RocketQuatParams(C.Structure):
    """scpp_rocketquat_params (RocketQuat::Parameters after loadFromFile, rocketQuat.cpp:234-289)."""

    _fields_ = [
        ("g_I", C.c_double * 3),
        ("J_B", C.c_double * 3),
        ("r_T_B", C.c_double * 3),
        ("alpha_m", C.c_double),
        ("T_min", C.c_double),
        ("T_max", C.c_double),
        ("t_max", C.c_double),
        ("gimbal_max", C.c_double),
        ("theta_max", C.c_double),
        ("gamma_gs", C.c_double),
        ("w_B_max", C.c_double),
        ("x_final", C.c_double * 14),
        ("final_time", C.c_double),
        ("exact_minimum_thrust", C.c_int),
        ("enable_roll_control", C.c_int),
    ]


class Rocket2dParams(C.Structure):
    """scpp_rocket2d_params (Rocket2d::Parameters after loadFromFile, rocket2d.cpp:150-198)."""

    _fields_ = [
        ("g_I", C.c_double * 2), ("r_T_B", C.c_double * 2), ("m", C.c_double), ("J_B", C.c_double),
        ("T_min", C.c_double), ("T_max", C.c_double),
        ("gimbal_max", C.c_double), ("theta_max", C.c_double), ("gamma_gs", C.c_double), ("w_B_max", C.c_double),
        ("x_final", C.c_double * 6), ("final_time", C.c_double),
    ]


class Lander3dofParams(C.Structure):
    """scpp_lander3dof_params (this repository's third model: csrc/model_lander3dof.h; not a model of the reference)."""

    _fields_ = [
        ("exact_minimum_thrust", C.c_int), ("g_I", C.c_double * 3),
        ("alpha_m", C.c_double), ("T_min", C.c_double), ("T_max", C.c_double), ("pointing_max", C.c_double), ("gamma_gs", C.c_double),
        ("x_final", C.c_double * 7), ("final_time", C.c_double),
    ]


class SCOpts(C.Structure):
    """scpp_sc_opts (SC.info, SCAlgorithm.cpp:22-46)."""

    _fields_ = [
        ("K", C.c_int),
        ("free_final_time", C.c_int),
        ("interpolate_input", C.c_int),
        ("nondimensionalize", C.c_int),
        ("max_iterations", C.c_int),
        ("weight_time", C.c_double),
        ("weight_trust_region_time", C.c_double),
        ("weight_trust_region_trajectory", C.c_double),
        ("weight_virtual_control", C.c_double),
        ("nu_tol", C.c_double),
        ("delta_tol", C.c_double),
    ]


class SCvxOpts(C.Structure):
    """scpp_scvx_opts (SCvx.info, SCvxAlgorithm.cpp:22-44)"""
    _fields_ = [
        ("K", C.c_int), ("interpolate_input", C.c_int), ("nondimensionalize", C.c_int), ("max_iterations", C.c_int),
        ("alpha", C.c_double), ("beta", C.c_double), ("rho_0", C.c_double), ("rho_1", C.c_double), ("rho_2", C.c_double),
        ("change_threshold", C.c_double), ("weight_virtual_control", C.c_double), ("trust_region", C.c_double),
    ]


class MpcOpts(C.Structure):
    """scpp_mpc_opts (MPC.info, MPCAlgorithm.cpp:17-32, + the Rocket2d data of rocket2d.cpp:40-84)"""
    _fields_ = [
        ("K", C.c_int32), ("nondimensionalize", C.c_int32), ("constant_dynamics", C.c_int32), ("intermediate_cost_active", C.c_int32),
        ("time_horizon", C.c_double),
        ("state_weights_intermediate", C.c_double * 6), ("state_weights_terminal", C.c_double * 6), ("input_weights", C.c_double * 2),
        ("x_eq", C.c_double * 6), ("u_eq", C.c_double * 2),
        ("tan_gamma_gs", C.c_double), ("theta_max", C.c_double), ("w_B_max", C.c_double), ("gimbal_max", C.c_double),
        ("T_min", C.c_double), ("T_max", C.c_double), ("x_scale_ref", C.c_double),
        ("feastol", C.c_double), ("abstol", C.c_double), ("reltol", C.c_double), ("maxit", C.c_int32),
    ]


class SocpOpts(C.Structure):
    _fields_ = [
        ("feastol", C.c_double),
        ("abstol", C.c_double),
        ("reltol", C.c_double),
        ("maxit", C.c_int),
        ("use_mfma", C.c_int),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("ms_discretize", C.c_double),
        ("ms_socp", C.c_double),
        ("ms_other", C.c_double),
        ("n_discretize", C.c_longlong),
        ("n_socp", C.c_longlong),
        ("inst_discretize", C.c_longlong),
        ("inst_socp", C.c_longlong),
        ("ms_discretize_union", C.c_double),
        ("ms_socp_union", C.c_double),
    ]


_lib = None
_lib_path = None

SYMBOLS = [
    "scpp_hip_create", "scpp_hip_destroy", "scpp_hip_version", "scpp_hip_query", "scpp_hip_set_flow_params", "scpp_hip_upload_traj", "scpp_hip_upload_traj_zoh",
    "scpp_hip_discretize", "scpp_hip_set_discretization_steps", "scpp_hip_download_dd", "scpp_hip_simulate", "scpp_hip_set_socp_opts", "scpp_hip_set_ipm_schedule", "scpp_hip_set_stream_engine", "scpp_hip_stream_profile", "scpp_hip_sc_setup", "scpp_hip_sc_setup_rocket2d",
    "scpp_hip_sc_set_active", "scpp_hip_sc_iterate", "scpp_hip_sc_solve", "scpp_hip_sc_finish", "scpp_hip_scvx_setup", "scpp_hip_scvx_solve", "scpp_hip_scvx_download_state", "scpp_hip_socp_solve", "scpp_hip_download", "scpp_hip_download_socp_info",
    "scpp_hip_get_timing", "scpp_hip_device_ptrs", "scpp_hip_synchronize",
    "scpp_hip_mpc_setup", "scpp_hip_mpc_get_model", "scpp_hip_mpc_solve", "scpp_hip_mpc_download", "scpp_hip_mpc_sim",
    "scpp_hip_mpc_sim_download",
    "scpp_hip_scvx_solve_stream", "scpp_hip_stream_rows", "scpp_hip_stream_download", "scpp_hip_stream_info",
    "scpp_hip_scvx_setup_rocket2d", "scpp_hip_scvx_solve_stream_rocket2d",
    "scpp_hip_sc_setup_lander3dof", "scpp_hip_scvx_setup_lander3dof", "scpp_hip_scvx_solve_stream_lander3dof",
    "scpp_hip_scvx_record_iterates", "scpp_hip_scvx_download_iterates",
]


def load_library(path=None):
    """Load libscpp_hip.so (built by `python -c 'import __graft_entry__ as g; g.build()'`).

    Raises ScppHipError if the HIP library is missing: the product has no CPU fallback.  Tests of the kernel
    logic on GPU-less machines pass the path of the emulation build (tests/emu/libscpp_emu.so) explicitly.
    """
    global _lib, _lib_path
    if path is None:
        path = os.environ.get("SCPP_HIP_LIBRARY", os.path.join(_HERE, "libscpp_hip.so"))
    if _lib is not None and _lib_path == path:
        return _lib
    if not os.path.exists(path):
        raise ScppHipError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "scpp_amd has no CPU fallback."
        )
    lib = C.CDLL(path)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise ScppHipError(f"{path} does not export {s}")
    lib.scpp_hip_version.restype = C.c_char_p
    for what, name, mine in ((Q_ABI_REVISION, "ABI revision", ABI_REVISION), (Q_STATUS_REJECTION_CAP, "SCPP_STATUS_REJECTION_CAP", STATUS_REJECTION_CAP),
                             (Q_SCVX_SOLVE_CAP, "SCVX_SOLVE_CAP", SCVX_SOLVE_CAP)):
        theirs = query(what, lib)
        if theirs != mine:
            raise ScppHipError(f"{path}: {name} is {theirs}, this binding was written against {mine} ({lib.scpp_hip_version().decode()}); rebuild")
    _lib, _lib_path = lib, path
    return lib


def query(what, lib=None):
    """scpp_hip_query: a build-defined constant of the loaded library"""
    v = C.c_longlong()
    rc = (lib or load_library()).scpp_hip_query(int(what), C.byref(v))
    if rc != 0:
        raise ScppHipError(f"scpp_hip_query({what}) failed with code {rc}")
    return int(v.value)


# suffix of a model's set-up / streaming entry points in the C ABI (include/scpp_hip.h)
_MODEL_SUFFIX = {MODEL_ROCKETQUAT: "", MODEL_ROCKET2D: "_rocket2d", MODEL_LANDER3DOF: "_lander3dof"}


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc != 0:
        raise ScppHipError(f"{what} failed with code {rc}")


class Context:
    """One scpp_hip_ctx (one GPU, one stream)."""

    def __init__(self, model=MODEL_ROCKETQUAT, K=50, batch_max=256, device=0, library=None):
        self.lib = load_library(library)
        self.model, self.K, self.batch_max = model, K, batch_max
        self.nx, self.nu, self.np_ = {MODEL_ROCKETQUAT: (14, 4, 10), MODEL_ROCKET2D: (6, 2, 6), MODEL_LANDER3DOF: (7, 3, 4)}[model]
        h = C.c_void_p()
        _chk(self.lib.scpp_hip_create(C.byref(h), int(device), int(model), int(K), int(batch_max), C.c_uint(0)), "scpp_hip_create")
        self.h = h
        self.B = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.scpp_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- multipleShooting boundary ----
    def set_flow_params(self, par):
        par = np.ascontiguousarray(par, dtype=np.float64).reshape(-1, self.np_)
        _chk(self.lib.scpp_hip_set_flow_params(self.h, _p(par), int(par.shape[0])), "set_flow_params")

    def upload_traj(self, X, U, sigma):
        """U [B][K][nu] (first-order hold) or [B][K-1][nu] (zero-order hold, trajectoryData.hpp:27-32)"""
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, self.K, self.nx)
        U = np.ascontiguousarray(U, dtype=np.float64)
        U = U.reshape(X.shape[0], -1, self.nu)
        sigma = np.ascontiguousarray(sigma, dtype=np.float64).reshape(-1)
        self.B = X.shape[0]
        if U.shape[1] == self.K - 1:
            _chk(self.lib.scpp_hip_upload_traj_zoh(self.h, _p(X), _p(U), _p(sigma), int(self.B)), "upload_traj_zoh")
        else:
            assert U.shape[1] == self.K
            _chk(self.lib.scpp_hip_upload_traj(self.h, _p(X), _p(U), _p(sigma), int(self.B)), "upload_traj")

    def set_stream_engine(self, engine):
        """scpp_hip_set_stream_engine: STREAM_POOLS (0: rounds of launches per slot pool) or STREAM_PERSISTENT (1: one launch, a wavefront per slot)"""
        _chk(self.lib.scpp_hip_set_stream_engine(self.h, int(engine)), "set_stream_engine")

    def stream_profile(self):
        """per-step wavefront ticks of the last persistent job: refill, multipleShooting, sub-problem solve, cost + accept / reject"""
        t = np.zeros(4)
        _chk(self.lib.scpp_hip_stream_profile(self.h, _p(t)), "stream_profile")
        return dict(zip(("refill", "discretize", "solve", "cost"), t.tolist()))

    def set_ipm_schedule(self, schedule, split_pairs=0):
        """scpp_hip_set_ipm_schedule: IPM_RESIDENT (0), IPM_SPLIT (1: two kernels per interior-point iteration), IPM_RESIDENT_WS (2)"""
        _chk(self.lib.scpp_hip_set_ipm_schedule(self.h, int(schedule), int(split_pairs)), "set_ipm_schedule")

    def set_discretization_steps(self, steps):
        """RKF78 steps per segment: 5 = the reference's fixed count (default), 0 = the opt-in step-length rule; include/scpp_hip.h"""
        _chk(self.lib.scpp_hip_set_discretization_steps(self.h, int(steps)), "set_discretization_steps")

    def discretize(self, mode=MODE_FOH | MODE_VT):
        _chk(self.lib.scpp_hip_discretize(self.h, int(mode)), "discretize")

    def download_dd(self):
        B, K, nx, nu = self.B, self.K, self.nx, self.nu
        A = np.zeros((B, K - 1, nx, nx))
        Bm = np.zeros((B, K - 1, nx, nu))
        Cm = np.zeros((B, K - 1, nx, nu))
        S = np.zeros((B, K - 1, nx))
        Z = np.zeros((B, K - 1, nx))
        _chk(self.lib.scpp_hip_download_dd(self.h, _p(A), _p(Bm), _p(Cm), _p(S), _p(Z)), "download_dd")
        return A, Bm, Cm, S, Z

    def simulate(self, dt, u0, u1, x):
        x = np.array(x, dtype=np.float64).reshape(-1, self.nx)
        B = x.shape[0]
        dt = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (B,)))
        u0 = np.ascontiguousarray(u0, dtype=np.float64).reshape(B, self.nu)
        u1 = np.ascontiguousarray(u1, dtype=np.float64).reshape(B, self.nu)
        _chk(self.lib.scpp_hip_simulate(self.h, _p(dt), _p(u0), _p(u1), _p(x), int(B)), "simulate")
        return x

    # ---- SC boundary ----
    def set_socp_opts(self, feastol=1e-8, abstol=1e-7, reltol=1e-7, maxit=60, use_mfma=1):
        o = SocpOpts(feastol, abstol, reltol, maxit, use_mfma)
        _chk(self.lib.scpp_hip_set_socp_opts(self.h, C.byref(o)), "set_socp_opts")

    def sc_setup(self, model_params, sc_opts, x_init, warm_start=False):
        x_init = np.ascontiguousarray(x_init, dtype=np.float64).reshape(-1, self.nx)
        self.B = x_init.shape[0]
        fn = getattr(self.lib, "scpp_hip_sc_setup" + _MODEL_SUFFIX[self.model])
        _chk(fn(self.h, C.byref(model_params), C.byref(sc_opts), _p(x_init), int(self.B), int(warm_start)), "sc_setup")

    def sc_set_active(self, mask):
        mask = np.ascontiguousarray(mask, dtype=np.int32).reshape(-1)
        _chk(self.lib.scpp_hip_sc_set_active(self.h, _p(mask), int(mask.shape[0])), "sc_set_active")

    def sc_iterate(self):
        n = C.c_int(0)
        _chk(self.lib.scpp_hip_sc_iterate(self.h, C.byref(n)), "sc_iterate")
        return n.value

    # ---- SCvx boundary ----
    def scvx_setup(self, model_params, scvx_opts, x_init, warm_start=False):
        x_init = np.ascontiguousarray(x_init, dtype=np.float64).reshape(-1, self.nx)
        self.B = x_init.shape[0]
        fn = getattr(self.lib, "scpp_hip_scvx_setup" + _MODEL_SUFFIX[self.model])
        _chk(fn(self.h, C.byref(model_params), C.byref(scvx_opts), _p(x_init), int(self.B), int(warm_start)), "scvx_setup")

    def scvx_solve(self):
        n = C.c_int(0)
        _chk(self.lib.scpp_hip_scvx_solve(self.h, C.byref(n)), "scvx_solve")
        return n.value

    def scvx_state(self):
        B = self.B
        tr, cost, solves, dec = np.zeros(B), np.zeros(B), np.zeros(B, dtype=np.int32), np.zeros((B, 4))
        _chk(self.lib.scpp_hip_scvx_download_state(self.h, _p(tr), _p(cost), _p(solves), _p(dec)), "scvx_download_state")
        return dict(trust_region=tr, nonlinear_cost=cost, solves=solves, last_decision=dec)

    def scvx_record_iterates(self, enable=True):
        """opt in to SCvxAlgorithm::getAllSolutions (SCvxAlgorithm.cpp:245-260): call before scvx_setup"""
        _chk(self.lib.scpp_hip_scvx_record_iterates(self.h, int(bool(enable))), "scvx_record_iterates")

    def scvx_iterates(self, capacity, first=0, count=None):
        """(X [count][capacity][K][nx], U [count][capacity][K][nu], n [count], scalars [count][capacity][4] = radius, solves, J, decision code) of the
        recorded trajectories (dimensional): the initial trajectory and the trajectory after every iteration of the last scvx_solve; rows beyond
        n[b] are zero."""
        count = self.B - first if count is None else int(count)
        X = np.zeros((count, capacity, self.K, self.nx)); U = np.zeros((count, capacity, self.K, self.nu)); n = np.zeros(count, dtype=np.int32)
        sc = np.zeros((count, capacity, 4))
        _chk(self.lib.scpp_hip_scvx_download_iterates(self.h, int(first), count, int(capacity), _p(X), _p(U), _p(sc), _p(n)), "scvx_download_iterates")
        return X, U, n, sc

    # ---- SCvx streaming engine (continuous batching) ----
    STREAM_SCALARS = ("sigma", "nu_norm", "nonlinear_cost", "trust_region", "sc_iters", "solves", "converged", "status",
                      "ipm_iters", "instance")

    def scvx_solve_stream(self, model_params, scvx_opts, x_init, slots=0, pools=0):
        """SCvxAlgorithm::solve of every row of x_init [N][nx] through `slots` resident slots. Returns #converged."""
        x_init = np.ascontiguousarray(x_init, dtype=np.float64).reshape(-1, self.nx)
        n = C.c_int(0)
        self._stream_N = x_init.shape[0]
        fn = getattr(self.lib, "scpp_hip_scvx_solve_stream" + _MODEL_SUFFIX[self.model])
        _chk(fn(self.h, C.byref(model_params), C.byref(scvx_opts), _p(x_init), int(x_init.shape[0]), int(slots), int(pools), C.byref(n)),
             "scvx_solve_stream")
        return n.value

    def stream_rows_device(self):
        """(device pointer, doubles per row, rows) of the result rows of the last streaming job."""
        ptr, rd, n = C.c_void_p(), C.c_int(0), C.c_int(0)
        _chk(self.lib.scpp_hip_stream_rows(self.h, C.byref(ptr), C.byref(rd), C.byref(n)), "stream_rows")
        return ptr.value, rd.value, n.value

    def stream_download_rows(self, first=0, count=None):
        """raw result rows [count][K*(nx+nu) + 10] of the last streaming job (the all-gather payload)"""
        count = self._stream_N - first if count is None else count
        rowd = self.K * (self.nx + self.nu) + len(self.STREAM_SCALARS)
        rows = np.zeros((count, rowd))
        _chk(self.lib.scpp_hip_stream_download(self.h, _p(rows), int(first), int(count)), "stream_download")
        return rows

    def stream_download(self, first=0, count=None):
        return self.unpack_stream_rows(self.stream_download_rows(first, count), self.K, self.nx, self.nu)

    def stream_rounds(self):
        """rounds the host enqueued for the last streaming job and the number of slot pools it used"""
        r, p = C.c_longlong(0), C.c_int(0)
        _chk(self.lib.scpp_hip_stream_info(self.h, C.byref(r), C.byref(p)), "stream_info")
        return {"rounds": r.value, "pools": p.value}

    @classmethod
    def unpack_stream_rows(cls, rows, K, nx=14, nu=4):
        n = rows.shape[0]
        out = dict(X=rows[:, :K * nx].reshape(n, K, nx), U=rows[:, K * nx:K * (nx + nu)].reshape(n, K, nu))
        for j, name in enumerate(cls.STREAM_SCALARS):
            col = rows[:, K * (nx + nu) + j]
            out[name] = col if name in ("sigma", "nu_norm", "nonlinear_cost", "trust_region") else col.astype(np.int32)
        return out

    # ---- MPC boundary (Rocket2D) ----
    def mpc_setup(self, opts, flow_par):
        flow_par = np.ascontiguousarray(flow_par, dtype=np.float64).reshape(6)
        self._mpc_K = int(opts.K)
        _chk(self.lib.scpp_hip_mpc_setup(self.h, C.byref(opts), _p(flow_par)), "mpc_setup")

    def mpc_model(self):
        A, Bm, z = np.zeros((6, 6)), np.zeros((6, 2)), np.zeros(6)
        _chk(self.lib.scpp_hip_mpc_get_model(self.h, _p(A), _p(Bm), _p(z)), "mpc_get_model")
        return A, Bm, z

    def mpc_solve(self, x_init, x_final):
        x_init = np.ascontiguousarray(x_init, dtype=np.float64).reshape(-1, 6)
        B = x_init.shape[0]
        x_final = np.ascontiguousarray(np.broadcast_to(np.asarray(x_final, dtype=np.float64).reshape(-1, 6), (B, 6)))
        n = C.c_int(0)
        _chk(self.lib.scpp_hip_mpc_solve(self.h, _p(x_init), _p(x_final), int(B), C.byref(n)), "mpc_solve")
        self._mpc_B = B
        return n.value

    def mpc_download(self):
        B, K = self._mpc_B, self._mpc_K
        out = dict(X=np.zeros((B, K, 6)), U=np.zeros((B, K - 1, 2)), cost=np.zeros((B, 2)), status=np.zeros(B, dtype=np.int32),
                   iters=np.zeros(B, dtype=np.int32))
        _chk(self.lib.scpp_hip_mpc_download(self.h, _p(out["X"]), _p(out["U"]), _p(out["cost"]), _p(out["status"]), _p(out["iters"])),
             "mpc_download")
        return out

    def mpc_sim(self, x_start, x_final, time_step=0.010, sim_time=15.0, stop_tol=0.02, max_steps=0):
        x_start = np.ascontiguousarray(x_start, dtype=np.float64).reshape(-1, 6)
        B = x_start.shape[0]
        x_final = np.ascontiguousarray(np.broadcast_to(np.asarray(x_final, dtype=np.float64).reshape(-1, 6), (B, 6)))
        n = C.c_int(0)
        _chk(self.lib.scpp_hip_mpc_sim(self.h, _p(x_start), _p(x_final), int(B), C.c_double(time_step), C.c_double(sim_time),
                                       C.c_double(stop_tol), int(max_steps), C.byref(n)), "mpc_sim")
        self._mpc_B = B
        out = dict(x=np.zeros((B, 6)), u=np.zeros((B, 2)), t=np.zeros(B), steps=np.zeros(B, dtype=np.int32),
                   failed_solves=np.zeros(B, dtype=np.int32), ipm_iters=np.zeros(B, dtype=np.int32), reached=np.zeros(B, dtype=np.int32))
        _chk(self.lib.scpp_hip_mpc_sim_download(self.h, _p(out["x"]), _p(out["u"]), _p(out["t"]), _p(out["steps"]),
                                                _p(out["failed_solves"]), _p(out["ipm_iters"]), _p(out["reached"])), "mpc_sim_download")
        out["n_reached"] = n.value
        return out

    def sc_finish(self):
        n = C.c_int(0)
        _chk(self.lib.scpp_hip_sc_finish(self.h, C.byref(n)), "sc_finish")
        return n.value

    def sc_solve(self):
        n = C.c_int(0)
        _chk(self.lib.scpp_hip_sc_solve(self.h, C.byref(n)), "sc_solve")
        return n.value

    def socp_solve(self):
        _chk(self.lib.scpp_hip_socp_solve(self.h), "socp_solve")

    _DOWNLOAD_FIELDS = ("X", "U", "sigma", "sc_iters", "nu_norm", "converged", "status", "ipm_iters", "sum_delta")

    def download(self, fields=None, out=None):
        """Results of the batch (scpp_hip_download).  `fields`: the subset to copy (default: everything) -- the entry point skips a NULL
        destination, and a receding-horizon driver that needs U, sigma and status only does not have to move 23 MB of states per step
        (SC_sim at size: 4096 x 50 x 14 doubles).  `out`: a dict returned by an earlier call with the same batch size, reused instead of
        allocating (fresh pages cost more than the copy)."""
        B, K = self.B, self.K
        fields = self._DOWNLOAD_FIELDS if fields is None else tuple(fields)
        shapes = dict(X=((B, K, self.nx), np.float64), U=((B, K, self.nu), np.float64), sigma=((B,), np.float64), sc_iters=((B,), np.int32),
                      nu_norm=((B,), np.float64), converged=((B,), np.int32), status=((B,), np.int32), ipm_iters=((B,), np.int32),
                      sum_delta=((B,), np.float64))
        res = {}
        for f in fields:
            shp, dt = shapes[f]
            a = out.get(f) if out is not None else None
            res[f] = a if (a is not None and a.shape == shp and a.dtype == dt and a.flags.c_contiguous) else np.zeros(shp, dtype=dt)
        _chk(self.lib.scpp_hip_download(self.h, *[_p(res.get(f)) for f in self._DOWNLOAD_FIELDS]), "download")
        return res

    def socp_info(self):
        info = np.zeros((self.B, 32))
        _chk(self.lib.scpp_hip_download_socp_info(self.h, _p(info)), "download_socp_info")
        return info

    def timing(self, reset=False):
        t = Timing()
        _chk(self.lib.scpp_hip_get_timing(self.h, C.byref(t), int(reset)), "get_timing")
        return {f[0]: getattr(t, f[0]) for f in Timing._fields_}

    def device_ptrs(self):
        X, U, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(self.lib.scpp_hip_device_ptrs(self.h, C.byref(X), C.byref(U), C.byref(s)), "device_ptrs")
        return X.value, U.value, s.value

    def synchronize(self):
        _chk(self.lib.scpp_hip_synchronize(self.h), "synchronize")
