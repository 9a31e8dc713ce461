// C-ABI implementation (see include/scpp_hip.h). Compiled by hipcc for gfx950 (product) or, with
// -DSCPP_HIP_EMU, by g++ against tests/emu/hip_emu.h (CPU-side kernel unit tests only).
#include "../../include/scpp_hip.h"

#include <algorithm>
#include <cmath>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <tuple>
#include <vector>

#include "common.h"
#include "discretize_kernel.h"
#include "ipm_solve.h"
#include "ipm_split.h"
#include "model_rocketquat.h"
#include "sc_kernels.h"
#include "scvx_kernels.h"
#include "scvx_persistent.h"
#include "mpc_kernel.h"
#include "mpc_setup.h"

using namespace scpp;

#define CHECK_HIP(expr)                  \
    do                                   \
    {                                    \
        hipError_t e_ = (expr);          \
        if (e_ != hipSuccess)            \
            return SCPP_E_HIP;           \
    } while (0)

// ---- dispatch over the registered model plugins (csrc/sc_kernels.h: `Plugins` is the one list) ----
template <class L>
struct ParamsOf;
template <class... PL>
struct ParamsOf<PluginList<PL...>>
{
    using type = std::tuple<typename PL::Params...>;
};
// f(Plugin{}) for the plugin whose ID is `model`; SCPP_E_ARG for an id nobody registered
template <class F, class... PL>
int withPluginOf(PluginList<PL...>, int model, F &&f)
{
    int rc = SCPP_E_ARG;
    (void)((model == PL::ID ? (rc = f(PL{}), true) : false) || ...);
    return rc;
}
template <class F>
int withPlugin(int model, F &&f)
{
    return withPluginOf(Plugins{}, model, std::forward<F>(f));
}

struct scpp_hip_ctx
{
    int device = 0, model = 0, K = 0, Bmax = 0, B = 0;
    int nx = 0, nu = 0, np = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // second half of the batch in the pipelined SC loop
    unsigned ipm_lds_pad = 0; // measured: restricting ipm to one wave per SIMD (36 KB pad) costs more than the overlap gains
    hipEvent_t ev_skew = nullptr, ev_join = nullptr;
    // trajectory + discretization
    double *X = nullptr, *U = nullptr, *sigma = nullptr, *par = nullptr;
    double *A = nullptr, *Bm = nullptr, *C = nullptr, *S = nullptr, *Z = nullptr;
    // SC state
    double *x_init = nullptr, *ip = nullptr, *uhat = nullptr, *wtrx = nullptr, *ws = nullptr, *dbg = nullptr;
    int *ipm_warm = nullptr; // [B] the workspace holds a warm-startable point
    int *active = nullptr, *converged = nullptr, *sc_iters = nullptr, *ipm_iters = nullptr, *status = nullptr, *counter = nullptr;
    double *norm1_nu = nullptr, *sum_delta = nullptr, *delta_sigma = nullptr;
    // SCvx state (allocated on first scvx_setup)
    double *vx_Xold = nullptr, *vx_Uold = nullptr, *vx_tr = nullptr, *vx_last = nullptr, *vx_cost = nullptr, *vx_info = nullptr;
    int *vx_has_last = nullptr, *vx_needs_disc = nullptr, *vx_solves = nullptr;
    // SCvxAlgorithm::getAllSolutions: opt-in record of every iterate of the batch entry point (scpp_hip_scvx_record_iterates)
    bool vx_record = false;
    double *vx_iter_ring = nullptr;
    int *vx_iter_count = nullptr;
    int vx_iter_cap = 0;
    scpp_scvx_opts scvx{};
    bool scvx_ready = false;
    // streaming engine (allocated on first scvx_solve_stream): instance queue, result rows, slot -> instance map
    double *q_xinit = nullptr, *q_rows = nullptr;
    int *q_counters = nullptr; // [4] head, done, nconv
    int *q_slot_inst = nullptr;
    size_t q_cap = 0;          // capacity (instances) of q_xinit / q_rows
    int q_N = 0;               // instances of the last job
    std::vector<hipStream_t> pool_streams; // streams of the slot pools beyond the first (which runs on `stream`)
    std::vector<hipEvent_t> pool_events;
    int *h_poll = nullptr;     // pinned host words the lagged termination polls land in
    // linear MPC state (allocated on first mpc_setup)
    mpc::MpcConst *mpc_const = nullptr;
    mpc::MpcConst *mpc_host = nullptr;
    double *mpc_xf = nullptr, *mpc_x0 = nullptr, *mpc_U = nullptr, *mpc_X = nullptr, *mpc_cost = nullptr, *mpc_uheld = nullptr, *mpc_t = nullptr;
    int *mpc_status = nullptr, *mpc_iters = nullptr, *mpc_steps = nullptr, *mpc_failed = nullptr, *mpc_ipm = nullptr, *mpc_reached = nullptr;
    bool mpc_ready = false;
    int mpc_B = 0;
    // simulate scratch
    double *sim_dt = nullptr, *sim_u0 = nullptr, *sim_u1 = nullptr, *sim_x = nullptr;
    scpp_sc_opts sc{};
    ParamsOf<Plugins>::type model_params{}; // the C-ABI parameter struct of every registered model (the context's own is the one in use)
    size_t ws_per = 0;                      // doubles of interior-point workspace per instance (Lay<Table> of the context's model)
    scpp_socp_opts socp{1e-8, 1e-7, 1e-7, 60, 1};
    bool sc_ready = false, par_from_ip = false;
    bool sc_warm = false; // the last scpp_hip_sc_setup was a warm start
    int sc_persistent_min = 3072; // smallest cold batch scpp_hip_sc_solve gives to the persistent kernel (SCPP_SC_PERSISTENT_MIN: tests)
    int mode = SCPP_MODE_FOH | SCPP_MODE_VT;
    // timing
    struct Span
    {
        hipEvent_t a, b;
        int kind;
        long long inst;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    scpp_timing timing{};
    hipEvent_t ev_base = nullptr; // time origin of the span intervals (recorded at the last timing reset)
    int last_active = 0;
    long long stream_rounds = 0; // rounds enqueued by the last streaming job (diagnostics)
    int stream_pools = 0;
    int stream_engine = SCPP_STREAM_ENGINE_DEFAULT; // SCPP_STREAM_POOLS / SCPP_STREAM_PERSISTENT (scpp_hip_set_stream_engine)
    double *persist_shares = nullptr;               // [8] device: per-step wavefront ticks of the last persistent job
    unsigned long long stream_ticks[4] = {0, 0, 0, 0};
    int ipm_schedule = SCPP_IPM_SCHEDULE_DEFAULT; // SCPP_IPM_RESIDENT / SCPP_IPM_SPLIT / SCPP_IPM_RESIDENT_WS (scpp_hip_set_ipm_schedule)
    int ipm_split_pairs = 0;                      // (factor, rest) launch pairs per solve of the split schedule; 0 = the worst case 2 maxit + 1
    int disc_steps = 5; // RKF78 steps per segment: 5 = the reference's fixed count (default since round 4), 1 .. 4 pinned, 0 = discretize_kernel.h's step-length rule (opt-in)
};

namespace
{

// Every entry point runs on the context's device whatever the calling thread's current device is (a host that drives
// several GPUs from one thread, or a framework that switches devices between calls), and restores the caller's device.
struct DeviceGuard
{
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const scpp_hip_ctx *c)
    {
        if (!c)
            return;
        if (hipGetDevice(&prev) == hipSuccess && prev != c->device)
            switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched)
            (void)hipSetDevice(prev);
    }
};

template <class T>
int devAlloc(T **p, size_t n)
{
    return hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)) == hipSuccess ? 0 : -1;
}

hipEvent_t getEvent(scpp_hip_ctx *c)
{
    if (!c->pool.empty())
    {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess)
        return nullptr;
    return e;
}
void collectTiming(scpp_hip_ctx *c);
constexpr size_t MAX_OPEN_SPANS = 1 << 15; // callers that never poll the timing must not grow the event pool without bound
bool spanBegin(scpp_hip_ctx *c, int kind, long long inst, hipStream_t st)
{
    if (c->spans.size() >= MAX_OPEN_SPANS)
        collectTiming(c); // folds the finished spans into the running totals and recycles their events
    scpp_hip_ctx::Span s;
    s.a = getEvent(c);
    s.b = getEvent(c);
    if (!s.a || !s.b)
    {
        // no event to be had: this launch goes untimed
        if (s.a)
            c->pool.push_back(s.a);
        if (s.b)
            c->pool.push_back(s.b);
        return false;
    }
    s.kind = kind;
    s.inst = inst;
    (void)hipEventRecord(s.a, st);
    c->spans.push_back(s);
    return true;
}
void spanEnd(scpp_hip_ctx *c, bool open, hipStream_t st)
{
    if (open)
        (void)hipEventRecord(c->spans.back().b, st);
}

// contiguous instance range processed by one launch, and the stream it is issued on
struct Range
{
    long first;
    int count;
    hipStream_t stream;
};
Range fullRange(scpp_hip_ctx *c) { return Range{0, c->B, c->stream}; }
// length of the union of [t0, t1) intervals (ms): the time during which at least one launch of a kernel family was in flight
double unionLength(std::vector<std::pair<double, double>> &iv)
{
    std::sort(iv.begin(), iv.end());
    double total = 0., lo = 0., hi = -1.;
    for (const auto &x : iv)
    {
        if (hi < lo || x.first > hi)
        {
            if (hi >= lo)
                total += hi - lo;
            lo = x.first;
            hi = x.second;
        }
        else if (x.second > hi)
            hi = x.second;
    }
    if (hi >= lo)
        total += hi - lo;
    return total;
}
void collectTiming(scpp_hip_ctx *c)
{
    (void)hipStreamSynchronize(c->stream);
    // Launches of different slot pools run on different streams and overlap in time: the SUM of their spans (ms_socp) can
    // exceed the wall clock.  ms_*_union is the length of the union of the spans on a common time axis (origin: ev_base) --
    // the time the GPU spent with at least one launch of that family in flight, which is what a rate may be divided by.
    std::vector<std::pair<double, double>> iv[2];
    for (auto &s : c->spans)
    {
        float ms = 0.f, t0 = 0.f;
        (void)hipEventSynchronize(s.b);
        (void)hipEventElapsedTime(&ms, s.a, s.b);
        const bool based = c->ev_base && hipEventElapsedTime(&t0, c->ev_base, s.a) == hipSuccess;
        if (s.kind == 0)
        {
            c->timing.ms_discretize += ms;
            c->timing.n_discretize++;
            c->timing.inst_discretize += s.inst;
        }
        else if (s.kind == 1)
        {
            c->timing.ms_socp += ms;
            c->timing.n_socp++;
            c->timing.inst_socp += s.inst;
        }
        else
            c->timing.ms_other += ms;
        if (s.kind == 0 || s.kind == 1)
        {
            if (based)
                iv[s.kind].emplace_back(double(t0), double(t0) + double(ms));
            else
                (s.kind == 0 ? c->timing.ms_discretize_union : c->timing.ms_socp_union) += ms; // no common axis: spans as they are
        }
        c->pool.push_back(s.a);
        c->pool.push_back(s.b);
    }
    c->timing.ms_discretize_union += unionLength(iv[0]);
    c->timing.ms_socp_union += unionLength(iv[1]);
    c->spans.clear();
}

template <class Model>
int launchDiscretize(scpp_hip_ctx *c, int mode, const double *par0, int stride, const int *active0, long long ninst, Range r)
{
    const int B = r.count, K = c->K;
    const long groups = ((long(B) + 7) / 8) * (K - 1);
    const unsigned grid = unsigned(groups * 8);
    const size_t f = size_t(r.first), nx = size_t(c->nx), nu = size_t(c->nu), seg = size_t(K - 1);
    const double *par = par0 + f * size_t(stride);
    const int *active = active0 ? active0 + f : nullptr;
    // instance-major buffers: a range is a pointer offset
    struct
    {
        double *X, *U, *sigma, *A, *Bm, *C, *S, *Z;
        hipStream_t stream;
    } v{c->X + f * K * nx, c->U + f * K * nu, c->sigma + f, c->A + f * seg * nx * nx, c->Bm + f * seg * nx * nu,
        c->C + f * seg * nx * nu, c->S + f * seg * nx, c->Z + f * seg * nx, r.stream};
    const bool timed = spanBegin(c, 0, ninst, r.stream);
    if (mode == (SCPP_MODE_FOH | SCPP_MODE_VT))
        hipLaunchKernelGGL((discretize_kernel<Model, true, true>), dim3(grid), dim3(WAVE), 0, v.stream, B, K, v.X, v.U,
                           v.sigma, par, stride, active, v.A, v.Bm, v.C, v.S, v.Z, c->disc_steps);
    else if (mode == SCPP_MODE_FOH)
        hipLaunchKernelGGL((discretize_kernel<Model, true, false>), dim3(grid), dim3(WAVE), 0, v.stream, B, K, v.X, v.U,
                           v.sigma, par, stride, active, v.A, v.Bm, v.C, v.S, v.Z, c->disc_steps);
    else if (mode == SCPP_MODE_VT)
        hipLaunchKernelGGL((discretize_kernel<Model, false, true>), dim3(grid), dim3(WAVE), 0, v.stream, B, K, v.X, v.U,
                           v.sigma, par, stride, active, v.A, v.Bm, v.C, v.S, v.Z, c->disc_steps);
    else
        hipLaunchKernelGGL((discretize_kernel<Model, false, false>), dim3(grid), dim3(WAVE), 0, v.stream, B, K, v.X, v.U,
                           v.sigma, par, stride, active, v.A, v.Bm, v.C, v.S, v.Z, c->disc_steps);
    spanEnd(c, timed, r.stream);
    return hipGetLastError() == hipSuccess ? 0 : SCPP_E_HIP;
}

int discretizeDispatch(scpp_hip_ctx *c, int mode, const double *par, int stride, const int *active, long long ninst, Range r)
{
    return withPlugin(c->model, [&](auto pl) { return launchDiscretize<typename decltype(pl)::Model>(c, mode, par, stride, active, ninst, r); });
}
int discretizeDispatch(scpp_hip_ctx *c, int mode, const double *par, int stride, const int *active, long long ninst)
{
    return discretizeDispatch(c, mode, par, stride, active, ninst, fullRange(c));
}

SCBuffers scBuffers(scpp_hip_ctx *c)
{
    SCBuffers b;
    b.B = c->B;
    b.K = c->K;
    b.x_init_dim = c->x_init;
    b.X = c->X;
    b.U = c->U;
    b.sigma = c->sigma;
    b.ip = c->ip;
    b.uhat = c->uhat;
    b.wtrx = c->wtrx;
    b.active = c->active;
    b.converged = c->converged;
    b.sc_iters = c->sc_iters;
    b.ipm_iters = c->ipm_iters;
    b.status = c->status;
    b.norm1_nu = c->norm1_nu;
    b.sum_delta = c->sum_delta;
    b.delta_sigma = c->delta_sigma;
    return b;
}

// instance-major per-instance arrays: a range is a pointer offset
SCBuffers scBuffersRange(scpp_hip_ctx *c, Range r)
{
    SCBuffers b = scBuffers(c);
    const size_t f = size_t(r.first), K = size_t(c->K);
    b.B = r.count;
    b.x_init_dim += f * size_t(c->nx);
    b.X += f * K * size_t(c->nx);
    b.U += f * K * size_t(c->nu);
    b.sigma += f;
    b.ip += f * ipm::IP_N;
    b.uhat += f * K * 3;
    b.wtrx += f;
    b.active += f;
    b.converged += f;
    b.sc_iters += f;
    b.ipm_iters += f;
    b.status += f;
    b.norm1_nu += f;
    b.sum_delta += f;
    b.delta_sigma += f;
    return b;
}

// the parameter struct of plugin PL inside a context
template <class PL>
typename PL::Params &paramsOf(scpp_hip_ctx *c)
{
    return std::get<typename PL::Params>(c->model_params);
}
ipm::KernelArgs ipmArgs(scpp_hip_ctx *c, int do_sc_update, bool masked, Range r, bool snapshot)
{
    ipm::KernelArgs a;
    const size_t f = size_t(r.first), K = size_t(c->K), seg = K - 1, nx = size_t(c->nx), nu = size_t(c->nu);
    a.B = r.count;
    a.K = c->K;
    a.X = c->X + f * K * nx;
    a.U = c->U + f * K * nu;
    a.sigma = c->sigma + f;
    a.A = c->A + f * seg * nx * nx;
    a.Bm = c->Bm + f * seg * nx * nu;
    a.C = c->C + f * seg * nx * nu;
    a.S = c->S + f * seg * nx;
    a.Z = c->Z + f * seg * nx;
    a.ip = c->ip + f * ipm::IP_N;
    a.uhat = c->uhat + f * K * 3;
    a.ws = c->ws + f * c->ws_per;
    a.wtrx = c->wtrx + f;
    a.active = (do_sc_update || masked) ? c->active + f : nullptr;
    a.converged = c->converged + f;
    a.sc_iters = c->sc_iters + f;
    a.ipm_iters = c->ipm_iters + f;
    a.status = c->status + f;
    a.norm1_nu = c->norm1_nu + f;
    a.sum_delta = c->sum_delta + f;
    a.delta_sigma = c->delta_sigma + f;
    a.nu_tol = c->sc.nu_tol;
    a.delta_tol = c->sc.delta_tol;
    a.max_sc_iterations = c->sc.max_iterations;
    a.warm = c->ipm_warm + f;
    a.do_sc_update = do_sc_update;
    a.dd_fresh = (snapshot && c->ipm_schedule == SCPP_IPM_RESIDENT) ? c->vx_needs_disc + f : nullptr; // SCvx rounds: needs_disc == 0 <=> a re-solve on the old dd
    a.Xold = snapshot ? c->vx_Xold + f * K * nx : nullptr;
    a.Uold = snapshot ? c->vx_Uold + f * K * nu : nullptr;
    a.opt.feastol = c->socp.feastol;
    a.opt.abstol = c->socp.abstol;
    a.opt.reltol = c->socp.reltol;
    a.opt.gamma = 0.99;
    a.opt.maxit = c->socp.maxit;
    a.opt.use_mfma = c->socp.use_mfma;
    a.dbg = c->dbg + f * 32;
    return a;
}
int launchIpm(scpp_hip_ctx *c, int do_sc_update, long long ninst, bool masked, Range r, unsigned lds_pad = 0, bool snapshot = false)
{
    const ipm::KernelArgs a = ipmArgs(c, do_sc_update, masked, r, snapshot);
    const bool timed = spanBegin(c, 1, ninst, r.stream);
    // lds_pad: dynamic LDS that is never touched -- it only limits how many ipm workgroups fit on a CU (pipelined loop)
    // one instantiation of the solver per model table (csrc/constraint_table.h) and input hold
    const bool zoh = !(c->mode & SCPP_MODE_FOH); // zero-order hold: the table's ZeroOrderHold variant (same record layout)
    (void)withPlugin(c->model, [&](auto pl) {
        using PL = decltype(pl);
        using T = typename PL::Table;
        const unsigned grid = unsigned(r.count);
        if constexpr (PL::SPLIT_SCHEDULE)
        {
            if (!zoh && c->ipm_schedule == SCPP_IPM_SPLIT)
            {
                // split schedule (ipm_split.h): A(first) [F A] x pairs.  An instance asks for at most maxit factor sweeps per attempt plus one for the cold
                // initialisation, and a warm attempt that breaks down is repeated cold: 2 maxit + 1 pairs cover every instance; one that has finished
                // returns at the top of each later launch.
                using W = ipm::SegFieldsInWorkspace<T>;
                const int pairs = c->ipm_split_pairs > 0 ? c->ipm_split_pairs : 2 * a.opt.maxit + 1;
                hipLaunchKernelGGL(ipm::ipm_split_kernel<W>, dim3(grid), dim3(WAVE), 0, r.stream, a, 1);
                for (int i = 0; i < pairs; i++)
                {
                    hipLaunchKernelGGL(ipm::ipm_factor_kernel<T>, dim3(grid), dim3(WAVE), 0, r.stream, a);
                    hipLaunchKernelGGL(ipm::ipm_split_kernel<W>, dim3(grid), dim3(WAVE), 0, r.stream, a, 0);
                }
                // a truncated schedule (measurement hook) may leave instances waiting for a factor sweep: retire them as failures instead of reporting stale rows
                if (pairs < 2 * a.opt.maxit + 1)
                    hipLaunchKernelGGL(ipm::ipm_split_finalize_kernel<T>, dim3(unsigned((r.count + WAVE - 1) / WAVE)), dim3(WAVE), 0, r.stream, a);
                return 0;
            }
#ifdef SCPP_HIP_EMU // (diagnostic of the layout policy; the device library does not carry a third instantiation of the solver for it)
            if (!zoh && c->ipm_schedule == SCPP_IPM_RESIDENT_WS)
            {
                hipLaunchKernelGGL(ipm::ipm_kernel<ipm::SegFieldsInWorkspace<T>>, dim3(grid), dim3(WAVE), lds_pad, r.stream, a);
                return 0;
            }
#endif
        }
        const unsigned lds = lds_pad + unsigned(ipm::segLdsBytes<T>(c->K));
        if (!zoh)
            hipLaunchKernelGGL(ipm::ipm_kernel<T>, dim3(grid), dim3(WAVE), lds, r.stream, a);
        else
            hipLaunchKernelGGL(ipm::ipm_kernel<ipm::ZeroOrderHold<T>>, dim3(grid), dim3(WAVE), lds, r.stream, a);
        return 0;
    });
    spanEnd(c, timed, r.stream);
    return hipGetLastError() == hipSuccess ? 0 : SCPP_E_HIP;
}
int launchIpm(scpp_hip_ctx *c, int do_sc_update, long long ninst, bool masked = false)
{
    return launchIpm(c, do_sc_update, ninst, masked, fullRange(c));
}

// the interior-point workspace (0.77 MB per RocketQuat instance at K = 50) is allocated by the first SC / SCvx set-up, so that
// contexts used for discretisation or linear MPC only do not carry it
int ensureWorkspace(scpp_hip_ctx *c)
{
    if (c->ws)
        return 0;
    return devAlloc(&c->ws, size_t(c->Bmax) * c->ws_per) ? SCPP_E_HIP : 0;
}

// ---- create-time self-test of the tile engine on the device the context is created on ----
// One wavefront: a known SPD 16 x 16 tile (diagonally dominant, entries from a fixed recurrence) goes through the matrix-core
// product, the inverse-Cholesky elimination (LDS exchange, DPP row broadcasts, v_readlane, hardware reciprocal + Newton) and
// the LDS transpose -- everything the sweeps of ipm_kernel are built from -- and || Li A Li' - I ||_max comes back.  It is a
// check of the toolchain + device combination the library is running on (DESIGN.md 4.2 "Toolchain hazard"), run once per
// process and device; it does not replace the GPU parity suite (the two miscompilations seen so far sat in argument handling of
// the big kernel, not in the tile engine).
__global__ void __launch_bounds__(WAVE) tile_selftest_kernel(double *out)
{
    using namespace ipm;
    __shared__ TileShared sh;
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    Tile A;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        const double off = 0.5 / double(1 + (row > i ? row - i : i - row)) * (((row + i) & 1) ? -1. : 1.);
        A.v[r] = row == i ? 4. + 0.25 * row : off; // symmetric, strictly diagonally dominant
    }
    const Tile Li = invCholFactor<16>(A, sh, lane);
    const Tile Lit = transposeTile(Li, sh, lane);
    // Li A Li' = mm(Lit, mm(A, Lit)) with mm(X, Y) = X'Y and A symmetric
    const Tile R = mm(Lit, mm(A, Lit));
    double err = 0.;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const double e = fabs(R.v[r] - ((g + 4 * r == i) ? 1. : 0.));
        err = e > err ? e : err;
    }
    err = wave_max(err);
    if (lane == 0)
        out[0] = err;
}
int tileSelfTest(int device)
{
    static std::mutex mtx; // contexts of different devices are created from different host threads (host/sc_oneshot --gpus N)
    static std::vector<int> done; // devices that passed in this process
    std::lock_guard<std::mutex> lock(mtx);
    for (int d : done)
        if (d == device)
            return SCPP_OK;
    double *d_out = nullptr, h = -1.;
    if (hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(double)) != hipSuccess)
        return SCPP_E_HIP;
    hipLaunchKernelGGL(tile_selftest_kernel, dim3(1), dim3(WAVE), 0, nullptr, d_out);
    const bool ok = hipMemcpy(&h, d_out, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d_out);
    if (!ok || !(h >= 0. && h < 1e-12))
    {
        std::fprintf(stderr, "scpp_hip: tile-engine self-test FAILED on device %d (|Li A Li' - I|_max = %g): this build / device must not be used (%s)\n",
                     device, h, scpp_hip_version());
        return SCPP_E_HIP;
    }
    done.push_back(device);
    return SCPP_OK;
}

int countActive(scpp_hip_ctx *c, int *n)
{
    hipLaunchKernelGGL(count_active_kernel, dim3(1), dim3(256), 0, c->stream, c->B, (const int *)c->active, c->counter);
    CHECK_HIP(hipMemcpyAsync(n, c->counter, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

} // namespace

extern "C"
{

// The toolchain this source was validated with on hardware (every kernel instantiation has a whole-run parity test under
// `pytest -m gpu`, DESIGN.md 4.2 "Toolchain hazard").  Two miscompilations of ipm_kernel were seen with it -- both in the
// SGPR-spill path under heavy VGPR pressure, both worked around in source -- so a build with ANY other compiler, or at another
// occupancy, is unvalidated until the GPU tests have run against it: the version string says so and scpp_hip_create warns once.
#define SCPP_VALIDATED_CLANG_MAJOR 22
#define SCPP_VALIDATED_HIP_MAJOR 7
#define SCPP_VALIDATED_HIP_MINOR 2
#define SCPP_STR2(x) #x
#define SCPP_STR(x) SCPP_STR2(x)
#ifndef SCPP_HIP_EMU
static_assert(IPM_WAVES_PER_SIMD <= 2 && DISC_WAVES_PER_SIMD <= 2,
              "three waves per SIMD (168 VGPRs) gave wrong results / memory faults on hardware with this toolchain (SGPR spills to VGPR "
              "lanes, DESIGN.md 4.2) and were slower: the occupancy-3 variants are not built");
#if defined(__clang_major__) && __clang_major__ == SCPP_VALIDATED_CLANG_MAJOR && HIP_VERSION_MAJOR == SCPP_VALIDATED_HIP_MAJOR &&         \
    HIP_VERSION_MINOR == SCPP_VALIDATED_HIP_MINOR
#define SCPP_TOOLCHAIN_VALIDATED 1
#define SCPP_TOOLCHAIN_NOTE "validated toolchain"
#else
#define SCPP_TOOLCHAIN_VALIDATED 0
#define SCPP_TOOLCHAIN_NOTE "UNVALIDATED toolchain: run pytest -m gpu before trusting results"
#endif
#endif

const char *scpp_hip_version(void)
{
#ifdef SCPP_HIP_EMU
    return "scpp_hip 0.4 (CPU emulation build: TEST ONLY)";
#else
    return "scpp_hip 0.4 (gfx950; clang " __clang_version__ "; HIP " SCPP_STR(HIP_VERSION_MAJOR) "." SCPP_STR(HIP_VERSION_MINOR) "." SCPP_STR(
        HIP_VERSION_PATCH) "; ipm_kernel " SCPP_STR(IPM_WAVES_PER_SIMD) " waves/SIMD, discretize_kernel " SCPP_STR(DISC_WAVES_PER_SIMD) " waves/SIMD; " SCPP_TOOLCHAIN_NOTE ")";
#endif
}

int scpp_hip_query(int what, long long *value)
{
    if (!value)
        return SCPP_E_ARG;
    switch (what)
    {
    case SCPP_Q_ABI_REVISION: *value = SCPP_ABI_REVISION; return SCPP_OK;
    case SCPP_Q_STATUS_REJECTION_CAP: *value = SCPP_STATUS_REJECTION_CAP; return SCPP_OK;
    case SCPP_Q_SCVX_SOLVE_CAP: *value = SCVX_SOLVE_CAP; return SCPP_OK;
    case SCPP_Q_MAX_K: *value = WAVE; return SCPP_OK;
    case SCPP_Q_MPC_MAX_K: *value = 8; return SCPP_OK;
    }
    return SCPP_E_ARG;
}

int scpp_hip_create(scpp_hip_ctx **out, int device_id, int model_id, int K, int batch_max, unsigned)
{
    if (!out || K < 3 || K > WAVE || batch_max < 1)
        return SCPP_E_ARG;
    if (withPlugin(model_id, [](auto) { return 0; }) != 0) // a model nobody registered (csrc/sc_kernels.h: Plugins)
        return SCPP_E_ARG;
#if !defined(SCPP_HIP_EMU) && !SCPP_TOOLCHAIN_VALIDATED
    {
        static std::once_flag warned; // contexts are created from several host threads (`--gpus N`)
        std::call_once(warned, [] {
            std::fprintf(stderr, "scpp_hip: built with an unvalidated toolchain (%s); run `pytest -m gpu` before trusting results\n", scpp_hip_version());
        });
    }
#endif
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    CHECK_HIP(hipSetDevice(device_id));
    struct Restore
    {
        int d;
        ~Restore()
        {
            if (d >= 0)
                (void)hipSetDevice(d);
        }
    } restore{prev_device == device_id ? -1 : prev_device};
    if (int rc = tileSelfTest(device_id))
        return rc;
    scpp_hip_ctx *c = new (std::nothrow) scpp_hip_ctx;
    if (!c)
        return SCPP_E_HIP;
    c->device = device_id;
    c->model = model_id;
    c->K = K;
    c->Bmax = batch_max;
    c->B = 0;
    // measurement hooks (A/B runs of one library): the schedule of a new context and its launch pairs per solve; scpp_hip_set_ipm_schedule overrides
    if (const char *e = std::getenv("SCPP_IPM_SCHEDULE"))
        if (std::atoi(e) >= SCPP_IPM_RESIDENT && std::atoi(e) <= SCPP_IPM_SPLIT)
            c->ipm_schedule = std::atoi(e);
    if (const char *e = std::getenv("SCPP_STREAM_ENGINE"))
        if (std::atoi(e) == SCPP_STREAM_POOLS || std::atoi(e) == SCPP_STREAM_PERSISTENT)
            c->stream_engine = std::atoi(e);
    if (const char *e = std::getenv("SCPP_SC_PERSISTENT_MIN"))
        if (std::atoi(e) > 0)
            c->sc_persistent_min = std::atoi(e);
    if (const char *e = std::getenv("SCPP_IPM_SPLIT_PAIRS"))
        if (std::atoi(e) > 0)
            c->ipm_split_pairs = std::atoi(e);
    (void)withPlugin(model_id, [&](auto pl) {
        using PL = decltype(pl);
        c->nx = PL::Model::NX;
        c->nu = PL::Model::NU;
        c->np = PL::Model::NP;
        c->ws_per = ipm::workspaceDoubles<typename PL::Table>(K);
        return 0;
    });
    if (hipStreamCreate(&c->stream) != hipSuccess)
    {
        delete c;
        return SCPP_E_HIP;
    }
    const size_t B = size_t(batch_max), nx = c->nx, nu = c->nu;
    int rc = 0;
    rc |= devAlloc(&c->X, B * K * nx);
    rc |= devAlloc(&c->U, B * K * nu);
    rc |= devAlloc(&c->sigma, B);
    rc |= devAlloc(&c->par, B * c->np);
    rc |= devAlloc(&c->A, B * (K - 1) * nx * nx);
    rc |= devAlloc(&c->Bm, B * (K - 1) * nx * nu);
    rc |= devAlloc(&c->C, B * (K - 1) * nx * nu);
    rc |= devAlloc(&c->S, B * (K - 1) * nx);
    rc |= devAlloc(&c->Z, B * (K - 1) * nx);
    rc |= devAlloc(&c->sim_dt, B);
    rc |= devAlloc(&c->sim_u0, B * nu);
    rc |= devAlloc(&c->sim_u1, B * nu);
    rc |= devAlloc(&c->sim_x, B * nx);
    rc |= devAlloc(&c->active, B);
    rc |= devAlloc(&c->ipm_warm, B);
    rc |= devAlloc(&c->converged, B);
    rc |= devAlloc(&c->sc_iters, B);
    rc |= devAlloc(&c->ipm_iters, B);
    rc |= devAlloc(&c->status, B);
    rc |= devAlloc(&c->counter, 4);
    rc |= devAlloc(&c->norm1_nu, B);
    rc |= devAlloc(&c->sum_delta, B);
    rc |= devAlloc(&c->delta_sigma, B);
    {
        rc |= devAlloc(&c->x_init, B * nx);
        rc |= devAlloc(&c->ip, B * ipm::IP_N);
        rc |= devAlloc(&c->uhat, B * K * 3);
        rc |= devAlloc(&c->wtrx, B);
        rc |= devAlloc(&c->dbg, B * 32);
    }
    if (rc)
    {
        scpp_hip_destroy(c);
        return SCPP_E_HIP;
    }
    (void)hipMemset(c->active, 0, B * sizeof(int));
    (void)hipMemset(c->ipm_warm, 0, B * sizeof(int));
    *out = c;
    return SCPP_OK;
}

int scpp_hip_destroy(scpp_hip_ctx *c)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_OK;
    (void)hipStreamSynchronize(c->stream);
    void *ptrs[] = {c->X, c->U, c->sigma, c->par, c->A, c->Bm, c->C, c->S, c->Z, c->x_init, c->ip, c->uhat, c->wtrx, c->ws,
                    c->dbg, c->active, c->converged, c->sc_iters, c->ipm_iters, c->status, c->counter, c->norm1_nu,
                    c->sum_delta, c->delta_sigma, c->sim_dt, c->sim_u0, c->sim_u1, c->sim_x, c->vx_Xold, c->vx_Uold, c->vx_tr,
                    c->vx_last, c->vx_cost, c->vx_info, c->vx_has_last, c->vx_needs_disc, c->vx_solves, c->ipm_warm,
                    c->mpc_const, c->mpc_xf, c->mpc_x0, c->mpc_U, c->mpc_X, c->mpc_cost, c->mpc_uheld, c->mpc_t, c->mpc_status,
                    c->mpc_iters, c->mpc_steps, c->mpc_failed, c->mpc_ipm, c->mpc_reached};
    delete c->mpc_host;
    c->mpc_host = nullptr;
    for (void *p : {(void *)c->q_xinit, (void *)c->q_rows, (void *)c->q_counters, (void *)c->q_slot_inst, (void *)c->persist_shares, (void *)c->vx_iter_ring,
                    (void *)c->vx_iter_count})
        if (p)
            (void)hipFree(p);
    if (c->h_poll)
        (void)hipHostFree(c->h_poll);
    for (auto st : c->pool_streams)
        (void)hipStreamDestroy(st);
    for (auto e : c->pool_events)
        (void)hipEventDestroy(e);
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    for (auto &s : c->spans)
    {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (auto e : c->pool)
        (void)hipEventDestroy(e);
    if (c->ev_base)
        (void)hipEventDestroy(c->ev_base);
    if (c->ev_skew)
        (void)hipEventDestroy(c->ev_skew);
    if (c->ev_join)
        (void)hipEventDestroy(c->ev_join);
    if (c->stream2)
        (void)hipStreamDestroy(c->stream2);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return SCPP_OK;
}

int scpp_hip_set_flow_params(scpp_hip_ctx *c, const double *par, int B)
{
    DeviceGuard guard(c);
    if (!c || !par || B < 1 || B > c->Bmax)
        return SCPP_E_ARG;
    CHECK_HIP(hipMemcpyAsync(c->par, par, size_t(B) * c->np * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->par_from_ip = false;
    return SCPP_OK;
}

int scpp_hip_upload_traj(scpp_hip_ctx *c, const double *X, const double *U, const double *sigma, int B)
{
    DeviceGuard guard(c);
    if (!c || !X || !U || !sigma || B < 1 || B > c->Bmax)
        return SCPP_E_ARG;
    c->B = B;
    CHECK_HIP(hipMemcpyAsync(c->X, X, size_t(B) * c->K * c->nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->U, U, size_t(B) * c->K * c->nu * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->sigma, sigma, size_t(B) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return SCPP_OK;
}

int scpp_hip_upload_traj_zoh(scpp_hip_ctx *c, const double *X, const double *U, const double *sigma, int B)
{
    DeviceGuard guard(c);
    if (!c || !X || !U || !sigma || B < 1 || B > c->Bmax)
        return SCPP_E_ARG;
    c->B = B;
    const size_t K = size_t(c->K), nu = size_t(c->nu);
    CHECK_HIP(hipMemcpyAsync(c->X, X, size_t(B) * K * c->nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // td.U holds K-1 inputs under zero-order hold (trajectoryData.hpp:27-32); the device keeps the [B][K][nu] pitch
    CHECK_HIP(hipMemsetAsync(c->U, 0, size_t(B) * K * nu * sizeof(double), c->stream));
    CHECK_HIP(hipMemcpy2DAsync(c->U, K * nu * sizeof(double), U, (K - 1) * nu * sizeof(double), (K - 1) * nu * sizeof(double), size_t(B),
                               hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->sigma, sigma, size_t(B) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return SCPP_OK;
}

int scpp_hip_discretize(scpp_hip_ctx *c, int mode)
{
    DeviceGuard guard(c);
    if (!c || c->B < 1 || mode < 0 || mode > 3)
        return SCPP_E_ARG;
    if (!(mode & SCPP_MODE_FOH))
    {
        /* zero-order hold: K-1 inputs per trajectory (scpp_hip_upload_traj_zoh); dd.C is empty in the reference
           (discretizationData.hpp:56-59) -- here it reads back as zeros */
        CHECK_HIP(hipMemsetAsync(c->C, 0, size_t(c->B) * (c->K - 1) * c->nx * c->nu * sizeof(double), c->stream));
    }
    const double *par = c->par_from_ip ? c->ip + ipm::IP_PAR : c->par;
    const int stride = c->par_from_ip ? ipm::IP_N : c->np;
    int rc = discretizeDispatch(c, mode, par, stride, nullptr, c->B);
    if (rc)
        return rc;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    return SCPP_OK;
}

int scpp_hip_download_dd(scpp_hip_ctx *c, double *A, double *B, double *C, double *S, double *Z)
{
    DeviceGuard guard(c);
    if (!c || c->B < 1)
        return SCPP_E_ARG;
    const size_t n = size_t(c->B) * (c->K - 1), nx = c->nx, nu = c->nu;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    if (A)
        CHECK_HIP(hipMemcpy(A, c->A, n * nx * nx * sizeof(double), hipMemcpyDeviceToHost));
    if (B)
        CHECK_HIP(hipMemcpy(B, c->Bm, n * nx * nu * sizeof(double), hipMemcpyDeviceToHost));
    if (C)
        CHECK_HIP(hipMemcpy(C, c->C, n * nx * nu * sizeof(double), hipMemcpyDeviceToHost));
    if (S)
        CHECK_HIP(hipMemcpy(S, c->S, n * nx * sizeof(double), hipMemcpyDeviceToHost));
    if (Z)
        CHECK_HIP(hipMemcpy(Z, c->Z, n * nx * sizeof(double), hipMemcpyDeviceToHost));
    return SCPP_OK;
}

int scpp_hip_simulate(scpp_hip_ctx *c, const double *dt, const double *u0, const double *u1, double *x, int B)
{
    DeviceGuard guard(c);
    if (!c || !dt || !u0 || !u1 || !x || B < 1 || B > c->Bmax)
        return SCPP_E_ARG;
    CHECK_HIP(hipMemcpyAsync(c->sim_dt, dt, size_t(B) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->sim_u0, u0, size_t(B) * c->nu * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->sim_u1, u1, size_t(B) * c->nu * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->sim_x, x, size_t(B) * c->nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const double *par = c->par_from_ip ? c->ip + ipm::IP_PAR : c->par;
    const int stride = c->par_from_ip ? ipm::IP_N : c->np;
    const unsigned grid = unsigned((B + 63) / 64);
    (void)withPlugin(c->model, [&](auto pl) {
        hipLaunchKernelGGL((simulate_kernel<typename decltype(pl)::Model>), dim3(grid), dim3(64), 0, c->stream, B, par, stride,
                           (const double *)c->sim_dt, (const double *)c->sim_u0, (const double *)c->sim_u1, c->sim_x, (const int *)nullptr);
        return 0;
    });
    CHECK_HIP(hipMemcpyAsync(x, c->sim_x, size_t(B) * c->nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CHECK_HIP(hipStreamSynchronize(c->stream));
    return SCPP_OK;
}

int scpp_hip_set_discretization_steps(scpp_hip_ctx *c, int steps)
{
    if (!c || steps < 0 || steps > DISC_STEPS_MAX) // 5 = the reference (default), 0 = the step-length rule (opt-in), 1 .. 4 pinned
        return SCPP_E_ARG;
    c->disc_steps = steps;
    return SCPP_OK;
}

int scpp_hip_set_stream_engine(scpp_hip_ctx *c, int engine)
{
    if (!c || (engine != SCPP_STREAM_POOLS && engine != SCPP_STREAM_PERSISTENT))
        return SCPP_E_ARG;
    c->stream_engine = engine;
    return SCPP_OK;
}

int scpp_hip_stream_profile(scpp_hip_ctx *c, double *ticks)
{
    if (!c || !ticks)
        return SCPP_E_ARG;
    for (int i = 0; i < 4; i++)
        ticks[i] = double(c->stream_ticks[i]);
    return SCPP_OK;
}

int scpp_hip_set_ipm_schedule(scpp_hip_ctx *c, int schedule, int split_pairs)
{
    if (!c || schedule < SCPP_IPM_RESIDENT || schedule > SCPP_IPM_RESIDENT_WS || split_pairs < 0)
        return SCPP_E_ARG;
#ifndef SCPP_HIP_EMU
    if (schedule == SCPP_IPM_RESIDENT_WS)
        return SCPP_E_UNSUPPORTED; // emulation build only
#endif
    c->ipm_schedule = schedule;
    c->ipm_split_pairs = split_pairs;
    return SCPP_OK;
}

int scpp_hip_set_socp_opts(scpp_hip_ctx *c, const scpp_socp_opts *o)
{
    DeviceGuard guard(c);
    if (!c || !o)
        return SCPP_E_ARG;
    c->socp = *o;
    return SCPP_OK;
}

extern "C++"
{
namespace
{
// scpp_hip_sc_setup of every model (PL = its plugin): SCAlgorithm::initialize + the start of solve(warm_start) for B instances
template <class PL>
int scSetupT(scpp_hip_ctx *c, const typename PL::Params *mp, const scpp_sc_opts *so, const double *x_init, int B, int warm_start)
{
    DeviceGuard guard(c);
    if (!c || !mp || !so || !x_init || B < 1 || B > c->Bmax)
        return SCPP_E_ARG;
    if (c->model != PL::ID)
        return SCPP_E_UNSUPPORTED;
    /* first-order or zero-order hold, free or fixed final time (SCProblem.cpp:33-59,78-100,116-120); what the plugin refuses (RocketQuat: roll
       control on -- SC_oneshot / SC_sim run with it off) */
    if (so->K != c->K || !PL::supported(*mp))
        return SCPP_E_UNSUPPORTED;
    if (warm_start && (!c->sc_ready || B != c->B))
        return SCPP_E_STATE;
    if (int rc = ensureWorkspace(c))
        return rc;
    c->B = B;
    paramsOf<PL>(c) = *mp;
    c->sc = *so;
    c->mode = (so->interpolate_input ? SCPP_MODE_FOH : 0) | (so->free_final_time ? SCPP_MODE_VT : 0);
    if (!so->free_final_time) // fixed final time (SCProblem.cpp:33-35): dS/dsigma = 0, the discretisation does not write it
        CHECK_HIP(hipMemsetAsync(c->S, 0, size_t(B) * (c->K - 1) * size_t(c->nx) * sizeof(double), c->stream));
    if (!so->interpolate_input) // zero-order hold: no C in the dynamics (discretizationData.hpp:56-59)
        CHECK_HIP(hipMemsetAsync(c->C, 0, size_t(B) * (c->K - 1) * size_t(c->nx) * size_t(c->nu) * sizeof(double), c->stream));
    CHECK_HIP(hipMemcpyAsync(c->x_init, x_init, size_t(B) * size_t(PL::NX) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (!warm_start || c->scvx_ready) // cold SC solve (or a context last used in SCvx mode): cold interior-point start
        CHECK_HIP(hipMemsetAsync(c->ipm_warm, 0, size_t(c->Bmax) * sizeof(int), c->stream));
    c->scvx_ready = false;
    SCBuffers b = scBuffers(c);
    hipLaunchKernelGGL((sc_setup_kernel<PL>), dim3(unsigned((B + 63) / 64)), dim3(64), 0, c->stream, b, paramsOf<PL>(c), c->sc, warm_start);
    c->sc_ready = true;
    c->sc_warm = warm_start != 0;
    c->par_from_ip = true;
    c->last_active = B;
    return hipGetLastError() == hipSuccess ? SCPP_OK : SCPP_E_HIP;
}
// redimensionalizeTrajectory of the batch, on the context's stream
void launchRedim(scpp_hip_ctx *c)
{
    (void)withPlugin(c->model, [&](auto pl) {
        hipLaunchKernelGGL((sc_redim_kernel<decltype(pl)>), dim3(unsigned((c->B + 63) / 64)), dim3(64), 0, c->stream, scBuffers(c));
        return 0;
    });
}
} // namespace
} // extern "C++"

int scpp_hip_sc_setup(scpp_hip_ctx *c, const scpp_rocketquat_params *mp, const scpp_sc_opts *so, const double *x_init,
                      int B, int warm_start)
{
    return scSetupT<RocketQuatPlugin>(c, mp, so, x_init, B, warm_start);
}

int scpp_hip_sc_setup_rocket2d(scpp_hip_ctx *c, const scpp_rocket2d_params *mp, const scpp_sc_opts *so, const double *x_init,
                               int B, int warm_start)
{
    return scSetupT<Rocket2dPlugin>(c, mp, so, x_init, B, warm_start);
}

int scpp_hip_sc_setup_lander3dof(scpp_hip_ctx *c, const scpp_lander3dof_params *mp, const scpp_sc_opts *so, const double *x_init,
                                 int B, int warm_start)
{
    return scSetupT<Lander3dofPlugin>(c, mp, so, x_init, B, warm_start);
}

int scpp_hip_sc_set_active(scpp_hip_ctx *c, const int32_t *mask, int B)
{
    DeviceGuard guard(c);
    if (!c || !mask || B != c->B)
        return SCPP_E_ARG;
    if (!c->sc_ready)
        return SCPP_E_STATE;
    std::vector<int> m(size_t(B), 0);
    int n = 0;
    for (int i = 0; i < B; i++)
    {
        m[size_t(i)] = mask[i] != 0;
        n += m[size_t(i)];
    }
    CHECK_HIP(hipMemcpyAsync(c->active, m.data(), size_t(B) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipStreamSynchronize(c->stream));
    c->last_active = n;
    return SCPP_OK;
}

int scpp_hip_sc_iterate(scpp_hip_ctx *c, int *n_active)
{
    DeviceGuard guard(c);
    if (!c || !c->sc_ready)
        return SCPP_E_STATE;
    int rc = discretizeDispatch(c, c->mode, c->ip + ipm::IP_PAR, ipm::IP_N, c->active, c->last_active);
    if (rc)
        return rc;
    rc = launchIpm(c, 1, c->last_active);
    if (rc)
        return rc;
    int n = 0;
    rc = countActive(c, &n);
    if (rc)
        return rc;
    c->last_active = n;
    if (n_active)
        *n_active = n;
    return SCPP_OK;
}

int scpp_hip_sc_finish(scpp_hip_ctx *c, int *n_converged)
{
    DeviceGuard guard(c);
    if (!c || !c->sc_ready)
        return SCPP_E_STATE;
    if (c->sc.nondimensionalize)
        launchRedim(c);
    CHECK_HIP(hipStreamSynchronize(c->stream));
    if (n_converged)
    {
        std::vector<int> conv(c->B);
        CHECK_HIP(hipMemcpy(conv.data(), c->converged, size_t(c->B) * sizeof(int), hipMemcpyDeviceToHost));
        int n = 0;
        for (int v : conv)
            n += v;
        *n_converged = n;
    }
    return SCPP_OK;
}

extern "C++"
{
namespace
{
// the whole SCAlgorithm::solve loop of every instance in ONE launch (scvx_persistent.h: sc_persistent_kernel) for the configuration the kernel is
// instantiated for: the plugins that ask for it (SC_PERSISTENT: RocketQuat), first-order hold, free final time (the shipped SC.info).  -1: not available.
template <class PL>
int scSolvePersistentT(scpp_hip_ctx *c)
{
    if (c->stream_engine != SCPP_STREAM_PERSISTENT || c->mode != (SCPP_MODE_FOH | SCPP_MODE_VT) || c->sc.max_iterations <= 0)
        return -1;
    // Where it pays (measured, same box, alternating: profiles/r05_ab_persistent_batch_sizes.json): a COLD solve of a batch that fills the chip more than
    // once.  A wavefront integrates its instance's 49 segments one after the other here, where discretize_kernel spreads them over 49 wavefronts: below
    // ~3000 instances the chip is not full and the serial integration shows (cold solve of 256 / 1024 / 2048 instances: -29 % / -8 % / +-0, 4096 / 8192:
    // +4.6 % / +4.0 %), and in a warm-started solve (SC_sim: 7 interior-point iterations per sub-problem instead of 22) the integration is half of the
    // work (4096 instances: -3 %, 1024: -12 %).
    static const bool warm_too = std::getenv("SCPP_SC_PERSISTENT_WARM") != nullptr; // measurement hook (round 6)
    if ((c->sc_warm && !warm_too) || c->B < c->sc_persistent_min)
        return -1;
    const Range r = fullRange(c);
    ScPersistentArgs args;
    args.a = ipmArgs(c, 1, false, r, false);
    args.B = c->B;
    args.K = c->K;
    args.max_iterations = c->sc.max_iterations;
    args.disc_steps = c->disc_steps;
    args.ip = c->ip;
    args.A = c->A;
    args.Bm = c->Bm;
    args.C = c->C;
    args.S = c->S;
    args.Z = c->Z;
    args.active = c->active;
    using Model = typename PL::Model;
    using PT = typename PL::PersistTable;
    const size_t seg_b = ipm::segLdsBytes<PT>(c->K), disc_b = sizeof(DiscLds<Model, true, true>);
    const bool timed = spanBegin(c, 1, c->B, c->stream);
    hipLaunchKernelGGL((sc_persistent_kernel<Model, PT, true, true>), dim3(unsigned(c->B)), dim3(WAVE), seg_b > disc_b ? seg_b : disc_b, c->stream, args);
    spanEnd(c, timed, c->stream);
    return hipGetLastError() == hipSuccess ? 0 : SCPP_E_HIP;
}
int scSolvePersistent(scpp_hip_ctx *c)
{
    return withPlugin(c->model, [&](auto pl) {
        using PL = decltype(pl);
        if constexpr (PL::SC_PERSISTENT)
            return scSolvePersistentT<PL>(c);
        else
            return -1;
    });
}
} // namespace
} // extern "C++"

namespace
{
int ensurePools(scpp_hip_ctx *c, int pools);
}
int scpp_hip_sc_solve(scpp_hip_ctx *c, int *n_converged)
{
    DeviceGuard guard(c);
    if (!c || !c->sc_ready)
        return SCPP_E_STATE;
    {
        const int prc = scSolvePersistent(c);
        if (prc > 0 || prc < -1)
            return prc;
        if (prc == 0)
        {
            int n = 0;
            if (int rc = countActive(c, &n))
                return rc;
            c->last_active = n;
            return scpp_hip_sc_finish(c, n_converged);
        }
    }
    const int B = c->B;
    static const bool single_stream = std::getenv("SCPP_SC_SINGLE_STREAM") != nullptr; // measurement hook (round 6)
    if (B < 1024 || c->last_active != B || single_stream)
    {
        // small or partially masked batches: one stream, stop as soon as every instance has terminated
        int n_active = c->last_active;
        for (int it = 0; it < c->sc.max_iterations && n_active > 0; it++)
        {
            int rc = scpp_hip_sc_iterate(c, &n_active);
            if (rc)
                return rc;
        }
        return scpp_hip_sc_finish(c, n_converged);
    }
    // Large batches: the two halves run the loop on two streams, skewed by one kernel, so that the memory-bound
    // interior-point kernel of one half overlaps the ALU/LDS-bound discretisation of the other half (with two resident
    // waves per SIMD the second ipm wave buys nothing, a discretisation wave does).  No host synchronisation inside
    // the loop: instances that have terminated return at the top of both kernels.
    if (!c->stream2)
    {
        if (const char *e = std::getenv("SCPP_IPM_LDS_PAD"))
        {
            // a measurement knob: never-touched dynamic LDS that limits the ipm workgroups per CU.  Clamped so that pad + the LDS-resident segment
            // fields (up to 21.5 KB at K = 64) + the kernel's 3 KB of static LDS stay inside the 64 KB a workgroup may ask for -- an unclamped
            // value surfaced only as SCPP_E_HIP from the launch
            long seg_b = 0;
            (void)withPlugin(c->model, [&](auto pl) {
                seg_b = long(ipm::segLdsBytes<typename decltype(pl)::Table>(c->K));
                return 0;
            });
            const long want = std::atol(e), room = 65536 - 4096 - seg_b;
            c->ipm_lds_pad = unsigned(want < 0 ? 0 : (want > room ? room : want));
        }
        CHECK_HIP(hipStreamCreate(&c->stream2));
        CHECK_HIP(hipEventCreate(&c->ev_skew));
        CHECK_HIP(hipEventCreate(&c->ev_join));
    }
    // measurement hook (round 6): SCPP_SC_CHUNKS = n > 2 runs n chunks on n streams, each looping discretize -> solve on its own, started one
    // discretisation apart
    static const int n_chunks = std::getenv("SCPP_SC_CHUNKS") ? std::atoi(std::getenv("SCPP_SC_CHUNKS")) : 2;
    if (n_chunks > 2 && n_chunks <= 8)
    {
        if (int rc = ensurePools(c, n_chunks))
            return rc;
        std::vector<Range> ch;
        const int per = ((B + n_chunks - 1) / n_chunks + 7) & ~7;
        for (int i = 0, f = 0; i < n_chunks && f < B; i++, f += per)
            ch.push_back(Range{f, (f + per <= B ? per : B - f), i == 0 ? c->stream : c->pool_streams[size_t(i - 1)]});
        CHECK_HIP(hipEventRecord(c->pool_events[0], c->stream));
        for (size_t i = 1; i < ch.size(); i++)
            CHECK_HIP(hipStreamWaitEvent(ch[i].stream, c->pool_events[0], 0));
        for (int it = 0; it < c->sc.max_iterations; it++)
            for (size_t i = 0; i < ch.size(); i++)
            {
                int rc = discretizeDispatch(c, c->mode, c->ip + ipm::IP_PAR, ipm::IP_N, c->active, ch[i].count, ch[i]);
                if (rc)
                    return rc;
                if (it == 0 && i + 1 < ch.size())
                {
                    CHECK_HIP(hipEventRecord(c->pool_events[i + 1 < c->pool_events.size() ? i + 1 : 0], ch[i].stream));
                    CHECK_HIP(hipStreamWaitEvent(ch[i + 1].stream, c->pool_events[i + 1 < c->pool_events.size() ? i + 1 : 0], 0));
                }
                rc = launchIpm(c, 1, ch[i].count, false, ch[i], 0);
                if (rc)
                    return rc;
            }
        for (size_t i = 1; i < ch.size(); i++)
        {
            CHECK_HIP(hipEventRecord(c->pool_events[i], ch[i].stream));
            CHECK_HIP(hipStreamWaitEvent(c->stream, c->pool_events[i], 0));
        }
        int n = 0;
        if (int rc = countActive(c, &n))
            return rc;
        c->last_active = n;
        return scpp_hip_sc_finish(c, n_converged);
    }
    const int h0 = (B / 2 + 7) & ~7; // keep the XCD groups of 8 instances intact
    const Range r0{0, h0, c->stream}, r1{h0, B - h0, c->stream2};
    // everything enqueued so far (set-up kernels, uploads) is on the main stream
    CHECK_HIP(hipEventRecord(c->ev_skew, c->stream));
    CHECK_HIP(hipStreamWaitEvent(c->stream2, c->ev_skew, 0));
    for (int it = 0; it < c->sc.max_iterations; it++)
    {
        int rc = discretizeDispatch(c, c->mode, c->ip + ipm::IP_PAR, ipm::IP_N, c->active, r0.count, r0);
        if (rc)
            return rc;
        if (it == 0)
        {
            // skew: the second half starts when the first half's first discretisation is done
            CHECK_HIP(hipEventRecord(c->ev_skew, c->stream));
            CHECK_HIP(hipStreamWaitEvent(c->stream2, c->ev_skew, 0));
        }
        rc = discretizeDispatch(c, c->mode, c->ip + ipm::IP_PAR, ipm::IP_N, c->active, r1.count, r1);
        if (rc)
            return rc;
        // optional occupancy limiter (SCPP_IPM_LDS_PAD bytes of untouched dynamic LDS per ipm workgroup); default 0
        const unsigned pad = c->ipm_lds_pad;
        rc = launchIpm(c, 1, r0.count, false, r0, pad);
        if (rc)
            return rc;
        rc = launchIpm(c, 1, r1.count, false, r1, pad);
        if (rc)
            return rc;
    }
    CHECK_HIP(hipEventRecord(c->ev_join, c->stream2));
    CHECK_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    int n = 0;
    int rc = countActive(c, &n);
    if (rc)
        return rc;
    c->last_active = n;
    return scpp_hip_sc_finish(c, n_converged);
}

// ---------------------------------------------------------------- SCvx
namespace
{
SCvxBuffers scvxBuffers(scpp_hip_ctx *c)
{
    SCvxBuffers v;
    v.Xold = c->vx_Xold;
    v.Uold = c->vx_Uold;
    v.tr = c->vx_tr;
    v.last_cost = c->vx_last;
    v.cost = c->vx_cost;
    v.info = c->vx_info;
    v.has_last = c->vx_has_last;
    v.needs_disc = c->vx_needs_disc;
    v.solves = c->vx_solves;
    v.iter_ring = c->vx_record ? c->vx_iter_ring : nullptr;
    v.iter_count = c->vx_iter_count;
    v.iter_cap = c->vx_iter_cap;
    return v;
}
} // namespace

namespace
{
// what SCvx set-up does for every model: buffers, options, the SC-style options the shared set-up kernels read
int scvxSetupCommon(scpp_hip_ctx *c, const scpp_scvx_opts *so, const double *x_init, int B, int warm_start)
{
    /* first-order hold (shipped) or zero-order hold (SCvxProblem.cpp:32-35: no C in the dynamics; :58-68: the trust-region loop runs
       over the K-1 inputs; SCvxAlgorithm.cpp:269: u1 = u0 in the nonlinear cost) */
    if (so->K != c->K)
        return SCPP_E_UNSUPPORTED;
    if (warm_start && (!c->scvx_ready || B != c->B))
        return SCPP_E_STATE;
    if (int rc = ensureWorkspace(c))
        return rc;
    if (!c->vx_tr)
    {
        const size_t Bm = size_t(c->Bmax), K = size_t(c->K);
        int rc = 0;
        rc |= devAlloc(&c->vx_Xold, Bm * K * size_t(c->nx));
        rc |= devAlloc(&c->vx_Uold, Bm * K * size_t(c->nu));
        rc |= devAlloc(&c->vx_tr, Bm);
        rc |= devAlloc(&c->vx_last, Bm);
        rc |= devAlloc(&c->vx_cost, Bm);
        rc |= devAlloc(&c->vx_info, Bm * 4);
        rc |= devAlloc(&c->vx_has_last, Bm);
        rc |= devAlloc(&c->vx_needs_disc, Bm);
        rc |= devAlloc(&c->vx_solves, Bm);
        if (rc)
            return SCPP_E_HIP;
    }
    if (c->vx_record && (!c->vx_iter_ring || c->vx_iter_cap < so->max_iterations + 1))
    {
        if (c->vx_iter_ring)
            (void)hipFree(c->vx_iter_ring);
        c->vx_iter_ring = nullptr;
        c->vx_iter_cap = so->max_iterations + 1;
        int rc = devAlloc(&c->vx_iter_ring, size_t(c->Bmax) * size_t(c->vx_iter_cap) * scvxIterRecordDoubles(c->K, c->nx, c->nu));
        if (!c->vx_iter_count)
            rc |= devAlloc(&c->vx_iter_count, size_t(c->Bmax));
        if (rc)
            return SCPP_E_HIP;
    }
    c->B = B;
    c->scvx = *so;
    // the trajectory / parameter set-up is the SC one (same model code); SCvx specifics are applied on top
    scpp_sc_opts sc{};
    sc.K = so->K;
    sc.free_final_time = 0;
    sc.interpolate_input = so->interpolate_input;
    sc.nondimensionalize = so->nondimensionalize;
    sc.max_iterations = so->max_iterations;
    sc.weight_time = 1.;
    sc.weight_trust_region_time = 1.;
    sc.weight_trust_region_trajectory = 0.;
    sc.weight_virtual_control = so->weight_virtual_control;
    sc.nu_tol = 0.;
    sc.delta_tol = 0.;
    c->sc = sc;
    c->mode = so->interpolate_input ? SCPP_MODE_FOH : 0; // fixed final time either way (SCvxAlgorithm.cpp:52: dd.initialize(K, interpolate_input, false))
    CHECK_HIP(hipMemcpyAsync(c->x_init, x_init, size_t(B) * size_t(c->nx) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemsetAsync(c->S, 0, size_t(B) * (c->K - 1) * size_t(c->nx) * sizeof(double), c->stream));
    if (!so->interpolate_input) // zero-order hold: no C in the dynamics (discretizationData.hpp:56-59); the discretisation does not write it
        CHECK_HIP(hipMemsetAsync(c->C, 0, size_t(B) * (c->K - 1) * size_t(c->nx) * size_t(c->nu) * sizeof(double), c->stream));
    if (!warm_start)
        CHECK_HIP(hipMemsetAsync(c->ipm_warm, 0, size_t(c->Bmax) * sizeof(int), c->stream));
    return SCPP_OK;
}
int scvxSetupDone(scpp_hip_ctx *c, double final_time, int B, int warm_start)
{
    hipLaunchKernelGGL(scvx_setup_kernel, dim3(unsigned((B + 63) / 64)), dim3(64), 0, c->stream, scBuffers(c), scvxBuffers(c), c->scvx, final_time,
                       warm_start);
    if (c->vx_record) // all_td.push_back(td) before the first iteration (SCvxAlgorithm.cpp:192); the record restarts with every set-up
        hipLaunchKernelGGL(scvx_record_initial_kernel, dim3(unsigned(B)), dim3(WAVE), 0, c->stream, scBuffers(c), scvxBuffers(c), c->nx, c->nu);
    c->sc_ready = false; // the SC entry points must not be mixed with an SCvx set-up
    c->scvx_ready = true;
    c->par_from_ip = true;
    c->last_active = B;
    return hipGetLastError() == hipSuccess ? SCPP_OK : SCPP_E_HIP;
}
} // namespace

extern "C++"
{
namespace
{
template <class PL>
int scvxSetupT(scpp_hip_ctx *c, const typename PL::Params *mp, const scpp_scvx_opts *so, const double *x_init, int B, int warm_start)
{
    DeviceGuard guard(c);
    if (!c || !mp || !so || !x_init || B < 1 || B > c->Bmax)
        return SCPP_E_ARG;
    if (c->model != PL::ID || !PL::supported(*mp))
        return SCPP_E_UNSUPPORTED;
    if (int rc = scvxSetupCommon(c, so, x_init, B, warm_start))
        return rc;
    paramsOf<PL>(c) = *mp;
    hipLaunchKernelGGL((sc_setup_kernel<PL>), dim3(unsigned((B + 63) / 64)), dim3(64), 0, c->stream, scBuffers(c), paramsOf<PL>(c), c->sc, warm_start);
    return scvxSetupDone(c, mp->final_time, B, warm_start);
}
} // namespace
} // extern "C++"

int scpp_hip_scvx_setup(scpp_hip_ctx *c, const scpp_rocketquat_params *mp, const scpp_scvx_opts *so, const double *x_init,
                        int B, int warm_start)
{
    return scvxSetupT<RocketQuatPlugin>(c, mp, so, x_init, B, warm_start);
}

int scpp_hip_scvx_setup_rocket2d(scpp_hip_ctx *c, const scpp_rocket2d_params *mp, const scpp_scvx_opts *so, const double *x_init,
                                 int B, int warm_start)
{
    return scvxSetupT<Rocket2dPlugin>(c, mp, so, x_init, B, warm_start);
}

int scpp_hip_scvx_setup_lander3dof(scpp_hip_ctx *c, const scpp_lander3dof_params *mp, const scpp_scvx_opts *so, const double *x_init,
                                   int B, int warm_start)
{
    return scvxSetupT<Lander3dofPlugin>(c, mp, so, x_init, B, warm_start);
}

namespace
{
SCvxBuffers scvxBuffersRange(scpp_hip_ctx *c, Range r)
{
    SCvxBuffers v = scvxBuffers(c);
    const size_t f = size_t(r.first), K = size_t(c->K);
    v.Xold += f * K * size_t(c->nx); // the pitch launchIpm snapshots with (ipm_kernel writes old_td at slot * K * nx)
    v.Uold += f * K * size_t(c->nu);
    v.tr += f;
    v.last_cost += f;
    v.cost += f;
    v.info += f * 4;
    v.has_last += f;
    v.needs_disc += f;
    v.solves += f;
    if (v.iter_ring)
    {
        v.iter_ring += f * size_t(v.iter_cap) * scvxIterRecordDoubles(c->K, c->nx, c->nu);
        v.iter_count += f;
    }
    return v;
}

// one SCvx round (one sub-problem solve of every active instance) of an instance range on its stream: instances whose
// previous candidate was rejected re-solve on their old discretisation (needs_disc = 0), the others start a new iteration.
// Three launches: discretisation, interior-point solve (which first snapshots td -> old_td), and the fused nonlinear-cost /
// accept-reject kernel (which rolls a rejected candidate back).
int scvxRound(scpp_hip_ctx *c, Range r)
{
    int rc = discretizeDispatch(c, c->mode & SCPP_MODE_FOH, c->ip + ipm::IP_PAR, ipm::IP_N, c->vx_needs_disc, r.count, r);
    if (rc)
        return rc;
    rc = launchIpm(c, 0, r.count, true, r, 0, true);
    if (rc)
        return rc;
    const SCBuffers b = scBuffersRange(c, r);
    const SCvxBuffers v = scvxBuffersRange(c, r);
    (void)withPlugin(c->model, [&](auto pl) {
        hipLaunchKernelGGL((scvx_cost_update_kernel<typename decltype(pl)::Model>), dim3(unsigned(r.count)), dim3(WAVE), 0, r.stream, b, v, c->scvx);
        return 0;
    });
    return hipGetLastError() == hipSuccess ? 0 : SCPP_E_HIP;
}

int ensurePoll(scpp_hip_ctx *c)
{
    if (!c->h_poll)
        CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_poll), 64 * sizeof(int), 0));
    return 0;
}
// slot pools beyond the first run on their own streams
int ensurePools(scpp_hip_ctx *c, int pools)
{
    while (int(c->pool_streams.size()) < pools - 1)
    {
        hipStream_t st;
        CHECK_HIP(hipStreamCreate(&st));
        c->pool_streams.push_back(st);
    }
    while (int(c->pool_events.size()) < pools)
    {
        hipEvent_t e;
        CHECK_HIP(hipEventCreate(&e));
        c->pool_events.push_back(e);
    }
    return 0;
}
} // namespace

extern "C++"
{
namespace
{
// Launch of the persistent kernel.  Instantiated for both models and both input holds since round 6 (until then: RocketQuat with first-order hold
// only, everything else ran the pool engine): T = the refill traits of the model, Model = its flow-map plugin, PF / PZ = the solver tables of the
// first-order / zero-order hold.  Dynamic LDS: the solver's LDS-resident segment fields during a solve, the integration's stage values / tables during
// multipleShooting (never live together).
template <class T, class Model, class PF, class PZ>
int launchPersistentT(scpp_hip_ctx *c, const ipm::KernelArgs &a, const SCBuffers &b, const SCvxBuffers &v, const StreamQueue &q,
                      const typename T::Params &mp, const scpp_sc_opts &sc, const scpp_scvx_opts &so, const PersistentOut &o, int S)
{
    const PersistentArgs<T> args{a, b, v, q, mp, sc, so, o};
    const bool foh = (c->mode & SCPP_MODE_FOH) != 0;
    const size_t seg_b = ipm::segLdsBytes<PF>(c->K), disc_b = foh ? sizeof(DiscLds<Model, true, false>) : sizeof(DiscLds<Model, false, false>);
    // SCPP_PERSIST_LDS_PAD (bytes of dynamic LDS that are never touched): measurement hook -- it only limits how many wavefronts fit on a CU
    // (profiles/r06_occupancy_curve.json: the throughput-against-resident-wavefronts curve of DESIGN.md 5)
    size_t pad = 0;
    if (const char *e = std::getenv("SCPP_PERSIST_LDS_PAD"))
        pad = size_t(std::atoi(e) > 0 ? std::atoi(e) : 0);
    const size_t lds = (seg_b > disc_b ? seg_b : disc_b) + pad;
    if (foh)
        hipLaunchKernelGGL((scvx_persistent_kernel<T, Model, PF, true>), dim3(unsigned(S)), dim3(WAVE), lds, c->stream, args);
    else
        hipLaunchKernelGGL((scvx_persistent_kernel<T, Model, PZ, false>), dim3(unsigned(S)), dim3(WAVE), lds, c->stream, args);
    return hipGetLastError() == hipSuccess ? 0 : SCPP_E_HIP;
}
template <class PL>
int launchPersistent(scpp_hip_ctx *c, const ipm::KernelArgs &a, const SCBuffers &b, const SCvxBuffers &v, const StreamQueue &q,
                     const typename PL::Params &mp, const scpp_sc_opts &sc, const scpp_scvx_opts &so, const PersistentOut &o, int S)
{
    if constexpr (PL::SCVX_PERSISTENT)
    {
        using PT = typename PL::PersistTable;
        return launchPersistentT<PL, typename PL::Model, PT, ipm::ZeroOrderHold<PT>>(c, a, b, v, q, mp, sc, so, o, S);
    }
    else
        return -1; // (persistentAvailable() said so before anything was set up)
}
// the persistent kernel runs a job when the context's engine asks for it and there is at least one iteration to run
// (and the model's plugin instantiates the kernel: SCVX_PERSISTENT)
bool persistentAvailable(const scpp_hip_ctx *c, int max_iterations)
{
    return c->stream_engine == SCPP_STREAM_PERSISTENT && max_iterations > 0 && withPlugin(c->model, [](auto pl) { return decltype(pl)::SCVX_PERSISTENT ? 0 : 1; }) == 0;
}

// queue arrays of the streaming engine (also used, with an empty queue, by the persistent batch solve)
int ensureQueue(scpp_hip_ctx *c)
{
    if (c->q_counters)
        return 0;
    int a = 0;
    a |= devAlloc(&c->q_counters, 4);
    a |= devAlloc(&c->q_slot_inst, size_t(c->Bmax));
    return a ? SCPP_E_HIP : 0;
}
// scpp_hip_scvx_solve on the persistent kernel: the batch is the slots, the queue is empty -- every wavefront takes its one instance through the
// whole of SCvxAlgorithm::solve and leaves; no rounds, no host polling.  -1: not available for this configuration.
int scvxSolvePersistent(scpp_hip_ctx *c)
{
    if (!persistentAvailable(c, c->scvx.max_iterations))
        return -1;
    if (int rc = ensureQueue(c))
        return rc;
    if (!c->persist_shares && devAlloc(&c->persist_shares, 8))
        return SCPP_E_HIP;
    CHECK_HIP(hipMemsetAsync(c->persist_shares, 0, 8 * sizeof(double), c->stream));
    CHECK_HIP(hipMemsetAsync(c->q_counters, 0, 4 * sizeof(int), c->stream));
    CHECK_HIP(hipMemsetAsync(c->q_slot_inst, 0xFF, size_t(c->Bmax) * sizeof(int), c->stream));
    const Range r = fullRange(c);
    const SCBuffers b = scBuffersRange(c, r);
    const SCvxBuffers v = scvxBuffersRange(c, r);
    StreamQueue q;
    q.N = 0; // nothing to pull: a slot whose instance has terminated finds the queue empty and its wavefront leaves
    q.x_init = nullptr;
    q.rows = nullptr;
    q.head = c->q_counters;
    q.done = c->q_counters + 1;
    q.nconv = c->q_counters + 2;
    q.slot_inst = c->q_slot_inst;
    q.warm = c->ipm_warm;
    const ipm::KernelArgs a = ipmArgs(c, 0, true, r, true);
    PersistentOut o;
    o.A = c->A;
    o.Bm = c->Bm;
    o.C = c->C;
    o.S = c->S;
    o.Z = c->Z;
    o.disc_steps = c->disc_steps;
    o.shares = c->persist_shares;
    const bool timed = spanBegin(c, 1, c->B, c->stream);
    const int rc = withPlugin(c->model, [&](auto pl) {
        using PL = decltype(pl);
        return launchPersistent<PL>(c, a, b, v, q, paramsOf<PL>(c), c->sc, c->scvx, o, c->B);
    });
    spanEnd(c, timed, c->stream);
    return rc;
}
} // namespace
} // extern "C++"

int scpp_hip_scvx_solve(scpp_hip_ctx *c, int *n_converged)
{
    DeviceGuard guard(c);
    if (!c || !c->scvx_ready)
        return SCPP_E_STATE;
    if (int rc = ensurePoll(c))
        return rc;
    if (int rc = ensurePools(c, 1))
        return rc;
    // an instance is retired at the cap (scvxDecide); max_iterations <= 0: no iteration at all (the initial trajectory comes back)
    // (retired on a REJECTION with solves >= CAP x max_iterations: up to max_iterations accepted solves can follow the last rejection
    // below the cap, hence CAP + 1)
    long max_rounds = c->scvx.max_iterations > 0 ? long(c->scvx.max_iterations) * (SCVX_SOLVE_CAP + 1) + 8 : 0;
    {
        const int prc = scvxSolvePersistent(c); // one launch where the persistent kernel exists (RocketQuat, first-order hold) ...
        if (prc > 0 || prc < -1)
            return prc;
        if (prc == 0)
            max_rounds = 0; // ... the rounds below are the pool-style loop of every other configuration
    }
    // One stream (measured: the two-stream skewed pipeline of scpp_hip_sc_solve loses here, rounds late in the run have few
    // active instances and are latency-bound either way).  The host does not wait for a round before enqueueing the next:
    // the active count is read back asynchronously every POLL rounds and looked at one poll later, so the device never
    // idles on the host; rounds enqueued after the last instance terminated return at the top of every kernel.
    constexpr int POLL = 4;
    bool pending = false;
    volatile int *h_active = c->h_poll;
    for (long round = 0; round < max_rounds; round++)
    {
        int rc = scvxRound(c, fullRange(c));
        if (rc)
            return rc;
        if (round % POLL == POLL - 1 || c->B < 64)
        {
            if (pending)
            {
                CHECK_HIP(hipEventSynchronize(c->pool_events[0]));
                if (*h_active == 0)
                    break;
            }
            hipLaunchKernelGGL(count_active_kernel, dim3(1), dim3(256), 0, c->stream, c->B, (const int *)c->active, c->counter);
            CHECK_HIP(hipMemcpyAsync(c->h_poll, c->counter, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            CHECK_HIP(hipEventRecord(c->pool_events[0], c->stream));
            pending = true;
        }
    }
    if (c->scvx.nondimensionalize)
        launchRedim(c);
    CHECK_HIP(hipStreamSynchronize(c->stream));
    {
        int n = 0;
        int rc = countActive(c, &n);
        if (rc)
            return rc;
        c->last_active = n;
    }
    if (n_converged)
    {
        std::vector<int> conv(c->B);
        CHECK_HIP(hipMemcpy(conv.data(), c->converged, size_t(c->B) * sizeof(int), hipMemcpyDeviceToHost));
        int n = 0;
        for (int x : conv)
            n += x;
        *n_converged = n;
    }
    return SCPP_OK;
}

// ---------------------------------------------------------------- SCvx streaming engine (continuous batching)
extern "C++"
{
namespace
{
template <class T>
int scvxSolveStream(scpp_hip_ctx *c, const typename T::Params *mp, const scpp_scvx_opts *so, const double *x_init, int N, int slots,
                    int pools, int *n_converged)
{
    constexpr int NX = T::NX, NU = T::NU;
    const int S = slots > 0 ? (slots < N ? slots : N) : (c->Bmax < N ? c->Bmax : N);
    c->q_N = 0; // whatever fails from here on, the rows of a previous job are no longer valid (stream_download -> SCPP_E_STATE)
    // the engine's per-slot state is the batch state of scvx_setup: set it up on the first S instances' worth of slots
    // WITHOUT starting them (every slot starts empty and is filled by the first refill)
    if (int rc = scvxSetupT<T>(c, mp, so, x_init, S, 0))
        return rc;
    const size_t K = size_t(c->K), rowd = size_t(streamRowDoubles(c->K, NX, NU));
    if (c->q_cap < size_t(N))
    {
        if (c->q_xinit)
            (void)hipFree(c->q_xinit);
        if (c->q_rows)
            (void)hipFree(c->q_rows);
        c->q_xinit = c->q_rows = nullptr;
        c->q_cap = 0;
        int a = 0;
        a |= devAlloc(&c->q_xinit, size_t(N) * NX);
        a |= devAlloc(&c->q_rows, size_t(N) * rowd);
        if (a)
            return SCPP_E_HIP;
        c->q_cap = size_t(N);
    }
    if (int rc = ensureQueue(c))
        return rc;
    (void)K;
    CHECK_HIP(hipMemcpyAsync(c->q_xinit, x_init, size_t(N) * NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemsetAsync(c->q_counters, 0, 4 * sizeof(int), c->stream));
    CHECK_HIP(hipMemsetAsync(c->q_slot_inst, 0xFF, size_t(c->Bmax) * sizeof(int), c->stream)); // -1: empty
    CHECK_HIP(hipMemsetAsync(c->active, 0, size_t(c->Bmax) * sizeof(int), c->stream));
    CHECK_HIP(hipMemsetAsync(c->vx_needs_disc, 0, size_t(c->Bmax) * sizeof(int), c->stream));
    // pools: disjoint slot ranges, each on its own stream, all pulling from the one queue.  Their rounds drift apart, so the
    // memory-bound interior-point kernel of one pool overlaps the ALU/LDS-bound discretisation of another without any of
    // the explicit skewing scpp_hip_sc_solve needs.
    // pools == 0 (and no SCPP_STREAM_POOLS): heuristic -- pools of about 2/3 of the 2048 wavefronts the chip holds (1365 slots), so
    // that one and a half ipm_kernel launches are resident at any time and a launch's tail (instances that need more
    // interior-point iterations) is covered by the next pool's kernels: measured on MI355X, same box, alternating runs
    // (profiles/r03_ab_pools.json), 8192 slots: 2 pools 4035, 3 pools 4140 / 4017, 4 pools 3722, 5 pools 3914, 6 pools 4070 / 4117,
    // 7 pools 4022, 8 pools 3482 converged/s (second figures: another box); 4096 slots: 1 pool 3365, 2 pools 3918, 3 pools 4158.
    // Pool sizes that divide the resident capacity exactly (2048, 1024) are the bad ones.  Below 2731 slots: one pool.
    // An explicit request (argument or environment) is honoured as given and only clamped to the slot count and to 8.
    int P = pools;
    if (const char *e = std::getenv("SCPP_STREAM_POOLS"))
        P = std::atoi(e) > 0 ? std::atoi(e) : P;
    if (P <= 0)
        P = S >= 2731 ? int((double(S) + 682.) / 1365.34) : 1;
    if (P > S)
        P = S;
    if (P > 8)
        P = 8;
    if (int rc = ensurePoll(c))
        return rc;
    if (int rc = ensurePools(c, P))
        return rc;
    std::vector<Range> pool;
    {
        // exactly P pools (P <= S), sizes differing by at most one XCD group: whole groups of 8 slots where the job is large enough
        // (a performance nicety), single slots otherwise
        const int unit = S >= 16 * P ? 8 : 1;
        const int units = S / unit, rem = S - units * unit; // the last pool also takes the slots beyond the last whole group
        int first = 0;
        for (int p = 0; p < P; p++)
        {
            int cnt = (units / P + (p < units % P ? 1 : 0)) * unit;
            if (p == P - 1)
                cnt += rem;
            pool.push_back(Range{first, cnt, p == 0 ? c->stream : c->pool_streams[size_t(p - 1)]});
            first += cnt;
        }
        P = int(pool.size());
    }
    // everything enqueued so far is on the main stream
    CHECK_HIP(hipEventRecord(c->pool_events[0], c->stream));
    for (int p = 1; p < P; p++)
        CHECK_HIP(hipStreamWaitEvent(pool[size_t(p)].stream, c->pool_events[0], 0));
    c->q_N = N; // from here on every failure goes through fail(), which resets it
    StreamQueue q;
    q.N = N;
    q.x_init = c->q_xinit;
    q.rows = c->q_rows;
    q.head = c->q_counters;
    q.done = c->q_counters + 1;
    q.nconv = c->q_counters + 2;
    scpp_sc_opts sc = c->sc; // as built by scvx_setup
    for (int i = 0; i < 4; i++)
        c->stream_ticks[i] = 0;
    // (an explicit pool count -- argument or SCPP_STREAM_POOLS -- asks for the pool engine)
    // (whether the persistent kernel runs is decided BEFORE a span is opened or persist_shares is touched: until round 5 a configuration the kernel
    // was not instantiated for recorded an empty span as one launch over N instances -- ADVICE r5)
    if (persistentAvailable(c, so->max_iterations) && pools == 0 && !std::getenv("SCPP_STREAM_POOLS"))
    {
        // ONE launch: a wavefront per slot takes instance after instance through the whole SCvx loop (scvx_persistent.h)
        if (!c->persist_shares && devAlloc(&c->persist_shares, 8))
            return SCPP_E_HIP;
        CHECK_HIP(hipMemsetAsync(c->persist_shares, 0, 8 * sizeof(double), c->stream));
        Range r = fullRange(c);
        r.count = S;
        SCBuffers b = scBuffersRange(c, r);
        b.B = S;
        SCvxBuffers v = scvxBuffersRange(c, r);
        v.iter_ring = nullptr; // a slot is re-used for many instances: iterates are recorded by the batch entry point only
        StreamQueue qp = q;
        qp.slot_inst = c->q_slot_inst;
        qp.warm = c->ipm_warm;
        const ipm::KernelArgs a = ipmArgs(c, 0, true, r, true);
        PersistentOut o;
        o.A = c->A;
        o.Bm = c->Bm;
        o.C = c->C;
        o.S = c->S;
        o.Z = c->Z;
        o.disc_steps = c->disc_steps;
        o.shares = c->persist_shares;
        const bool timed = spanBegin(c, 1, N, c->stream);
        const int prc = launchPersistent<T>(c, a, b, v, qp, *mp, sc, *so, o, S);
        spanEnd(c, timed, c->stream);
        {
            int counters[4] = {0, 0, 0, 0};
            auto bad = [&](int rc) {
                (void)hipStreamSynchronize(c->stream);
                c->q_N = 0;
                return rc;
            };
            if (prc != 0)
                return bad(prc);
            if (hipMemcpyAsync(counters, c->q_counters, sizeof counters, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipMemcpyAsync(c->stream_ticks, c->persist_shares, sizeof c->stream_ticks, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess)
                return bad(SCPP_E_HIP);
            c->stream_rounds = 1;
            c->stream_pools = 0; // no pools: one persistent launch
            c->last_active = 0;
            if (n_converged)
                *n_converged = counters[2];
            if (counters[1] != N)
                return bad(SCPP_E_STATE);
            return SCPP_OK;
        }
    }
    // an instance needs at most max_iterations accepted + ~log2 rejected solves each; the queue drains in ceil(N/S) waves
    const long per_instance = long(so->max_iterations) * (SCVX_SOLVE_CAP + 1) + 8; // retired at the cap on a rejection (scvxDecide) + the accepted solves after it
    const long max_rounds = per_instance * ((N + S - 1) / S + 1);
    constexpr int POLL = 4;
    bool pending = false;
    volatile int *h_done = c->h_poll;
    long round = 0;
    // every failure exit: wait for whatever is still in flight on the pool streams (later calls on the context must not race
    // with it) and invalidate the result rows, so that stream_download / stream_rows report SCPP_E_STATE instead of stale rows
    auto fail = [&](int rc) {
        for (int p = 0; p < P; p++)
            (void)hipStreamSynchronize(pool[size_t(p)].stream);
        (void)hipStreamSynchronize(c->stream);
        c->q_N = 0;
        c->stream_rounds = round;
        c->stream_pools = P;
        return rc;
    };
    for (; round < max_rounds; round++)
    {
        for (int p = 0; p < P; p++)
        {
            const Range r = pool[size_t(p)];
            const SCBuffers b = scBuffersRange(c, r);
            SCvxBuffers v = scvxBuffersRange(c, r);
            v.iter_ring = nullptr;
            StreamQueue qp = q;
            qp.slot_inst = c->q_slot_inst + r.first;
            qp.warm = c->ipm_warm + r.first;
            hipLaunchKernelGGL((scvx_stream_refill_kernel<T>), dim3(unsigned(r.count)), dim3(WAVE), 0, r.stream, b, v, qp, *mp, sc, *so);
            if (int rc = scvxRound(c, r))
                return fail(rc);
        }
        if (round % POLL == POLL - 1)
        {
            if (pending)
            {
                int done = 0;
                for (int p = 0; p < P; p++)
                {
                    if (hipEventSynchronize(c->pool_events[size_t(p)]) != hipSuccess)
                        return fail(SCPP_E_HIP);
                    done = h_done[p] > done ? h_done[p] : done;
                }
                if (done >= N)
                    break;
            }
            for (int p = 0; p < P; p++)
            {
                if (hipMemcpyAsync(c->h_poll + p, c->q_counters + 1, sizeof(int), hipMemcpyDeviceToHost, pool[size_t(p)].stream) != hipSuccess ||
                    hipEventRecord(c->pool_events[size_t(p)], pool[size_t(p)].stream) != hipSuccess)
                    return fail(SCPP_E_HIP);
            }
            pending = true;
        }
    }
    c->stream_rounds = round;
    c->stream_pools = P;
    for (int p = 1; p < P; p++)
    {
        if (hipEventRecord(c->pool_events[size_t(p)], pool[size_t(p)].stream) != hipSuccess ||
            hipStreamWaitEvent(c->stream, c->pool_events[size_t(p)], 0) != hipSuccess)
            return fail(SCPP_E_HIP);
    }
    int counters[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(counters, c->q_counters, sizeof counters, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return fail(SCPP_E_HIP);
    c->last_active = 0;
    if (n_converged)
        *n_converged = counters[2];
    if (counters[1] != N) // round cap hit with instances still running: cannot happen with finite max_iterations
        return fail(SCPP_E_STATE);
    return SCPP_OK;
}
} // namespace
} // extern "C++"

int scpp_hip_scvx_solve_stream(scpp_hip_ctx *c, const scpp_rocketquat_params *mp, const scpp_scvx_opts *so, const double *x_init,
                               int N, int slots, int pools, int *n_converged)
{
    DeviceGuard guard(c);
    if (!c || !mp || !so || !x_init || N < 1 || slots < 0 || slots > c->Bmax || pools < 0 || pools > 8)
        return SCPP_E_ARG;
    if (c->model != RocketQuatPlugin::ID || !RocketQuatPlugin::supported(*mp))
        return SCPP_E_UNSUPPORTED;
    return scvxSolveStream<RocketQuatPlugin>(c, mp, so, x_init, N, slots, pools, n_converged);
}

int scpp_hip_scvx_solve_stream_rocket2d(scpp_hip_ctx *c, const scpp_rocket2d_params *mp, const scpp_scvx_opts *so, const double *x_init,
                                        int N, int slots, int pools, int *n_converged)
{
    DeviceGuard guard(c);
    if (!c || !mp || !so || !x_init || N < 1 || slots < 0 || slots > c->Bmax || pools < 0 || pools > 8)
        return SCPP_E_ARG;
    if (c->model != Rocket2dPlugin::ID || !Rocket2dPlugin::supported(*mp))
        return SCPP_E_UNSUPPORTED;
    return scvxSolveStream<Rocket2dPlugin>(c, mp, so, x_init, N, slots, pools, n_converged);
}

int scpp_hip_scvx_solve_stream_lander3dof(scpp_hip_ctx *c, const scpp_lander3dof_params *mp, const scpp_scvx_opts *so, const double *x_init,
                                          int N, int slots, int pools, int *n_converged)
{
    DeviceGuard guard(c);
    if (!c || !mp || !so || !x_init || N < 1 || slots < 0 || slots > c->Bmax || pools < 0 || pools > 8)
        return SCPP_E_ARG;
    if (c->model != Lander3dofPlugin::ID || !Lander3dofPlugin::supported(*mp))
        return SCPP_E_UNSUPPORTED;
    return scvxSolveStream<Lander3dofPlugin>(c, mp, so, x_init, N, slots, pools, n_converged);
}

int scpp_hip_stream_rows(scpp_hip_ctx *c, void **rows, int *row_doubles, int *n)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_E_ARG;
    if (!c->q_rows || c->q_N < 1)
        return SCPP_E_STATE;
    if (rows)
        *rows = c->q_rows;
    if (row_doubles)
        *row_doubles = streamRowDoubles(c->K, c->nx, c->nu);
    if (n)
        *n = c->q_N;
    return SCPP_OK;
}

int scpp_hip_stream_info(scpp_hip_ctx *c, long long *rounds, int *pools)
{
    if (!c)
        return SCPP_E_ARG;
    if (rounds)
        *rounds = c->stream_rounds;
    if (pools)
        *pools = c->stream_pools;
    return SCPP_OK;
}

int scpp_hip_stream_download(scpp_hip_ctx *c, double *rows, int first, int count)
{
    DeviceGuard guard(c);
    if (!c || !rows || first < 0 || count < 1)
        return SCPP_E_ARG;
    if (!c->q_rows || first + count > c->q_N)
        return SCPP_E_STATE;
    const size_t rowd = size_t(streamRowDoubles(c->K, c->nx, c->nu));
    CHECK_HIP(hipStreamSynchronize(c->stream));
    CHECK_HIP(hipMemcpy(rows, c->q_rows + size_t(first) * rowd, size_t(count) * rowd * sizeof(double), hipMemcpyDeviceToHost));
    return SCPP_OK;
}

int scpp_hip_scvx_download_state(scpp_hip_ctx *c, double *trust_region, double *nonlinear_cost, int32_t *solves, double *last_decision)
{
    DeviceGuard guard(c);
    if (!c || !c->scvx_ready)
        return SCPP_E_STATE;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    const size_t B = size_t(c->B);
    if (trust_region)
        CHECK_HIP(hipMemcpy(trust_region, c->vx_tr, B * sizeof(double), hipMemcpyDeviceToHost));
    if (nonlinear_cost)
        CHECK_HIP(hipMemcpy(nonlinear_cost, c->vx_last, B * sizeof(double), hipMemcpyDeviceToHost));
    if (solves)
        CHECK_HIP(hipMemcpy(solves, c->vx_solves, B * sizeof(int), hipMemcpyDeviceToHost));
    if (last_decision)
        CHECK_HIP(hipMemcpy(last_decision, c->vx_info, B * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return SCPP_OK;
}

int scpp_hip_scvx_record_iterates(scpp_hip_ctx *c, int enable)
{
    if (!c || (enable != 0 && enable != 1))
        return SCPP_E_ARG;
    c->vx_record = enable != 0;
    c->scvx_ready = false; // takes effect with the next scvx_setup (which allocates the record and stores the initial trajectory)
    return SCPP_OK;
}

int scpp_hip_scvx_download_iterates(scpp_hip_ctx *c, int first, int count, int capacity, double *X, double *U, double *scalars, int32_t *n_iterates)
{
    DeviceGuard guard(c);
    if (!c || first < 0 || count < 1 || capacity < 1)
        return SCPP_E_ARG;
    if (!c->scvx_ready || !c->vx_record || !c->vx_iter_ring || first + count > c->B)
        return SCPP_E_STATE;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    const size_t K = size_t(c->K), nx = size_t(c->nx), nu = size_t(c->nu), rec = scvxIterRecordDoubles(c->K, c->nx, c->nu), cap = size_t(c->vx_iter_cap);
    std::vector<int> cnt(static_cast<size_t>(count));
    CHECK_HIP(hipMemcpy(cnt.data(), c->vx_iter_count + first, size_t(count) * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<double> ip(size_t(count) * ipm::IP_N), buf(cap * rec);
    CHECK_HIP(hipMemcpy(ip.data(), c->ip + size_t(first) * ipm::IP_N, ip.size() * sizeof(double), hipMemcpyDeviceToHost));
    // factor that redimensionalises entry e of a state (is_u = 0) / input (is_u = 1) vector: the plugin's rx / ru
    auto redim = [&](int is_u, int e, double ms, double rs) {
        double f = 1.;
        (void)withPlugin(c->model, [&](auto pl) {
            f = is_u ? decltype(pl)::ru(e, ms, rs) : decltype(pl)::rx(e, ms, rs);
            return 0;
        });
        return f;
    };
    for (int b = 0; b < count; b++)
    {
        const int n = cnt[size_t(b)] < capacity ? cnt[size_t(b)] : capacity;
        if (n_iterates)
            n_iterates[b] = cnt[size_t(b)];
        if (n < 1 || (!X && !U && !scalars))
            continue;
        CHECK_HIP(hipMemcpy(buf.data(), c->vx_iter_ring + (size_t(first + b) * cap) * rec, size_t(n) * rec * sizeof(double), hipMemcpyDeviceToHost));
        // redimensionalizeTrajectory of every recorded trajectory (SCvxAlgorithm.cpp:247-255; rocketQuat.cpp:188-201, rocket2d.cpp:108-118): the
        // factors of the result rows (the plugin's rx / ru, csrc/sc_kernels.h); 1 when the run is not nondimensionalised
        const double ms = ip[size_t(b) * ipm::IP_N + ipm::IP_MSCALE], rs = ip[size_t(b) * ipm::IP_N + ipm::IP_RSCALE];
        for (int j = 0; j < n; j++)
        {
            if (scalars)
                for (int e = 0; e < SCVX_ITER_SCALARS; e++)
                    scalars[(size_t(b) * size_t(capacity) + size_t(j)) * SCVX_ITER_SCALARS + size_t(e)] = buf[size_t(j) * rec + K * (nx + nu) + size_t(e)];
            for (size_t k = 0; k < K; k++)
            {
                if (X)
                    for (size_t e = 0; e < nx; e++)
                    {
                        const double f = redim(0, int(e), ms, rs);
                        X[((size_t(b) * size_t(capacity) + size_t(j)) * K + k) * nx + e] = buf[size_t(j) * rec + k * nx + e] * f;
                    }
                if (U)
                    for (size_t e = 0; e < nu; e++)
                    {
                        const double f = redim(1, int(e), ms, rs);
                        U[((size_t(b) * size_t(capacity) + size_t(j)) * K + k) * nu + e] = buf[size_t(j) * rec + K * nx + k * nu + e] * f;
                    }
            }
        }
    }
    return SCPP_OK;
}

int scpp_hip_socp_solve(scpp_hip_ctx *c)
{
    DeviceGuard guard(c);
    if (!c || !c->sc_ready)
        return SCPP_E_STATE;
    int rc = launchIpm(c, 0, c->B);
    if (rc)
        return rc;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    return SCPP_OK;
}

int scpp_hip_download(scpp_hip_ctx *c, double *X, double *U, double *sigma, int32_t *sc_iters, double *nu_norm,
                      int32_t *converged, int32_t *status, int32_t *ipm_iters, double *sum_delta)
{
    DeviceGuard guard(c);
    if (!c || c->B < 1)
        return SCPP_E_ARG;
    const size_t B = size_t(c->B);
    CHECK_HIP(hipStreamSynchronize(c->stream));
    if (X)
        CHECK_HIP(hipMemcpy(X, c->X, B * c->K * c->nx * sizeof(double), hipMemcpyDeviceToHost));
    if (U)
        CHECK_HIP(hipMemcpy(U, c->U, B * c->K * c->nu * sizeof(double), hipMemcpyDeviceToHost));
    if (sigma)
        CHECK_HIP(hipMemcpy(sigma, c->sigma, B * sizeof(double), hipMemcpyDeviceToHost));
    if (sc_iters)
        CHECK_HIP(hipMemcpy(sc_iters, c->sc_iters, B * sizeof(int), hipMemcpyDeviceToHost));
    if (nu_norm)
        CHECK_HIP(hipMemcpy(nu_norm, c->norm1_nu, B * sizeof(double), hipMemcpyDeviceToHost));
    if (converged)
        CHECK_HIP(hipMemcpy(converged, c->converged, B * sizeof(int), hipMemcpyDeviceToHost));
    if (status)
        CHECK_HIP(hipMemcpy(status, c->status, B * sizeof(int), hipMemcpyDeviceToHost));
    if (ipm_iters)
        CHECK_HIP(hipMemcpy(ipm_iters, c->ipm_iters, B * sizeof(int), hipMemcpyDeviceToHost));
    if (sum_delta)
        CHECK_HIP(hipMemcpy(sum_delta, c->sum_delta, B * sizeof(double), hipMemcpyDeviceToHost));
    return SCPP_OK;
}

int scpp_hip_download_socp_info(scpp_hip_ctx *c, double *info)
{
    DeviceGuard guard(c);
    if (!c || !info || !c->dbg || c->B < 1)
        return SCPP_E_ARG;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    CHECK_HIP(hipMemcpy(info, c->dbg, size_t(c->B) * 32 * sizeof(double), hipMemcpyDeviceToHost));
    return SCPP_OK;
}

// ---- linear MPC (Rocket2D): MPCAlgorithm.cpp:34-139, MPC_sim.cpp:49-86 ----
int scpp_hip_mpc_setup(scpp_hip_ctx *c, const scpp_mpc_opts *o, const double *flow_par)
{
    DeviceGuard guard(c);
    if (!c || !o || !flow_par)
        return SCPP_E_ARG;
    if (c->model != SCPP_MODEL_ROCKET2D)
        return SCPP_E_UNSUPPORTED;
    if (!c->mpc_host)
        c->mpc_host = new (std::nothrow) mpc::MpcConst;
    if (!c->mpc_host)
        return SCPP_E_HIP;
    c->mpc_ready = false;
    const int rc = mpc::buildMpcConst(*o, flow_par, *c->mpc_host);
    if (rc != SCPP_OK)
        return rc;
    if (!c->mpc_const)
    {
        const size_t B = size_t(c->Bmax);
        int a = 0;
        a |= devAlloc(&c->mpc_const, 1);
        a |= devAlloc(&c->mpc_xf, B * mpc::NX);
        a |= devAlloc(&c->mpc_x0, B * mpc::NX);
        a |= devAlloc(&c->mpc_U, B * mpc::NMAX * mpc::NU);
        a |= devAlloc(&c->mpc_X, B * mpc::KMAX * mpc::NX);
        a |= devAlloc(&c->mpc_cost, B * 2);
        a |= devAlloc(&c->mpc_uheld, B * mpc::NU);
        a |= devAlloc(&c->mpc_t, B);
        a |= devAlloc(&c->mpc_status, B);
        a |= devAlloc(&c->mpc_iters, B);
        a |= devAlloc(&c->mpc_steps, B);
        a |= devAlloc(&c->mpc_failed, B);
        a |= devAlloc(&c->mpc_ipm, B);
        a |= devAlloc(&c->mpc_reached, B);
        if (a)
            return SCPP_E_HIP;
    }
    CHECK_HIP(hipMemcpyAsync(c->mpc_const, c->mpc_host, sizeof(mpc::MpcConst), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->par, flow_par, size_t(c->np) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipStreamSynchronize(c->stream));
    c->par_from_ip = false;
    c->mpc_ready = true;
    c->mpc_B = 0;
    return SCPP_OK;
}

int scpp_hip_mpc_get_model(scpp_hip_ctx *c, double *A, double *B, double *z)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_E_ARG;
    if (!c->mpc_ready)
        return SCPP_E_STATE;
    if (A)
        std::memcpy(A, c->mpc_host->A, sizeof c->mpc_host->A);
    if (B)
        std::memcpy(B, c->mpc_host->B, sizeof c->mpc_host->B);
    if (z)
        std::memcpy(z, c->mpc_host->z, sizeof c->mpc_host->z);
    return SCPP_OK;
}

namespace
{
void launchMpcSolve(scpp_hip_ctx *c, const double *x0, const int *active, int B)
{
    // the shipped horizon (K = 7: 14 variables) carries 14 columns, any other the whole 16-column tile
    if (c->mpc_host->nv == 14)
        hipLaunchKernelGGL((mpc::mpc_solve_kernel<14>), dim3(unsigned(B)), dim3(64), 0, c->stream, (const mpc::MpcConst *)c->mpc_const,
                           x0, (const double *)c->mpc_xf, c->mpc_U, c->mpc_X, c->mpc_cost, c->mpc_status, c->mpc_iters, active, B);
    else
        hipLaunchKernelGGL((mpc::mpc_solve_kernel<16>), dim3(unsigned(B)), dim3(64), 0, c->stream, (const mpc::MpcConst *)c->mpc_const,
                           x0, (const double *)c->mpc_xf, c->mpc_U, c->mpc_X, c->mpc_cost, c->mpc_status, c->mpc_iters, active, B);
}
} // namespace

int scpp_hip_mpc_solve(scpp_hip_ctx *c, const double *x_init, const double *x_final, int B, int *n_solved)
{
    DeviceGuard guard(c);
    if (!c || !x_init || !x_final || B < 1 || B > c->Bmax)
        return SCPP_E_ARG;
    if (!c->mpc_ready)
        return SCPP_E_STATE;
    CHECK_HIP(hipMemcpyAsync(c->mpc_x0, x_init, size_t(B) * mpc::NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->mpc_xf, x_final, size_t(B) * mpc::NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (c->mpc_B != B)
    {
        // a failed solve leaves X / U untouched: define them for the first use
        CHECK_HIP(hipMemsetAsync(c->mpc_U, 0, size_t(B) * mpc::NMAX * mpc::NU * sizeof(double), c->stream));
        CHECK_HIP(hipMemsetAsync(c->mpc_X, 0, size_t(B) * mpc::KMAX * mpc::NX * sizeof(double), c->stream));
        CHECK_HIP(hipMemsetAsync(c->mpc_cost, 0, size_t(B) * 2 * sizeof(double), c->stream));
        c->mpc_B = B;
    }
    const bool timed = spanBegin(c, 1, B, c->stream);
    launchMpcSolve(c, c->mpc_x0, nullptr, B);
    spanEnd(c, timed, c->stream);
    CHECK_HIP(hipGetLastError());
    if (n_solved)
    {
        const size_t nb = size_t(B);
        std::vector<int> st(nb);
        CHECK_HIP(hipMemcpyAsync(st.data(), c->mpc_status, size_t(B) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        CHECK_HIP(hipStreamSynchronize(c->stream));
        int n = 0;
        for (int v : st)
            n += v >= 0;
        *n_solved = n;
    }
    return SCPP_OK;
}

int scpp_hip_mpc_download(scpp_hip_ctx *c, double *X, double *U, double *cost, int32_t *status, int32_t *iters)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_E_ARG;
    if (!c->mpc_ready || c->mpc_B < 1)
        return SCPP_E_STATE;
    const int B = c->mpc_B, K = c->mpc_host->K, N = K - 1;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    if (X)
        CHECK_HIP(hipMemcpy2D(X, size_t(K) * mpc::NX * sizeof(double), c->mpc_X, size_t(mpc::KMAX) * mpc::NX * sizeof(double),
                              size_t(K) * mpc::NX * sizeof(double), size_t(B), hipMemcpyDeviceToHost));
    if (U)
        CHECK_HIP(hipMemcpy2D(U, size_t(N) * mpc::NU * sizeof(double), c->mpc_U, size_t(mpc::NMAX) * mpc::NU * sizeof(double),
                              size_t(N) * mpc::NU * sizeof(double), size_t(B), hipMemcpyDeviceToHost));
    if (cost)
        CHECK_HIP(hipMemcpy(cost, c->mpc_cost, size_t(B) * 2 * sizeof(double), hipMemcpyDeviceToHost));
    if (status)
        CHECK_HIP(hipMemcpy(status, c->mpc_status, size_t(B) * sizeof(int), hipMemcpyDeviceToHost));
    if (iters)
        CHECK_HIP(hipMemcpy(iters, c->mpc_iters, size_t(B) * sizeof(int), hipMemcpyDeviceToHost));
    return SCPP_OK;
}

int scpp_hip_mpc_sim(scpp_hip_ctx *c, const double *x_start, const double *x_final, int B, double time_step, double sim_time,
                     double stop_tol, int max_steps, int *n_reached)
{
    DeviceGuard guard(c);
    if (!c || !x_start || !x_final || B < 1 || B > c->Bmax || !(time_step > 0.) || !(sim_time > 0.))
        return SCPP_E_ARG;
    if (!c->mpc_ready)
        return SCPP_E_STATE;
    const size_t nb = size_t(B);
    CHECK_HIP(hipMemcpyAsync(c->sim_x, x_start, nb * mpc::NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHECK_HIP(hipMemcpyAsync(c->mpc_xf, x_final, nb * mpc::NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    {
        std::vector<double> dtv(nb, time_step);
        std::vector<int> one(nb, 1);
        CHECK_HIP(hipMemcpyAsync(c->sim_dt, dtv.data(), nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
        CHECK_HIP(hipMemcpyAsync(c->active, one.data(), nb * sizeof(int), hipMemcpyHostToDevice, c->stream));
        CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    CHECK_HIP(hipMemsetAsync(c->mpc_uheld, 0, nb * mpc::NU * sizeof(double), c->stream));
    CHECK_HIP(hipMemsetAsync(c->mpc_t, 0, nb * sizeof(double), c->stream));
    CHECK_HIP(hipMemsetAsync(c->mpc_U, 0, nb * mpc::NMAX * mpc::NU * sizeof(double), c->stream));
    CHECK_HIP(hipMemsetAsync(c->mpc_X, 0, nb * mpc::KMAX * mpc::NX * sizeof(double), c->stream));
    CHECK_HIP(hipMemsetAsync(c->mpc_cost, 0, nb * 2 * sizeof(double), c->stream));
    for (int *p : {c->mpc_steps, c->mpc_failed, c->mpc_ipm, c->mpc_reached, c->mpc_status, c->mpc_iters})
        CHECK_HIP(hipMemsetAsync(p, 0, nb * sizeof(int), c->stream));
    c->mpc_B = B;
    const unsigned grid = unsigned((B + 63) / 64);
    const long cap = max_steps > 0 ? long(max_steps) : long(std::ceil(sim_time / time_step)) + 2;
    for (long step = 0; step < cap; step++)
    {
        launchMpcSolve(c, c->sim_x, c->active, B);
        hipLaunchKernelGGL((simulate_kernel<Rocket2dModel>), dim3(grid), dim3(64), 0, c->stream, B, (const double *)c->par, 0,
                           (const double *)c->sim_dt, (const double *)c->mpc_uheld, (const double *)c->mpc_uheld, c->sim_x,
                           (const int *)c->active);
        CHECK_HIP(hipMemsetAsync(c->counter, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(mpc::mpc_sim_advance_kernel, dim3(grid), dim3(64), 0, c->stream, B, (const double *)c->mpc_U,
                           (const int *)c->mpc_status, (const int *)c->mpc_iters, (const double *)c->sim_x,
                           (const double *)c->mpc_xf, c->mpc_uheld, c->active, c->mpc_steps, c->mpc_failed, c->mpc_ipm,
                           c->mpc_reached, c->mpc_t, time_step, sim_time, stop_tol, c->counter);
        if ((step & 15) == 15 || step + 1 == cap)
        {
            int n = 0;
            CHECK_HIP(hipMemcpyAsync(&n, c->counter, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            CHECK_HIP(hipStreamSynchronize(c->stream));
            if (n == 0)
                break;
        }
    }
    CHECK_HIP(hipGetLastError());
    if (n_reached)
    {
        std::vector<int> r(nb);
        CHECK_HIP(hipMemcpyAsync(r.data(), c->mpc_reached, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        CHECK_HIP(hipStreamSynchronize(c->stream));
        int n = 0;
        for (int v : r)
            n += v;
        *n_reached = n;
    }
    return SCPP_OK;
}

int scpp_hip_mpc_sim_download(scpp_hip_ctx *c, double *x, double *u, double *t, int32_t *steps, int32_t *failed_solves,
                              int32_t *ipm_iters, int32_t *reached)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_E_ARG;
    if (!c->mpc_ready || c->mpc_B < 1)
        return SCPP_E_STATE;
    const size_t nb = size_t(c->mpc_B);
    CHECK_HIP(hipStreamSynchronize(c->stream));
    if (x)
        CHECK_HIP(hipMemcpy(x, c->sim_x, nb * mpc::NX * sizeof(double), hipMemcpyDeviceToHost));
    if (u)
        CHECK_HIP(hipMemcpy(u, c->mpc_uheld, nb * mpc::NU * sizeof(double), hipMemcpyDeviceToHost));
    if (t)
        CHECK_HIP(hipMemcpy(t, c->mpc_t, nb * sizeof(double), hipMemcpyDeviceToHost));
    if (steps)
        CHECK_HIP(hipMemcpy(steps, c->mpc_steps, nb * sizeof(int), hipMemcpyDeviceToHost));
    if (failed_solves)
        CHECK_HIP(hipMemcpy(failed_solves, c->mpc_failed, nb * sizeof(int), hipMemcpyDeviceToHost));
    if (ipm_iters)
        CHECK_HIP(hipMemcpy(ipm_iters, c->mpc_ipm, nb * sizeof(int), hipMemcpyDeviceToHost));
    if (reached)
        CHECK_HIP(hipMemcpy(reached, c->mpc_reached, nb * sizeof(int), hipMemcpyDeviceToHost));
    return SCPP_OK;
}


int scpp_hip_get_timing(scpp_hip_ctx *c, scpp_timing *out, int reset)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_E_ARG;
    collectTiming(c);
    if (out)
        *out = c->timing;
    if (reset)
    {
        c->timing = scpp_timing{};
        // new time origin for the span intervals (all streams of the context are idle here: collectTiming waited for them)
        if (!c->ev_base && hipEventCreate(&c->ev_base) != hipSuccess)
            c->ev_base = nullptr;
        if (c->ev_base)
        {
            (void)hipEventRecord(c->ev_base, c->stream);
            (void)hipEventSynchronize(c->ev_base);
        }
    }
    return SCPP_OK;
}

int scpp_hip_device_ptrs(scpp_hip_ctx *c, void **X, void **U, void **sigma)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_E_ARG;
    if (X)
        *X = c->X;
    if (U)
        *U = c->U;
    if (sigma)
        *sigma = c->sigma;
    return SCPP_OK;
}

int scpp_hip_synchronize(scpp_hip_ctx *c)
{
    DeviceGuard guard(c);
    if (!c)
        return SCPP_E_ARG;
    CHECK_HIP(hipStreamSynchronize(c->stream));
    return SCPP_OK;
}

} // extern "C"
