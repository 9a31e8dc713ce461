// ORACLE (test infrastructure, NOT product code): C entry points for ctypes.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <cstring>
#include <memory>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "discretization.hpp"
#include "models.hpp"
#include "sc.hpp"
#include "sc_sim.hpp"
#include "scvx.hpp"
#include "mpc.hpp"
#include "socp.hpp"

using namespace oracle;

namespace
{

template <class M>
int flowImpl(const double *x, const double *u, const double *par, double *f, double *A, double *B)
{
    M m;
    for (int i = 0; i < M::NP; i++)
        m.par[i] = par[i];
    m.computef(x, u, f);
    m.computeJacobians(x, u, A, B);
    return 0;
}

template <class M>
int discretizeImpl(int K, int foh, int vt, const double *par, const double *X, const double *U, double t, double *A,
                   double *B, double *C, double *s, double *z)
{
    M m;
    for (int i = 0; i < M::NP; i++)
        m.par[i] = par[i];
    TrajectoryData td;
    td.initialize(M::NX, M::NU, K, foh != 0);
    std::memcpy(td.X.data(), X, td.X.size() * sizeof(double));
    std::memcpy(td.U.data(), U, td.U.size() * sizeof(double));
    td.t = t;
    DiscretizationData dd;
    dd.initialize(M::NX, M::NU, K, foh != 0, vt != 0);
    multipleShooting(m, td, dd);
    std::memcpy(A, dd.A.data(), dd.A.size() * sizeof(double));
    std::memcpy(B, dd.B.data(), dd.B.size() * sizeof(double));
    if (foh)
        std::memcpy(C, dd.C.data(), dd.C.size() * sizeof(double));
    if (vt)
        std::memcpy(s, dd.s.data(), dd.s.size() * sizeof(double));
    std::memcpy(z, dd.z.data(), dd.z.size() * sizeof(double));
    return 0;
}

struct SCHandleBase
{
    virtual ~SCHandleBase() {}
    virtual int solve(int warm) = 0;
    virtual SCAlgorithm<RocketQuat> *rq() { return nullptr; }
    virtual SCAlgorithm<Rocket2d> *r2() { return nullptr; }
    virtual SCAlgorithm<Lander3dof> *l3() { return nullptr; }
};

template <class M>
struct SCHandle : SCHandleBase
{
    M model;
    std::unique_ptr<SCAlgorithm<M>> alg;
    SCHandle(const std::string &root, int K)
    {
        const std::string folder = root + "/" + M::modelName();
        model.loadParameters(folder);
        alg.reset(new SCAlgorithm<M>(&model, folder, K));
        alg->initialize();
    }
    int solve(int warm) override
    {
        alg->solve(warm != 0);
        return alg->solver_failed ? -1 : 0;
    }
    SCAlgorithm<RocketQuat> *rq() override { return pick<RocketQuat>(); }
    SCAlgorithm<Rocket2d> *r2() override { return pick<Rocket2d>(); }
    SCAlgorithm<Lander3dof> *l3() override { return pick<Lander3dof>(); }
    template <class T>
    SCAlgorithm<T> *pick()
    {
        if constexpr (std::is_same<T, M>::value)
            return alg.get();
        else
            return nullptr;
    }
};

template <class F>
auto withAlg(void *h, F f)
{
    SCHandleBase *b = static_cast<SCHandleBase *>(h);
    if (b->rq())
        return f(*b->rq());
    if (b->l3())
        return f(*b->l3());
    return f(*b->r2());
}

} // namespace

extern "C"
{

int oracle_model_dims(int model, int *dims)
{
    if (model == 0)
    {
        dims[0] = RocketQuat::NX;
        dims[1] = RocketQuat::NU;
        dims[2] = RocketQuat::NP;
    }
    else if (model == 2)
    {
        dims[0] = Lander3dof::NX;
        dims[1] = Lander3dof::NU;
        dims[2] = Lander3dof::NP;
    }
    else
    {
        dims[0] = Rocket2d::NX;
        dims[1] = Rocket2d::NU;
        dims[2] = Rocket2d::NP;
    }
    return 0;
}

int oracle_flow(int model, const double *x, const double *u, const double *par, double *f, double *A, double *B)
{
    return model == 0 ? flowImpl<RocketQuat>(x, u, par, f, A, B) : model == 2 ? flowImpl<Lander3dof>(x, u, par, f, A, B) : flowImpl<Rocket2d>(x, u, par, f, A, B);
}

int oracle_rkf78_tableau(double *c, double *a, double *b)
{
    const RKF78Tableau &T = rkf78();
    for (int i = 0; i < 13; i++)
    {
        c[i] = T.c[i];
        b[i] = T.b[i];
        for (int j = 0; j < 13; j++)
            a[i * 13 + j] = T.a[i][j];
    }
    return 0;
}

// y'' = -omega^2 y, (y, y') from (1, 0): convergence-order probe for the stepper
int oracle_rkf78_harmonic(double omega, double dt, int N, double *y)
{
    std::vector<double> v{1., 0.};
    auto ode = [&](const std::vector<double> &s, std::vector<double> &d, double) {
        d[0] = s[1];
        d[1] = -omega * omega * s[0];
    };
    integrateRKF78(ode, v, dt, N);
    y[0] = v[0];
    y[1] = v[1];
    return 0;
}

int oracle_discretize(int model, int K, int foh, int vt, const double *par, const double *X, const double *U, double t,
                      double *A, double *B, double *C, double *s, double *z)
{
    return model == 0   ? discretizeImpl<RocketQuat>(K, foh, vt, par, X, U, t, A, B, C, s, z)
           : model == 2 ? discretizeImpl<Lander3dof>(K, foh, vt, par, X, U, t, A, B, C, s, z)
                        : discretizeImpl<Rocket2d>(K, foh, vt, par, X, U, t, A, B, C, s, z);
}

int oracle_simulate(int model, const double *par, double dt, const double *u0, const double *u1, double *x)
{
    if (model == 0)
    {
        RocketQuat m;
        for (int i = 0; i < RocketQuat::NP; i++)
            m.par[i] = par[i];
        simulate(m, dt, u0, u1, x);
    }
    else if (model == 2)
    {
        Lander3dof m;
        for (int i = 0; i < Lander3dof::NP; i++)
            m.par[i] = par[i];
        simulate(m, dt, u0, u1, x);
    }
    else
    {
        Rocket2d m;
        for (int i = 0; i < Rocket2d::NP; i++)
            m.par[i] = par[i];
        simulate(m, dt, u0, u1, x);
    }
    return 0;
}

// ---- generic SOCP (dense input, small tests) ----
// info: [exitflag, iter, pcost, dcost, pres, dres, gap]
int oracle_socp_solve(int n, int p, int l, int ncones, const int *q, const double *c, const double *A, const double *b,
                      const double *G, const double *h, double *x, double *y, double *z, double *s, double *info)
{
    Socp prob;
    prob.addVars(n, 0);
    for (int j = 0; j < n; j++)
        prob.c[j] = c[j];
    for (int r = 0; r < p; r++)
    {
        SocpRow row;
        for (int j = 0; j < n; j++)
            if (A[r * n + j] != 0.)
                row.t.push_back({j, A[r * n + j]});
        row.rhs = b[r];
        row.key = 2;
        prob.eq.push_back(row);
    }
    int m = l;
    for (int k = 0; k < ncones; k++)
        m += q[k];
    auto mkrow = [&](int r) {
        SocpRow row;
        for (int j = 0; j < n; j++)
            if (G[r * n + j] != 0.)
                row.t.push_back({j, G[r * n + j]});
        row.rhs = h[r];
        row.key = 1;
        return row;
    };
    int r = 0;
    for (; r < l; r++)
        prob.lp.push_back(mkrow(r));
    for (int k = 0; k < ncones; k++)
    {
        std::vector<SocpRow> cone;
        for (int i = 0; i < q[k]; i++, r++)
            cone.push_back(mkrow(r));
        prob.soc.push_back(cone);
    }
    // order: cone rows first would hit zero pivots for variables outside all cones; use
    // variables (delta) -> cone rows -> equality rows, which is always quasi-definite
    for (int j = 0; j < n; j++)
        prob.var_key[j] = 0;
    SocpSolver solver(prob);
    SocpResult R = solver.solve();
    for (int j = 0; j < n; j++)
        x[j] = R.x[j];
    for (int j = 0; j < p; j++)
        y[j] = R.y[j];
    for (int j = 0; j < m; j++)
    {
        z[j] = R.z[j];
        s[j] = R.s[j];
    }
    info[0] = R.exitflag;
    info[1] = R.iter;
    info[2] = R.pcost;
    info[3] = R.dcost;
    info[4] = R.pres;
    info[5] = R.dres;
    info[6] = R.gap;
    return 0;
}

// ---- SC driver ----
void *oracle_sc_create(int model, const char *config_root, int K_override)
{
    try
    {
        if (model == 0)
            return new SCHandle<RocketQuat>(config_root, K_override);
        if (model == 2)
            return new SCHandle<Lander3dof>(config_root, K_override);
        return new SCHandle<Rocket2d>(config_root, K_override);
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_sc_create: %s\n", e.what());
        return nullptr;
    }
}
void oracle_sc_destroy(void *h) { delete static_cast<SCHandleBase *>(h); }

int oracle_sc_set_tolerances(void *h, double feastol, double abstol, double reltol, int maxit)
{
    return withAlg(h, [&](auto &a) {
        a.socp_settings.feastol = feastol;
        a.socp_settings.abstol = abstol;
        a.socp_settings.reltol = reltol;
        a.socp_settings.maxit = maxit;
        return 0;
    });
}
// candidate point in the literal SC sub-problem linearised at (Xbar, Ubar, tbar), trajectories DIMENSIONAL (sc.hpp: checkPoint)
// out[10]: eq_violation, min_lp_slack, min_cone_slack, cost, norm1_nu, lit_cost, lit_sigma, lit_exitflag, lit_iters, sum_delta
int oracle_sc_check_point(void *h, const double *Xbar, const double *Ubar, double tbar, double w_trx, const double *Xc, const double *Uc,
                          double tc, int solve_literal, double *out, double *Xlit, double *Ulit)
{
    try
    {
        return withAlg(h, [&](auto &a) {
            const auto r = a.checkPoint(Xbar, Ubar, tbar, w_trx, Xc, Uc, tc, solve_literal != 0);
            out[0] = r.eq_violation;
            out[1] = r.min_lp_slack;
            out[2] = r.min_cone_slack;
            out[3] = r.cost;
            out[4] = r.norm1_nu;
            out[5] = r.lit_cost;
            out[6] = r.lit_sigma;
            out[7] = r.lit_exitflag;
            out[8] = r.lit_iters;
            out[9] = r.sum_delta;
            if (Xlit && !r.Xlit.empty())
                std::memcpy(Xlit, r.Xlit.data(), r.Xlit.size() * sizeof(double));
            if (Ulit && !r.Ulit.empty())
                std::memcpy(Ulit, r.Ulit.data(), r.Ulit.size() * sizeof(double));
            return 0;
        });
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_sc_check_point: %s\n", e.what());
        return -2;
    }
}
// primal and dual step lengths of their own in the structured twin (default on = the device's IPM_SPLIT_STEPS); 0: ECOS's common one
int oracle_sc_set_twin_split_steps(void *h, int on)
{
    return withAlg(h, [&](auto &a) {
        a.structured_settings.split_steps = on != 0;
        return 0;
    });
}
int oracle_sc_set_solver(void *h, int kind)
{
    return withAlg(h, [&](auto &a) {
        a.solver_kind = kind;
        return 0;
    });
}
int oracle_sc_verbose(void *h, int v)
{
    return withAlg(h, [&](auto &a) {
        a.socp_settings.verbose = v != 0;
        a.structured_settings.verbose = v != 0;
        return 0;
    });
}

// RocketQuat only: perturb x_init per SURVEY §8(d)
int oracle_sc_randomize(void *h, unsigned long long seed, unsigned long long instance)
{
    SCHandleBase *b = static_cast<SCHandleBase *>(h);
    if (!b->rq())
        return -1;
    b->rq()->model->p.randomizeInitialState(seed, instance);
    return 0;
}
int oracle_sc_get_x_init(void *h, double *x)
{
    return withAlg(h, [&](auto &a) {
        for (int i = 0; i < a.td.nx; i++)
            x[i] = a.model->p.x_init[i];
        return 0;
    });
}
int oracle_sc_set_x_init(void *h, const double *x)
{
    return withAlg(h, [&](auto &a) {
        for (int i = 0; i < a.td.nx; i++)
            a.model->p.x_init[i] = x[i];
        return 0;
    });
}
int oracle_sc_get_x_final(void *h, double *x)
{
    return withAlg(h, [&](auto &a) {
        for (int i = 0; i < a.td.nx; i++)
            x[i] = a.model->p.x_final[i];
        return 0;
    });
}
int oracle_sc_solve(void *h, int warm_start)
{
    try
    {
        return static_cast<SCHandleBase *>(h)->solve(warm_start);
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_sc_solve: %s\n", e.what());
        return -2;
    }
}
// SC_sim.cpp:19-104 closed loop from the handle's current x_init.  Outputs (caller-allocated for max_steps):
// X_sim [max_steps][nx], U_sim [max_steps][nu], t_plan [max_steps], sc_iters [max_steps]; meta = {steps, reached_end, solver_failed}
int oracle_sc_sim(void *h, double time_step, int max_steps, double *X_sim, double *U_sim, double *t_plan, int *sc_iters, int *meta)
{
    try
    {
        return withAlg(h, [&](auto &a) {
            auto r = scSim(a, time_step, max_steps);
            std::memcpy(X_sim, r.X_sim.data(), r.X_sim.size() * sizeof(double));
            std::memcpy(U_sim, r.U_sim.data(), r.U_sim.size() * sizeof(double));
            std::memcpy(t_plan, r.t_plan.data(), r.t_plan.size() * sizeof(double));
            for (size_t i = 0; i < r.sc_iterations.size(); i++)
                sc_iters[i] = r.sc_iterations[i];
            meta[0] = r.steps;
            meta[1] = r.reached_end;
            meta[2] = r.solver_failed;
            return 0;
        });
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_sc_sim: %s\n", e.what());
        return -2;
    }
}
// meta: [K, nU, nx, nu, iterations, converged, n_all_td, n, p, l, ncones, m]
int oracle_sc_meta(void *h, int *meta)
{
    return withAlg(h, [&](auto &a) {
        meta[0] = a.td.K;
        meta[1] = a.td.nU;
        meta[2] = a.td.nx;
        meta[3] = a.td.nu;
        meta[4] = a.iterations;
        meta[5] = a.converged ? 1 : 0;
        meta[6] = int(a.all_td.size());
        for (int i = 0; i < 5; i++)
            meta[7 + i] = a.last_dims[i];
        return 0;
    });
}
int oracle_sc_get_solution(void *h, double *X, double *U, double *t)
{
    return withAlg(h, [&](auto &a) {
        std::memcpy(X, a.td.X.data(), a.td.X.size() * sizeof(double));
        std::memcpy(U, a.td.U.data(), a.td.U.size() * sizeof(double));
        *t = a.td.t;
        return 0;
    });
}
// all iterates, NON-dimensional as stored (index 0 = initial guess of the first solve)
int oracle_sc_get_iterate(void *h, int idx, double *X, double *U, double *t)
{
    return withAlg(h, [&](auto &a) {
        if (idx < 0 || idx >= int(a.all_td.size()))
            return -1;
        const TrajectoryData &td = a.all_td[idx];
        std::memcpy(X, td.X.data(), td.X.size() * sizeof(double));
        std::memcpy(U, td.U.data(), td.U.size() * sizeof(double));
        *t = td.t;
        return 0;
    });
}
// per-iteration info rows: [norm1_nu, sum_delta, delta_sigma, sigma, ipm_iters, exitflag, pres, dres, gap]
int oracle_sc_get_info(void *h, double *rows, int max_rows)
{
    return withAlg(h, [&](auto &a) {
        int n = std::min<int>(max_rows, int(a.info.size()));
        for (int i = 0; i < n; i++)
        {
            const SCIterationInfo &f = a.info[i];
            double *r = rows + i * 9;
            r[0] = f.norm1_nu;
            r[1] = f.sum_delta;
            r[2] = f.delta_sigma;
            r[3] = f.sigma;
            r[4] = f.ipm_iters;
            r[5] = f.exitflag;
            r[6] = f.pres;
            r[7] = f.dres;
            r[8] = f.gap;
        }
        return n;
    });
}
// last sub-problem primal vector and variable offsets [X,U,nu,nu_bound,norm1_nu,delta,sigma,delta_sigma]
int oracle_sc_get_last_socp_x(void *h, double *x, int *offsets)
{
    return withAlg(h, [&](auto &a) {
        for (size_t i = 0; i < a.last_result.x.size(); i++)
            x[i] = a.last_result.x[i];
        const SCVarIndex &ix = a.last_ix;
        const int o[8] = {ix.X, ix.U, ix.nu, ix.nu_bound, ix.norm1_nu, ix.delta, ix.sigma, ix.delta_sigma};
        for (int i = 0; i < 8; i++)
            offsets[i] = o[i];
        return int(a.last_result.x.size());
    });
}
// scales [m_scale, r_scale] and current weight_trust_region_trajectory
int oracle_sc_get_scales(void *h, double *out)
{
    return withAlg(h, [&](auto &a) {
        out[0] = a.model->p.m_scale;
        out[1] = a.model->p.r_scale;
        out[2] = a.weight_trust_region_trajectory;
        return 0;
    });
}

// ---- batched RocketQuat SC_oneshot over randomised instances (cpu_baseline leg) ----
// outputs per instance: X[K][14], U[K][4] (dimensional), t, iters, converged, final norm1_nu, total ipm iters
int oracle_sc_batch(const char *config_root, int K, unsigned long long seed, long first, long count, int nthreads,
                    int solver_kind, double *X, double *U, double *t, int *iters, int *conv, double *nu, int *ipm_iters)
{
    if (nthreads < 1)
        nthreads = 1;
    std::vector<std::thread> pool;
    std::vector<int> rc(nthreads, 0);
    for (int th = 0; th < nthreads; th++)
    {
        pool.emplace_back([&, th]() {
            for (long i = th; i < count; i += nthreads)
            {
                try
                {
                    SCHandle<RocketQuat> hnd(config_root, K);
                    hnd.model.p.randomizeInitialState(seed, (unsigned long long)(first + i));
                    hnd.alg->solver_kind = solver_kind;
                    hnd.solve(0);
                    auto &a = *hnd.alg;
                    std::memcpy(X + size_t(i) * K * 14, a.td.X.data(), sizeof(double) * K * 14);
                    std::memcpy(U + size_t(i) * K * 4, a.td.U.data(), sizeof(double) * K * 4);
                    t[i] = a.td.t;
                    iters[i] = a.iterations;
                    conv[i] = a.converged ? 1 : 0;
                    nu[i] = a.info.empty() ? 0. : a.info.back().norm1_nu;
                    int tot = 0;
                    for (auto &f : a.info)
                        tot += f.ipm_iters;
                    ipm_iters[i] = tot;
                }
                catch (const std::exception &e)
                {
                    rc[th] = -1;
                }
            }
        });
    }
    for (auto &th : pool)
        th.join();
    for (int r : rc)
        if (r)
            return r;
    return 0;
}

} // extern "C"


// ---- SCvx (any model; the structured twin exists for RocketQuat only) ----
namespace
{
struct SCvxHandleBase
{
    virtual ~SCvxHandleBase() {}
    virtual SCvxAlgorithm<RocketQuat> *rq() { return nullptr; }
    virtual SCvxAlgorithm<Rocket2d> *r2() { return nullptr; }
    virtual SCvxAlgorithm<Lander3dof> *l3() { return nullptr; }
};
template <class M>
struct SCvxHandleT : SCvxHandleBase
{
    M model;
    std::unique_ptr<SCvxAlgorithm<M>> alg;
    SCvxHandleT(const std::string &root, int K)
    {
        const std::string folder = root + "/" + M::modelName();
        model.loadParameters(folder);
        alg.reset(new SCvxAlgorithm<M>(&model, folder, K));
        alg->initialize();
    }
    SCvxAlgorithm<RocketQuat> *rq() override { return pick<RocketQuat>(); }
    SCvxAlgorithm<Rocket2d> *r2() override { return pick<Rocket2d>(); }
    SCvxAlgorithm<Lander3dof> *l3() override { return pick<Lander3dof>(); }
    template <class T>
    SCvxAlgorithm<T> *pick()
    {
        if constexpr (std::is_same<T, M>::value)
            return alg.get();
        else
            return nullptr;
    }
};
template <class F>
auto withScvx(void *h, F f)
{
    SCvxHandleBase *b = static_cast<SCvxHandleBase *>(h);
    if (b->rq())
        return f(*b->rq());
    if (b->l3())
        return f(*b->l3());
    return f(*b->r2());
}
} // namespace

extern "C"
{
void *oracle_scvx_create_model(int model, const char *config_root, int K_override)
{
    try
    {
        if (model == 0)
            return static_cast<SCvxHandleBase *>(new SCvxHandleT<RocketQuat>(config_root, K_override));
        if (model == 2)
            return static_cast<SCvxHandleBase *>(new SCvxHandleT<Lander3dof>(config_root, K_override));
        return static_cast<SCvxHandleBase *>(new SCvxHandleT<Rocket2d>(config_root, K_override));
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_scvx_create: %s\n", e.what());
        return nullptr;
    }
}
void *oracle_scvx_create(const char *config_root, int K_override) { return oracle_scvx_create_model(0, config_root, K_override); }
void oracle_scvx_destroy(void *h) { delete static_cast<SCvxHandleBase *>(h); }
int oracle_scvx_set_solver(void *h, int kind)
{
    return withScvx(h, [&](auto &a) {
        a.solver_kind = kind;
        return 0;
    });
}
int oracle_scvx_set_tolerances(void *h, double feastol, double abstol, double reltol, int maxit)
{
    return withScvx(h, [&](auto &a) {
        a.socp_settings.feastol = feastol;
        a.socp_settings.abstol = abstol;
        a.socp_settings.reltol = reltol;
        a.socp_settings.maxit = maxit;
        return 0;
    });
}
// tolerances of the structured twin (defaults: feastol 1e-8, abstol / reltol 1e-7, maxit 60 = the device's)
int oracle_scvx_set_twin_tolerances(void *h, double feastol, double abstol, double reltol, int maxit)
{
    return withScvx(h, [&](auto &a) {
        a.structured_settings.feastol = feastol;
        a.structured_settings.abstol = abstol;
        a.structured_settings.reltol = reltol;
        a.structured_settings.maxit = maxit;
        return 0;
    });
}
// primal and dual step lengths of their own in the structured twin (default: on, like the device kernel's IPM_SPLIT_STEPS); 0: ECOS's common one
int oracle_scvx_set_twin_split_steps(void *h, int on)
{
    return withScvx(h, [&](auto &a) {
        a.structured_settings.split_steps = on != 0;
        return 0;
    });
}
int oracle_scvx_verbose(void *h, int v)
{
    return withScvx(h, [&](auto &a) {
        a.socp_settings.verbose = v != 0;
        a.structured_settings.verbose = v != 0;
        return 0;
    });
}
int oracle_scvx_set_reg(void *h, double dx, double deq, double dcone)
{
    return withScvx(h, [&](auto &a) {
        a.socp_settings.delta_x = dx;
        a.socp_settings.delta_eq = deq;
        a.socp_settings.delta_cone = dcone;
        return 0;
    });
}
int oracle_scvx_randomize(void *h, unsigned long long seed, unsigned long long instance)
{
    SCvxHandleBase *b = static_cast<SCvxHandleBase *>(h);
    if (!b->rq())
        return -1; // the randomisation recipe is RocketQuat's (rocketQuat.cpp:203-227)
    b->rq()->model->p.randomizeInitialState(seed, instance);
    return 0;
}
int oracle_scvx_set_x_init(void *h, const double *x)
{
    return withScvx(h, [&](auto &a) {
        for (size_t i = 0; i < sizeof(a.model->p.x_init) / sizeof(double); i++)
            a.model->p.x_init[i] = x[i];
        return 0;
    });
}
int oracle_scvx_set_max_iterations(void *h, int n)
{
    return withScvx(h, [&](auto &a) {
        a.max_iterations_override = size_t(n);
        a.max_iterations = size_t(n);
        return 0;
    });
}
// test support (scvx.hpp: solve_cap): retire an instance like the device engine does; 0 = the reference's behaviour
int oracle_scvx_set_solve_cap(void *h, int cap)
{
    return withScvx(h, [&](auto &a) {
        a.solve_cap = size_t(cap > 0 ? cap : 0);
        return 0;
    });
}
int oracle_scvx_retired(void *h)
{
    return withScvx(h, [&](auto &a) { return a.retired ? 1 : 0; });
}
int oracle_scvx_solve(void *h, int warm_start)
{
    try
    {
        return withScvx(h, [&](auto &a) {
            a.solve(warm_start != 0);
            return a.solver_failed ? -1 : 0;
        });
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_scvx_solve: %s\n", e.what());
        return -2;
    }
}
// meta: [K, nU, iterations, converged, n_all_td, n_info, solves, n, p, l, ncones, m]
int oracle_scvx_meta(void *h, int *meta)
{
    return withScvx(h, [&](auto &a) {
        meta[0] = a.td.K;
        meta[1] = a.td.nU;
        meta[2] = a.iterations;
        meta[3] = a.converged;
        meta[4] = int(a.all_td.size());
        meta[5] = int(a.info.size());
        meta[6] = a.solves;
        for (int i = 0; i < 5; i++)
            meta[7 + i] = a.last_dims[i];
        return 0;
    });
}
// The CPU baseline of bench.py: `n` randomised RocketQuat SCvx runs (instances first .. first + n - 1 of `seed`) on `threads` native
// threads pulling instance numbers from one counter -- the same calls the Python wrapper makes per instance (create, randomize,
// set_solver, solve), without the interpreter in between.  counts: [converged, failures, sum of SCvx iterations, sum of solves]
int oracle_scvx_run_batch(const char *config_root, int K, unsigned long long seed, unsigned long long first, int n, int solver,
                          int threads, long long *counts)
{
    if (n < 1 || threads < 1 || !counts)
        return -1;
    std::atomic<int> next{0};
    std::atomic<long long> conv{0}, fail{0}, iters{0}, solves{0};
    auto worker = [&]() {
        for (;;)
        {
            const int i = next.fetch_add(1);
            if (i >= n)
                return;
            void *h = oracle_scvx_create_model(0, config_root, K);
            if (!h)
            {
                fail++;
                continue;
            }
            oracle_scvx_randomize(h, seed, first + (unsigned long long)i);
            oracle_scvx_set_solver(h, solver);
            const int rc = oracle_scvx_solve(h, 0);
            int meta[12];
            oracle_scvx_meta(h, meta);
            if (rc != 0)
                fail++;
            else
                conv += meta[3];
            iters += meta[2];
            solves += meta[6];
            oracle_scvx_destroy(h);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back(worker);
    for (auto &t : pool)
        t.join();
    counts[0] = conv;
    counts[1] = fail;
    counts[2] = iters;
    counts[3] = solves;
    return 0;
}
int oracle_scvx_get_iterate(void *h, int idx, double *X, double *U, double *t)
{
    return withScvx(h, [&](auto &a) {
        const TrajectoryData *td = &a.td;
        if (idx >= 0)
        {
            if (idx >= int(a.all_td.size()))
                return -1;
            td = &a.all_td[size_t(idx)];
        }
        std::memcpy(X, td->X.data(), td->X.size() * sizeof(double));
        std::memcpy(U, td->U.data(), td->U.size() * sizeof(double));
        *t = td->t;
        return 0;
    });
}
// candidate point in the literal sub-problem linearised at (Xbar, Ubar), all trajectories DIMENSIONAL (scvx.hpp: checkPoint).
// out[10]: eq_violation, min_lp_slack, min_cone_slack, cost, norm1_nu, lit_cost, lit_norm1_nu, lit_exitflag, lit_iters, n
int oracle_scvx_check_point(void *h, const double *Xbar, const double *Ubar, double radius, const double *Xc, const double *Uc,
                            int solve_literal, double *out, double *Xlit, double *Ulit, int bar_nondim, int cand_nondim)
{
    try
    {
        return withScvx(h, [&](auto &a) {
                    const auto r = a.checkPoint(Xbar, Ubar, radius, Xc, Uc, solve_literal != 0, bar_nondim != 0, cand_nondim != 0);
            out[0] = r.eq_violation;
            out[1] = r.min_lp_slack;
            out[2] = r.min_cone_slack;
            out[3] = r.cost;
            out[4] = r.norm1_nu;
            out[5] = r.lit_cost;
            out[6] = r.lit_norm1_nu;
            out[7] = r.lit_exitflag;
            out[8] = r.lit_iters;
            out[9] = 0.;
            if (Xlit && !r.Xlit.empty())
                std::memcpy(Xlit, r.Xlit.data(), r.Xlit.size() * sizeof(double));
            if (Ulit && !r.Ulit.empty())
                std::memcpy(Ulit, r.Ulit.data(), r.Ulit.size() * sizeof(double));
            return 0;
        });
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_scvx_check_point: %s\n", e.what());
        return -2;
    }
}
// per-solve rows: [norm1_nu, nonlinear_cost, actual_change, predicted_change, rho, trust_region(after), accepted, ipm_iters, exitflag]
int oracle_scvx_get_info(void *h, double *rows, int max_rows)
{
    return withScvx(h, [&](auto &a) {
        const int n = std::min<int>(max_rows, int(a.info.size()));
        for (int i = 0; i < n; i++)
        {
            const SCvxIterationInfo &f = a.info[size_t(i)];
            double *r = rows + i * 9;
            r[0] = f.norm1_nu;
            r[1] = f.nonlinear_cost;
            r[2] = f.actual_change;
            r[3] = f.predicted_change;
            r[4] = f.rho;
            r[5] = f.trust_region;
            r[6] = f.accepted;
            r[7] = f.ipm_iters;
            r[8] = f.exitflag;
        }
        return n;
    });
}
}

// ---- linear MPC (oracle/mpc.hpp) ----
namespace
{
struct MPCHandle
{
    std::shared_ptr<Rocket2d> model;
    std::unique_ptr<MPCAlgorithm> alg;
};
} // namespace
extern "C"
{
// exp(A), n x n row-major
void oracle_expm(int n, const double *A, double *E)
{
    const std::vector<double> r = expm(n, std::vector<double>(A, A + size_t(n) * n));
    std::memcpy(E, r.data(), r.size() * sizeof(double));
}
void *oracle_mpc_create(const char *config_root)
{
    try
    {
        auto h = new MPCHandle;
        const std::string folder = std::string(config_root) + "/Rocket2D";
        h->model = std::make_shared<Rocket2d>();
        h->model->loadParameters(folder);
        h->model->p.constrain_initial_final = false; // model.info: "enable for SC and disable for MPC/LQR"
        h->alg.reset(new MPCAlgorithm(h->model, folder));
        h->alg->initialize();
        return h;
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "oracle_mpc_create: %s\n", e.what());
        return nullptr;
    }
}
void oracle_mpc_destroy(void *h) { delete static_cast<MPCHandle *>(h); }
int oracle_mpc_K(void *h) { return int(static_cast<MPCHandle *>(h)->alg->K); }
void oracle_mpc_get_model(void *h, double *A, double *B, double *z, double *x_init, double *x_final)
{
    auto &a = *static_cast<MPCHandle *>(h)->alg;
    std::memcpy(A, a.A, sizeof a.A);
    std::memcpy(B, a.B, sizeof a.B);
    std::memcpy(z, a.z, sizeof a.z);
    std::memcpy(x_init, a.model->p.x_init, 6 * sizeof(double));
    std::memcpy(x_final, a.model->p.x_final, 6 * sizeof(double));
}
void oracle_mpc_set_tolerances(void *h, double feastol, double abstol, double reltol, int maxit)
{
    static_cast<MPCHandle *>(h)->alg->setTolerances(feastol, abstol, reltol, maxit);
}
// one solve; kind 0 literal (reference formulation, generic solver), 1 condensed twin.  info: [iters, pres, dres, gap, pcost, input_cost, error_cost]
int oracle_mpc_solve(void *h, int kind, const double *x_init, const double *x_final, double *X, double *U, double *info)
{
    auto &a = *static_cast<MPCHandle *>(h)->alg;
    a.solver_kind = kind;
    a.setInitialState(x_init);
    a.setFinalState(x_final);
    const int st = a.solve();
    if (st >= 0)
    {
        std::memcpy(X, a.X.data(), a.X.size() * sizeof(double));
        std::memcpy(U, a.U.data(), a.U.size() * sizeof(double));
    }
    info[0] = a.last.iters;
    info[1] = a.last.pres;
    info[2] = a.last.dres;
    info[3] = a.last.gap;
    info[4] = a.last.pcost;
    info[5] = a.input_cost;
    info[6] = a.error_cost;
    return st;
}
// closed loop (MPC_sim.cpp:49-86, deterministic step); out: x [6], u [2], meta [steps, failed, ipm_iters, reached]
void oracle_mpc_sim(void *h, int kind, const double *x_start, double sim_time, double time_step, int max_steps, double *x,
                    double *u, int *meta)
{
    auto &a = *static_cast<MPCHandle *>(h)->alg;
    a.solver_kind = kind;
    const MPCSimResult r = runMPCSim(a, x_start, sim_time, time_step, max_steps > 0 ? max_steps : (1 << 30));
    for (int i = 0; i < 6; i++)
        x[i] = r.steps ? r.X_sim[size_t(r.steps - 1) * 6 + i] : x_start[i];
    for (int i = 0; i < 2; i++)
        u[i] = r.steps ? r.U_sim[size_t(r.steps - 1) * 2 + i] : 0.;
    meta[0] = r.steps;
    meta[1] = r.failed_solves;
    meta[2] = r.ipm_iters;
    meta[3] = int(r.reached);
}
}
