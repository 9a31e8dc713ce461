// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the reference Successive-Convexification driver:
//   buildSCProblem                 scpp_core/src/SCProblem.cpp:6-138
//   SCAlgorithm::{loadParameters,initialize,iterate,solve,readSolution}
//                                  scpp_core/src/SCAlgorithm.cpp:22-210
// The sub-problem is re-assembled into standard form every iteration (the reference binds
// Epigraph `dynpar` pointers into td/dd instead; same numbers).
// Parity status: UNPINNED (the reference cannot be built or run here and ships no outputs); trajectory initialisation,
// weight doubling and convergence logic are tested against the statements of the reference source they restate.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <type_traits>
#include <string>
#include <vector>

#include "discretization.hpp"
#include "models.hpp"
#include "socp.hpp"
#include "structured_ipm.hpp"

namespace oracle
{

// elimination-order keys handed to the sparse LDL (not part of the maths)
struct SCKeys
{
    int stageCone(int) const { return 0; }
    int nuBound() const { return 1; }
    int nu() const { return 2; }
    int stageVar(int k) const { return 10 + 3 * k; }
    int stageEq(int k) const { return 10 + 3 * k + 1; }
    int dyn(int k) const { return 10 + 3 * k + 2; }
    int globalCone() const { return 1000000; }
    int globalVar() const { return 1000001; }
};

struct SCVarIndex
{
    int NX, NU, K, nU;
    int X, U, nu, nu_bound, norm1_nu, delta, sigma, delta_sigma;
    int vX(int i, int k) const { return X + k * NX + i; }
    int vU(int i, int k) const { return U + k * NU + i; }
    int vNu(int i, int k) const { return nu + k * NX + i; }
    int vNuB(int i, int k) const { return nu_bound + k * NX + i; }
};

// SCProblem.cpp:6-138
template <class Model>
Socp buildSCProblem(double weight_time, double weight_trust_region_time, double weight_trust_region_trajectory,
                    double weight_virtual_control, const TrajectoryData &td, const DiscretizationData &dd,
                    SCVarIndex &ix)
{
    constexpr int NX = Model::NX, NU = Model::NU;
    const int K = td.K, nU = td.nU;
    SCKeys key;
    Socp socp;
    ix.NX = NX;
    ix.NU = NU;
    ix.K = K;
    ix.nU = nU;
    ix.X = socp.addVars(NX * K, 0);
    ix.U = socp.addVars(NU * nU, 0);
    for (int k = 0; k < K; k++)
        for (int i = 0; i < NX; i++)
            socp.setKey(ix.vX(i, k), key.stageVar(k));
    for (int k = 0; k < nU; k++)
        for (int i = 0; i < NU; i++)
            socp.setKey(ix.vU(i, k), key.stageVar(k));
    ix.nu = socp.addVars(NX * (K - 1), key.nu());
    ix.nu_bound = socp.addVars(NX * (K - 1), key.nuBound());
    ix.norm1_nu = socp.addVars(1, key.globalVar());
    ix.delta = socp.addVars(K, 0);
    for (int k = 0; k < K; k++)
        socp.setKey(ix.delta + k, key.stageVar(k));
    ix.sigma = ix.delta_sigma = -1;
    if (dd.variableTime())
    {
        ix.sigma = socp.addVars(1, key.globalVar());
        ix.delta_sigma = socp.addVars(1, key.globalVar());
        socp.c[ix.sigma] += weight_time;
        // sigma >= 0.001
        socp.addGe0(Aff(-0.001).add(ix.sigma, 1.), key.globalCone());
    }

    // dynamics: A x_k + B u_k + z + nu_k (+ C u_{k+1}) (+ s sigma) - x_{k+1} == 0
    for (int k = 0; k < K - 1; k++)
    {
        const double *A = &dd.A[size_t(k) * NX * NX];
        const double *B = &dd.B[size_t(k) * NX * NU];
        for (int i = 0; i < NX; i++)
        {
            Aff e(dd.z[size_t(k) * NX + i]);
            for (int j = 0; j < NX; j++)
                if (A[i * NX + j] != 0.)
                    e.add(ix.vX(j, k), A[i * NX + j]);
            for (int j = 0; j < NU; j++)
                if (B[i * NU + j] != 0.)
                    e.add(ix.vU(j, k), B[i * NU + j]);
            e.add(ix.vNu(i, k), 1.);
            if (dd.interpolatedInput())
            {
                const double *C = &dd.C[size_t(k) * NX * NU];
                for (int j = 0; j < NU; j++)
                    if (C[i * NU + j] != 0.)
                        e.add(ix.vU(j, k + 1), C[i * NU + j]);
            }
            if (dd.variableTime())
                e.add(ix.sigma, dd.s[size_t(k) * NX + i]);
            e.add(ix.vX(i, k + 1), -1.);
            socp.addEq(e, key.dyn(k));
        }
    }

    // virtual control norm: -nu_bound <= nu <= nu_bound ; sum(nu_bound) <= norm1_nu
    for (int k = 0; k < K - 1; k++)
        for (int i = 0; i < NX; i++)
        {
            socp.addGe0(Aff().add(ix.vNu(i, k), 1.).add(ix.vNuB(i, k), 1.), key.stageCone(k));
            socp.addGe0(Aff().add(ix.vNuB(i, k), 1.).add(ix.vNu(i, k), -1.), key.stageCone(k));
        }
    {
        Aff e;
        e.add(ix.norm1_nu, 1.);
        for (int k = 0; k < K - 1; k++)
            for (int i = 0; i < NX; i++)
                e.add(ix.vNuB(i, k), -1.);
        socp.addGe0(e, key.globalCone());
        socp.c[ix.norm1_nu] += weight_virtual_control;
    }

    if (dd.variableTime())
    {
        // || (0.5 - 0.5 ds, sigma - sigma0) || <= 0.5 + 0.5 ds
        socp.addSoc({Aff(0.5).add(ix.delta_sigma, 0.5), Aff(0.5).add(ix.delta_sigma, -0.5),
                     Aff(-td.t).add(ix.sigma, 1.)},
                    key.globalCone());
        socp.c[ix.delta_sigma] += weight_trust_region_time;
    }

    // trust regions || (x0 - x ; u0 - u) || <= delta_k
    for (int k = 0; k < K; k++)
    {
        std::vector<Aff> e;
        e.push_back(Aff().add(ix.delta + k, 1.));
        for (int i = 0; i < NX; i++)
            e.push_back(Aff(td.x(k)[i]).add(ix.vX(i, k), -1.));
        if (dd.interpolatedInput() || k < K - 1)
            for (int i = 0; i < NU; i++)
                e.push_back(Aff(td.u(k)[i]).add(ix.vU(i, k), -1.));
        socp.addSoc(e, key.stageCone(k));
        socp.c[ix.delta + k] += weight_trust_region_trajectory;
    }
    return socp;
}

struct SCIterationInfo
{
    double norm1_nu, sum_delta, delta_sigma, sigma;
    int ipm_iters, exitflag;
    double pres, dres, gap;
};

template <class Model>
class SCAlgorithm
{
  public:
    Model *model;
    std::string param_folder;
    int K_override = 0;
    SocpSettings socp_settings;

    size_t K = 0;
    bool free_final_time = true, interpolate_input = true, nondimensionalize = true;
    double weight_time, weight_trust_region_time = 0., weight_trust_region_trajectory, weight_virtual_control;
    double nu_tol, delta_tol;
    size_t max_iterations;

    DiscretizationData dd;
    TrajectoryData td;
    std::vector<TrajectoryData> all_td;
    std::vector<SCIterationInfo> info;
    bool converged = false;
    int iterations = 0;
    bool solver_failed = false;

    SCAlgorithm(Model *m, const std::string &folder, int K_over = 0) : model(m), param_folder(folder), K_override(K_over)
    {
        loadParameters();
    }

    // SCAlgorithm.cpp:22-46
    void loadParameters()
    {
        ParameterServer param(param_folder + "/SC.info");
        param.loadScalar("K", K);
        if (K_override > 0)
            K = size_t(K_override);
        param.loadScalar("free_final_time", free_final_time);
        param.loadScalar("nondimensionalize", nondimensionalize);
        param.loadScalar("delta_tol", delta_tol);
        param.loadScalar("max_iterations", max_iterations);
        param.loadScalar("nu_tol", nu_tol);
        param.loadScalar("weight_time", weight_time);
        param.loadScalar("weight_virtual_control", weight_virtual_control);
        param.loadScalar("weight_trust_region_trajectory", weight_trust_region_trajectory);
        param.loadScalar("interpolate_input", interpolate_input);
        if (free_final_time)
            param.loadScalar("weight_trust_region_time", weight_trust_region_time);
    }

    // SCAlgorithm.cpp:48-64
    void initialize()
    {
        dd.initialize(Model::NX, Model::NU, int(K), interpolate_input, free_final_time);
        td.initialize(Model::NX, Model::NU, int(K), interpolate_input);
    }

    int solver_kind = 0; // 0: literal standard form + ECOS-style solver (socp.hpp); 1: structured IPM twin
    RQSocpSettings structured_settings;

    // same sub-problem, solved by the structured IPM (RocketQuat, FOH, free final time, roll control off)
    template <class M = Model>
    typename std::enable_if<std::is_same<M, RocketQuat>::value, bool>::type iterateStructured()
    {
        RQSocpInput in;
        in.K = td.K;
        in.Xbar = td.X.data();
        in.Ubar = td.U.data();
        in.sigbar = td.t;
        in.A = dd.A.data();
        in.B = dd.B.data();
        in.C = dd.C.data();
        in.S = dd.s.data();
        in.Z = dd.z.data();
        in.x_init = model->p.x_init;
        in.x_final = model->p.x_final;
        std::vector<double> uhat(size_t(td.K) * 3, 0.);
        for (int k = 0; k < td.K; k++)
        {
            if (model->p.exact_minimum_thrust)
                for (int i = 0; i < 3; i++)
                    uhat[size_t(k) * 3 + i] = model->p_dyn.thrust_const[size_t(k) * 3 + i];
            else
                uhat[size_t(k) * 3 + 2] = 1.;
        }
        in.uhat = uhat.data();
        in.cst.gs = model->p_dyn.gs_const;
        in.cst.tilt = model->p_dyn.tilt_const;
        in.cst.wmax = model->p.w_B_max;
        in.cst.Tmin = model->p.T_min;
        in.cst.Tmax = model->p.T_max;
        in.cst.gim = model->p_dyn.gimbal_const;
        in.cst.mdry = model->p.x_final[0];
        in.w_t = weight_time;
        in.w_trt = weight_trust_region_time;
        in.w_trx = weight_trust_region_trajectory;
        in.w_vc = weight_virtual_control;
        // sub-problems of consecutive SC iterations are close: the interior-point iteration restarts from the previous
        // solve's point (structured_ipm.hpp: warm start); ORACLE_WARM=0 forces ECOS-style cold starts
        const char *we = std::getenv("ORACLE_WARM");
        const bool warm = ipm_warm_start && !(we && std::atoi(we) == 0);
        twin.opt = structured_settings;
        RQSocpOutput r = twin.solve(in, warm);
        last_structured = r;
        if (r.status != 0)
        {
            solver_failed = true;
            SCIterationInfo inf{0, 0, 0, td.t, r.iters, r.status, r.pres, r.dres, r.gap};
            info.push_back(inf);
            return false;
        }
        td.t = r.sigma;
        td.X = r.X;
        td.U = r.U;
        if (r.norm1_nu < nu_tol)
            weight_trust_region_trajectory *= 2.;
        SCIterationInfo inf{r.norm1_nu, r.sum_delta, r.delta_sigma, td.t, r.iters, r.status, r.pres, r.dres, r.gap};
        info.push_back(inf);
        return r.sum_delta < delta_tol && r.norm1_nu < nu_tol;
    }
    template <class M = Model>
    typename std::enable_if<!std::is_same<M, RocketQuat>::value, bool>::type iterateStructured()
    {
        throw std::runtime_error("structured IPM: RocketQuat only");
    }
    RQSocpOutput last_structured;
    RQStructuredSocp twin;
    bool ipm_warm_start = true;

    // SCAlgorithm.cpp:66-132
    bool iterate()
    {
        multipleShooting(*model, td, dd);
        if (solver_kind == 1)
            return iterateStructured();
        SCVarIndex ix;
        Socp socp = buildSCProblem<Model>(weight_time, weight_trust_region_time, weight_trust_region_trajectory,
                                          weight_virtual_control, td, dd, ix);
        SCKeys key;
        model->addApplicationConstraints(
            socp, td.K, td.nU, [&](int i, int k) { return ix.vX(i, k); }, [&](int i, int k) { return ix.vU(i, k); }, key);
        last_dims[0] = socp.n;
        last_dims[1] = socp.numEq();
        last_dims[2] = socp.numLp();
        last_dims[3] = int(socp.soc.size());
        last_dims[4] = socp.numConeRows();
        last_ix = ix;
        SocpSolver solver(socp);
        solver.opt = socp_settings;
        SocpResult r = solver.solve();
        last_result = r;
        if (r.exitflag != 0 && r.exitflag != 10) // 10 = ECOS_OPTIMAL + ECOS_INACC_OFFSET: accepted like the twin / device do
        {
            solver_failed = true; // reference: std::terminate() (SCAlgorithm.cpp:94-98)
            SCIterationInfo inf{0, 0, 0, td.t, r.iter, r.exitflag, r.pres, r.dres, r.gap};
            info.push_back(inf);
            return false;
        }
        // readSolution  SCAlgorithm.cpp:191-210
        if (free_final_time)
            td.t = r.x[ix.sigma];
        for (int k = 0; k < td.K; k++)
            for (int i = 0; i < Model::NX; i++)
                td.x(k)[i] = r.x[ix.vX(i, k)];
        for (int k = 0; k < td.nU; k++)
            for (int i = 0; i < Model::NU; i++)
                td.u(k)[i] = r.x[ix.vU(i, k)];
        const double norm1_nu = r.x[ix.norm1_nu];
        double sum_delta = 0.;
        for (int k = 0; k < td.K; k++)
            sum_delta += r.x[ix.delta + k];
        const double delta_sigma = free_final_time ? r.x[ix.delta_sigma] : 0.;
        if (norm1_nu < nu_tol)
            weight_trust_region_trajectory *= 2.;
        SCIterationInfo inf{norm1_nu, sum_delta, delta_sigma, td.t, r.iter, r.exitflag, r.pres, r.dres, r.gap};
        info.push_back(inf);
        return sum_delta < delta_tol && norm1_nu < nu_tol;
    }

    // SCAlgorithm.cpp:134-189
    void solve(bool warm_start = false)
    {
        if (nondimensionalize)
            model->nondimensionalize();
        if (warm_start)
        {
            if (nondimensionalize)
                model->nondimensionalizeTrajectory(td);
        }
        else
        {
            loadParameters();
            model->getInitializedTrajectory(td);
            twin.have_prev = false; // cold SC solve: cold interior-point start
        }
        model->getNewModelParameters(td); // updateModelParameters()
        size_t iteration = 0;
        converged = false;
        solver_failed = false;
        all_td.push_back(td);
        while (iteration < max_iterations && !converged && !solver_failed)
        {
            iteration++;
            converged = iterate();
            all_td.push_back(td);
        }
        iterations = int(iteration);
        if (nondimensionalize)
        {
            model->redimensionalize();
            model->getNewModelParameters(td);
            model->redimensionalizeTrajectory(td);
        }
    }

    // ---- test support: one candidate point in the LITERAL (reference-shaped) SC sub-problem (cf. scvx.hpp: checkPoint) ----
    // Problem: buildSCProblem + addApplicationConstraints linearised at the DIMENSIONAL trajectory (Xbar, Ubar, tbar) with the
    // given trust-region weight (SCAlgorithm doubles it along the run, SCAlgorithm.cpp:112-115; <= 0: the configured one) for
    // the model's current x_init, thrust_const as a cold solve() sets it.  Candidate (Xc, Uc, tc), dimensional: nu := dynamics
    // defect, nu_bound := |nu|, norm1_nu := sum, delta_k := ||(x - xbar; u - ubar)||, delta_sigma := (sigma - sigmabar)^2 -- the
    // cheapest completion -- then every row of the literal standard form is evaluated.
    struct PointCheck
    {
        double eq_violation = 0., min_lp_slack = 0., min_cone_slack = 0., cost = 0., norm1_nu = 0., sum_delta = 0.;
        double lit_cost = 0., lit_sigma = 0.;
        int lit_exitflag = -99, lit_iters = 0;
        std::vector<double> Xlit, Ulit;
    };
    PointCheck checkPoint(const double *Xbar, const double *Ubar, double tbar, double w_trx, const double *Xc, const double *Uc, double tc,
                          bool solve_literal)
    {
        constexpr int NX = Model::NX, NU = Model::NU;
        PointCheck out;
        loadParameters();
        if (w_trx > 0.)
            weight_trust_region_trajectory = w_trx;
        if (nondimensionalize)
            model->nondimensionalize();
        {
            TrajectoryData init;
            init.initialize(NX, NU, int(K), interpolate_input);
            model->getInitializedTrajectory(init);
            model->getNewModelParameters(init);
        }
        td.X.assign(Xbar, Xbar + size_t(td.K) * NX);
        td.U.assign(Ubar, Ubar + size_t(td.nU) * NU);
        td.t = tbar;
        TrajectoryData cand = td;
        cand.X.assign(Xc, Xc + size_t(td.K) * NX);
        cand.U.assign(Uc, Uc + size_t(td.nU) * NU);
        cand.t = tc;
        if (nondimensionalize)
        {
            model->nondimensionalizeTrajectory(td);
            model->nondimensionalizeTrajectory(cand);
        }
        multipleShooting(*model, td, dd);
        SCVarIndex ix;
        Socp socp = buildSCProblem<Model>(weight_time, weight_trust_region_time, weight_trust_region_trajectory, weight_virtual_control, td, dd, ix);
        SCKeys key;
        model->addApplicationConstraints(
            socp, td.K, td.nU, [&](int i, int k) { return ix.vX(i, k); }, [&](int i, int k) { return ix.vU(i, k); }, key);
        std::vector<double> x(size_t(socp.n), 0.);
        for (int k = 0; k < td.K; k++)
            for (int i = 0; i < NX; i++)
                x[size_t(ix.vX(i, k))] = cand.x(k)[i];
        for (int k = 0; k < td.nU; k++)
            for (int i = 0; i < NU; i++)
                x[size_t(ix.vU(i, k))] = cand.u(k)[i];
        if (ix.sigma >= 0)
        {
            x[size_t(ix.sigma)] = cand.t;
            x[size_t(ix.delta_sigma)] = (cand.t - td.t) * (cand.t - td.t);
        }
        double n1 = 0.;
        for (int k = 0; k + 1 < td.K; k++)
        {
            const double *A = &dd.A[size_t(k) * NX * NX], *B = &dd.B[size_t(k) * NX * NU];
            for (int i = 0; i < NX; i++)
            {
                double v = cand.x(k + 1)[i] - dd.z[size_t(k) * NX + i];
                for (int j = 0; j < NX; j++)
                    v -= A[i * NX + j] * cand.x(k)[j];
                for (int j = 0; j < NU; j++)
                    v -= B[i * NU + j] * cand.u(k)[j];
                if (dd.interpolatedInput())
                    for (int j = 0; j < NU; j++)
                        v -= dd.C[size_t(k) * NX * NU + i * NU + j] * cand.u(k + 1)[j];
                if (dd.variableTime())
                    v -= dd.s[size_t(k) * NX + i] * cand.t;
                x[size_t(ix.vNu(i, k))] = v;
                x[size_t(ix.vNuB(i, k))] = std::fabs(v);
                n1 += std::fabs(v);
            }
        }
        x[size_t(ix.norm1_nu)] = n1;
        out.norm1_nu = n1;
        for (int k = 0; k < td.K; k++)
        {
            double d2 = 0.;
            for (int i = 0; i < NX; i++)
                d2 += (td.x(k)[i] - cand.x(k)[i]) * (td.x(k)[i] - cand.x(k)[i]);
            if (dd.interpolatedInput() || k < td.K - 1)
                for (int i = 0; i < NU; i++)
                    d2 += (td.u(k)[i] - cand.u(k)[i]) * (td.u(k)[i] - cand.u(k)[i]);
            x[size_t(ix.delta + k)] = std::sqrt(d2);
            out.sum_delta += std::sqrt(d2);
        }
        auto rowValue = [&](const SocpRow &r) {
            double v = 0.;
            for (auto &t : r.t)
                v += t.second * x[size_t(t.first)];
            return v;
        };
        for (auto &r : socp.eq)
            out.eq_violation = std::max(out.eq_violation, std::fabs(rowValue(r) - r.rhs));
        out.min_lp_slack = 1e300;
        for (auto &r : socp.lp)
            out.min_lp_slack = std::min(out.min_lp_slack, r.rhs - rowValue(r));
        out.min_cone_slack = 1e300;
        for (auto &cn : socp.soc)
        {
            double s0 = cn[0].rhs - rowValue(cn[0]), nn = 0.;
            for (size_t i = 1; i < cn.size(); i++)
            {
                const double si = cn[i].rhs - rowValue(cn[i]);
                nn += si * si;
            }
            out.min_cone_slack = std::min(out.min_cone_slack, s0 - std::sqrt(nn));
        }
        for (int j = 0; j < socp.n; j++)
            out.cost += socp.c[size_t(j)] * x[size_t(j)];
        if (solve_literal)
        {
            SocpSolver solver(socp);
            solver.opt = socp_settings;
            SocpResult r = solver.solve();
            out.lit_exitflag = r.exitflag;
            out.lit_iters = r.iter;
            if (r.exitflag == 0 || r.exitflag == 10)
            {
                out.lit_cost = r.pcost;
                TrajectoryData sol = td;
                for (int k = 0; k < td.K; k++)
                    for (int i = 0; i < NX; i++)
                        sol.x(k)[i] = r.x[size_t(ix.vX(i, k))];
                for (int k = 0; k < td.nU; k++)
                    for (int i = 0; i < NU; i++)
                        sol.u(k)[i] = r.x[size_t(ix.vU(i, k))];
                out.lit_sigma = ix.sigma >= 0 ? r.x[size_t(ix.sigma)] : td.t;
                if (nondimensionalize)
                    model->redimensionalizeTrajectory(sol);
                out.Xlit = sol.X;
                out.Ulit = sol.U;
            }
        }
        if (nondimensionalize)
            model->redimensionalize();
        loadParameters(); // restores the configured trust-region weight
        return out;
    }

    int last_dims[5] = {0, 0, 0, 0, 0};
    SocpResult last_result;
    SCVarIndex last_ix;
};

} // namespace oracle
