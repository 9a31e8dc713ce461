// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the reference's SCvx variant (fixed final time, hard input trust region, rho-ratio radius update):
//   buildSCvxProblem                                   scpp_core/src/SCvxProblem.cpp:6-71
//   SCvxAlgorithm::{loadParameters,initialize,iterate,solve,readSolution,getNonlinearCost}
//                                                      scpp_core/src/SCvxAlgorithm.cpp:22-278
// No executable of the reference calls SCvxAlgorithm; it is compiled into libscpp.so and restated
// here from its source.  Parity status: unpinned at the ECOS boundary like the SC path.
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "discretization.hpp"
#include "models.hpp"
#include "sc.hpp"
#include "socp.hpp"
#include "structured_ipm.hpp"

namespace oracle
{

// elimination-order keys for the sparse LDL (not part of the maths).  The SCvx sub-problem has no Hessian on the
// states (no state trust region), so the dynamics rows -- which carry a proper -E^-1 diagonal once nu is eliminated --
// must be pivoted BEFORE the stage variables they couple; the SC order (variables first) would pivot on the static
// regularisation alone.
struct SCvxKeys
{
    int stageCone(int) const { return 0; }
    int nuBound() const { return 1; }
    int nu() const { return 2; }
    int dyn(int k) const { return 10 + 3 * k; }
    int stageVar(int k) const { return 10 + 3 * k + 1; }
    int stageEq(int k) const { return 10 + 3 * k + 2; }
    int globalCone() const { return 1000000; }
    int globalVar() const { return 1000001; }
};

// SCvxProblem.cpp:6-71
template <class Model>
Socp buildSCvxProblem(double trust_region, double weight_virtual_control, const TrajectoryData &td,
                      const DiscretizationData &dd, SCVarIndex &ix)
{
    constexpr int NX = Model::NX, NU = Model::NU;
    const int K = td.K, nU = td.nU;
    SCvxKeys key;
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
    ix.delta = ix.sigma = ix.delta_sigma = -1;

    // x(k+1) == A x(k) + B u(k) + C u(k+1) + z + nu      (:20-40)
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
            if (td.interpolatedInput())
            {
                const double *C = &dd.C[size_t(k) * NX * NU];
                for (int j = 0; j < NU; j++)
                    if (C[i * NU + j] != 0.)
                        e.add(ix.vU(j, k + 1), C[i * NU + j]);
            }
            e.add(ix.vX(i, k + 1), -1.);
            socp.addEq(e, key.dyn(k));
        }
    }
    // -nu_bound <= nu <= nu_bound ; sum(nu_bound) <= norm1_nu ; minimise w_vc norm1_nu      (:42-56)
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
    // input trust region norm2(u0 - u) <= trust_region      (:58-68)
    for (int k = 0; k < nU; k++)
    {
        std::vector<Aff> e;
        e.push_back(Aff(trust_region));
        for (int i = 0; i < NU; i++)
            e.push_back(Aff(td.u(k)[i]).add(ix.vU(i, k), -1.));
        socp.addSoc(e, key.stageCone(k));
    }
    return socp;
}

struct SCvxIterationInfo
{
    double norm1_nu, nonlinear_cost, actual_change, predicted_change, rho, trust_region;
    int accepted; // 1 accepted, 0 rejected (re-solve with smaller radius), 2 first pass, 3 converged
    int ipm_iters, exitflag;
};

template <class Model>
class SCvxAlgorithm
{
  public:
    Model *model;
    std::string param_folder;
    int K_override = 0;
    SocpSettings socp_settings;
    RQSocpSettings structured_settings;
    int solver_kind = 0; // 0 literal standard form + ECOS-style solver, 1 structured twin (RocketQuat)

    size_t K = 0;
    bool interpolate_input = true, nondimensionalize = true;
    size_t max_iterations = 0;
    double alpha = 0, beta = 0, rho_0 = 0, rho_1 = 0, rho_2 = 0, change_threshold = 0, weight_virtual_control = 0, trust_region = 0;

    DiscretizationData dd;
    TrajectoryData td;
    std::vector<TrajectoryData> all_td;
    std::vector<SCvxIterationInfo> info;
    bool has_last = false;
    double last_nonlinear_cost = 0.;
    bool converged = false, solver_failed = false;
    int iterations = 0, solves = 0;
    // Test support, NOT the reference: SCvxAlgorithm::iterate's `while (true)` (SCvxAlgorithm.cpp:75-153) leaves only through an accepted
    // candidate, and with the shipped Rocket2D SCvx.info some start states are rejected indefinitely.  The device engine retires such an
    // instance on a rejection once it has used solve_cap x max_iterations sub-problem solves (csrc/scvx_kernels.h: SCVX_SOLVE_CAP = 64,
    // status SCPP_STATUS_REJECTION_CAP, last accepted iterate kept); with solve_cap > 0 this restatement does the same so that the two can
    // be compared at the cap.  0 (default) = the reference's behaviour: no exit.
    size_t solve_cap = 0;
    bool retired = false;

    SCvxAlgorithm(Model *m, const std::string &folder, int K_over = 0) : model(m), param_folder(folder), K_override(K_over)
    {
        loadParameters();
    }

    // SCvxAlgorithm.cpp:22-44
    void loadParameters()
    {
        ParameterServer param(param_folder + "/SCvx.info");
        param.loadScalar("K", K);
        if (K_override > 0)
            K = size_t(K_override);
        param.loadScalar("nondimensionalize", nondimensionalize);
        param.loadScalar("max_iterations", max_iterations);
        param.loadScalar("alpha", alpha);
        param.loadScalar("beta", beta);
        param.loadScalar("rho_0", rho_0);
        param.loadScalar("rho_1", rho_1);
        param.loadScalar("rho_2", rho_2);
        param.loadScalar("change_threshold", change_threshold);
        param.loadScalar("weight_virtual_control", weight_virtual_control);
        param.loadScalar("trust_region", trust_region);
        param.loadScalar("interpolate_input", interpolate_input);
        if (max_iterations_override > 0)
            max_iterations = max_iterations_override; // test hook (the cold start re-reads the file, :179)
    }
    size_t max_iterations_override = 0;
    // SCvxAlgorithm.cpp:46-59
    void initialize()
    {
        dd.initialize(Model::NX, Model::NU, int(K), interpolate_input, false);
        td.initialize(Model::NX, Model::NU, int(K), interpolate_input);
    }

    // SCvxAlgorithm.cpp:262-278
    double getNonlinearCost()
    {
        double cost = 0.;
        for (size_t k = 0; k + 1 < K; k++)
        {
            double x[Model::NX];
            for (int i = 0; i < Model::NX; i++)
                x[i] = td.x(int(k))[i];
            const double *u0 = td.u(int(k));
            const double *u1 = interpolate_input ? td.u(int(k) + 1) : u0;
            simulate(*model, td.t / double(K - 1), u0, u1, x);
            for (int i = 0; i < Model::NX; i++)
                cost += std::fabs(x[i] - td.x(int(k) + 1)[i]);
        }
        return cost;
    }

    // one sub-problem solve; fills td on success, returns norm1_nu
    bool solveSubproblem(double &norm1_nu, int &ipm_iters, int &exitflag)
    {
        solves++;
        if (solver_kind == 1)
            return solveStructured(norm1_nu, ipm_iters, exitflag);
        SCVarIndex ix;
        Socp socp = buildSCvxProblem<Model>(trust_region, weight_virtual_control, td, dd, ix);
        SCvxKeys key;
        model->addApplicationConstraints(
            socp, td.K, td.nU, [&](int i, int k) { return ix.vX(i, k); }, [&](int i, int k) { return ix.vU(i, k); }, key);
        last_dims[0] = socp.n;
        last_dims[1] = socp.numEq();
        last_dims[2] = socp.numLp();
        last_dims[3] = int(socp.soc.size());
        last_dims[4] = socp.numConeRows();
        SocpSolver solver(socp);
        solver.opt = socp_settings;
        SocpResult r = solver.solve();
        ipm_iters = r.iter;
        exitflag = r.exitflag;
        // ECOS_OPTIMAL, or ECOS_OPTIMAL + ECOS_INACC_OFFSET ("close to optimal": a breakdown / iteration limit at an iterate that
        // meets the reduced tolerances) -- the same acceptance rule as the structured twin and the device solver
        if (r.exitflag != 0 && r.exitflag != 10)
            return false;
        // readSolution  SCvxAlgorithm.cpp:229-243
        for (int k = 0; k < td.K; k++)
            for (int i = 0; i < Model::NX; i++)
                td.x(k)[i] = r.x[ix.vX(i, k)];
        for (int k = 0; k < td.nU; k++)
            for (int i = 0; i < Model::NU; i++)
                td.u(k)[i] = r.x[ix.vU(i, k)];
        norm1_nu = r.x[ix.norm1_nu];
        return true;
    }
    template <class M = Model>
    typename std::enable_if<std::is_same<M, RocketQuat>::value, bool>::type solveStructured(double &norm1_nu, int &ipm_iters, int &exitflag)
    {
        RQSocpInput in;
        in.K = td.K;
        in.Xbar = td.X.data();
        in.Ubar = td.U.data();
        in.sigbar = td.t;
        std::vector<double> S0(size_t(td.K - 1) * Model::NX, 0.);
        in.A = dd.A.data();
        in.B = dd.B.data();
        in.C = dd.C.data();
        in.S = S0.data();
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
        in.w_t = 1.;  // dummy decoupled sigma block (S = 0)
        in.w_trt = 1.;
        in.w_trx = 0.;
        in.w_vc = weight_virtual_control;
        in.scvx = true;
        in.trust_region = trust_region;
        twin.opt = structured_settings;
        RQSocpOutput r = twin.solve(in, true); // warm interior-point start from the previous sub-problem (sc.hpp)
        ipm_iters = r.iters;
        exitflag = r.status;
        if (r.status != 0)
            return false;
        td.X = r.X;
        td.U = r.U;
        norm1_nu = r.norm1_nu;
        return true;
    }
    template <class M = Model>
    typename std::enable_if<!std::is_same<M, RocketQuat>::value, bool>::type solveStructured(double &, int &, int &)
    {
        throw std::runtime_error("structured IPM: RocketQuat only");
    }

    // SCvxAlgorithm.cpp:61-164
    bool iterate()
    {
        multipleShooting(*model, td, dd);
        bool conv = false;
        while (true)
        {
            const TrajectoryData old_td = td;
            double norm1_nu = 0.;
            int ipm_iters = 0, exitflag = 0;
            if (!solveSubproblem(norm1_nu, ipm_iters, exitflag))
            {
                solver_failed = true; // reference: std::terminate() (SCvxAlgorithm.cpp:87-91)
                info.push_back({0, 0, 0, 0, 0, trust_region, -1, ipm_iters, exitflag});
                return false;
            }
            const double nonlinear_cost = getNonlinearCost(); // J
            const double linear_cost = norm1_nu;              // L
            if (!has_last)
            {
                has_last = true;
                last_nonlinear_cost = nonlinear_cost;
                info.push_back({norm1_nu, nonlinear_cost, 0, 0, 0, trust_region, 2, ipm_iters, exitflag});
                break;
            }
            const double actual_change = last_nonlinear_cost - nonlinear_cost;
            const double predicted_change = last_nonlinear_cost - linear_cost;
            last_nonlinear_cost = nonlinear_cost; // (overwritten even when the candidate is rejected, :118)
            if (std::fabs(predicted_change) < change_threshold)
            {
                conv = true;
                info.push_back({norm1_nu, nonlinear_cost, actual_change, predicted_change, 0, trust_region, 3, ipm_iters, exitflag});
                break;
            }
            const double rho = actual_change / predicted_change;
            if (rho < rho_0)
            {
                trust_region /= alpha;
                td = old_td;
                info.push_back({norm1_nu, nonlinear_cost, actual_change, predicted_change, rho, trust_region, 0, ipm_iters, exitflag});
                if (solve_cap && size_t(solves) >= solve_cap * max_iterations)
                {
                    retired = true; // (build-defined, see solve_cap)
                    return false;
                }
            }
            else
            {
                if (rho < rho_1)
                    trust_region /= alpha;
                else if (rho >= rho_2)
                    trust_region *= beta;
                info.push_back({norm1_nu, nonlinear_cost, actual_change, predicted_change, rho, trust_region, 1, ipm_iters, exitflag});
                break;
            }
        }
        return conv;
    }

    // SCvxAlgorithm.cpp:166-227
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
            twin.have_prev = false; // cold solve: cold interior-point start
        }
        model->getNewModelParameters(td); // updateModelParameters()
        size_t iteration = 0;
        all_td.push_back(td);
        converged = false;
        solver_failed = false;
        retired = false;
        while (iteration < max_iterations && !converged && !solver_failed && !retired)
        {
            iteration++;
            converged = iterate();
            all_td.push_back(td);
        }
        iterations = int(iteration); // (a retired instance reports the iteration it was retired in, like the device's sc_iters)
        if (nondimensionalize)
        {
            model->redimensionalize();
            model->getNewModelParameters(td);
            model->redimensionalizeTrajectory(td);
        }
    }

    // ---- test support: one candidate point in the LITERAL (reference-shaped) sub-problem ----
    // Builds buildSCvxProblem + addApplicationConstraints (SCvxProblem.cpp:6-71, rocketQuat.cpp:70-144) linearised at the given
    // DIMENSIONAL trajectory (Xbar, Ubar) with trust radius `radius` for the model's current x_init -- exactly what
    // SCvxAlgorithm::iterate hands to the solver in a cold solve() -- and evaluates a DIMENSIONAL candidate (Xc, Uc) in it:
    // nu := the dynamics defect, nu_bound := |nu|, norm1_nu := sum |nu| (the cheapest completion), then every equality, LP row
    // and second-order cone of the literal standard form.  With solve_literal the same problem is solved by the literal solver.
    // Used by the parity tests to CHECK the statement "two solvers' inputs differ because the optimum is not unique": both
    // points feasible in the same literal problem with the same objective.
    struct PointCheck
    {
        double eq_violation = 0.;   // max |Ax - b| over the equality rows (dynamics rows are 0 by construction of nu)
        double min_lp_slack = 0.;   // min over LP rows of (h - Gx); >= 0 <=> feasible
        double min_cone_slack = 0.; // min over cones of s0 - ||s1||
        double cost = 0.;           // c'x = weight_virtual_control * sum |nu|
        double norm1_nu = 0.;
        double lit_cost = 0., lit_norm1_nu = 0.;
        int lit_exitflag = -99, lit_iters = 0;
        std::vector<double> Xlit, Ulit; // dimensional optimum of the literal solver
    };
    // bar_nondim / cand_nondim: that trajectory is already in the solver's nondimensional units (e.g. an entry of all_td)
    PointCheck checkPoint(const double *Xbar, const double *Ubar, double radius, const double *Xc, const double *Uc, bool solve_literal,
                          bool bar_nondim = false, bool cand_nondim = false)
    {
        constexpr int NX = Model::NX, NU = Model::NU;
        PointCheck out;
        loadParameters();
        if (nondimensionalize)
            model->nondimensionalize();
        {
            // thrust_const of a cold solve(): refreshed once, from the initial trajectory (quirk F9(d), SCvxAlgorithm.cpp:181)
            TrajectoryData init;
            init.initialize(NX, NU, int(K), interpolate_input);
            model->getInitializedTrajectory(init);
            model->getNewModelParameters(init);
            td.t = init.t;
        }
        td.X.assign(Xbar, Xbar + size_t(td.K) * NX);
        td.U.assign(Ubar, Ubar + size_t(td.nU) * NU);
        TrajectoryData cand = td;
        cand.X.assign(Xc, Xc + size_t(td.K) * NX);
        cand.U.assign(Uc, Uc + size_t(td.nU) * NU);
        if (nondimensionalize && !bar_nondim)
            model->nondimensionalizeTrajectory(td);
        if (nondimensionalize && !cand_nondim)
            model->nondimensionalizeTrajectory(cand);
        multipleShooting(*model, td, dd);
        const double keep_radius = trust_region;
        trust_region = radius;
        SCVarIndex ix;
        Socp socp = buildSCvxProblem<Model>(trust_region, weight_virtual_control, td, dd, ix);
        SCvxKeys key;
        model->addApplicationConstraints(
            socp, td.K, td.nU, [&](int i, int k) { return ix.vX(i, k); }, [&](int i, int k) { return ix.vU(i, k); }, key);
        std::vector<double> x(size_t(socp.n), 0.);
        for (int k = 0; k < td.K; k++)
            for (int i = 0; i < NX; i++)
                x[size_t(ix.vX(i, k))] = cand.x(k)[i];
        for (int k = 0; k < td.nU; k++)
            for (int i = 0; i < NU; i++)
                x[size_t(ix.vU(i, k))] = cand.u(k)[i];
        double n1 = 0.;
        for (int k = 0; k + 1 < td.K; k++)
        {
            const double *A = &dd.A[size_t(k) * NX * NX], *B = &dd.B[size_t(k) * NX * NU], *C = &dd.C[size_t(k) * NX * NU];
            for (int i = 0; i < NX; i++)
            {
                double v = cand.x(k + 1)[i] - dd.z[size_t(k) * NX + i];
                for (int j = 0; j < NX; j++)
                    v -= A[i * NX + j] * cand.x(k)[j];
                for (int j = 0; j < NU; j++)
                    v -= B[i * NU + j] * cand.u(k)[j] + (td.interpolatedInput() ? C[i * NU + j] * cand.u(k + 1)[j] : 0.);
                x[size_t(ix.vNu(i, k))] = v;
                x[size_t(ix.vNuB(i, k))] = std::fabs(v);
                n1 += std::fabs(v);
            }
        }
        x[size_t(ix.norm1_nu)] = n1;
        out.norm1_nu = n1;
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
                out.lit_norm1_nu = r.x[size_t(ix.norm1_nu)];
                TrajectoryData sol = td;
                for (int k = 0; k < td.K; k++)
                    for (int i = 0; i < NX; i++)
                        sol.x(k)[i] = r.x[size_t(ix.vX(i, k))];
                for (int k = 0; k < td.nU; k++)
                    for (int i = 0; i < NU; i++)
                        sol.u(k)[i] = r.x[size_t(ix.vU(i, k))];
                if (nondimensionalize)
                    model->redimensionalizeTrajectory(sol);
                out.Xlit = sol.X;
                out.Ulit = sol.U;
            }
        }
        trust_region = keep_radius;
        if (nondimensionalize)
            model->redimensionalize();
        return out;
    }

    int last_dims[5] = {0, 0, 0, 0, 0};
    RQStructuredSocp twin;
};

} // namespace oracle
