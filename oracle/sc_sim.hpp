// ORACLE (test infrastructure only -- never linked or imported by the product).
// CPU restatement of the reference's receding-horizon driver:
//   main() of SC_sim                 scpp/src/SC_sim.cpp:19-104
//   scpp::interpolatedInput          scpp/src/commonFunctions.cpp:6-19
// Parity status: unpinned like the rest of the SC path (the reference cannot be built here and ships no outputs).
#pragma once
#include <cmath>
#include <vector>
#include "discretization.hpp"
#include "sc.hpp"

namespace oracle
{

// commonFunctions.cpp:6-19
template <int NU>
inline void interpolatedInput(const std::vector<double> &U, int K, double t, double total_time, bool first_order_hold, double *u)
{
    const double time_step = total_time / double(K - 1);
    size_t i = size_t(t / time_step);
    if (i > size_t(K - 2))
        i = size_t(K - 2);
    const double *u0 = &U[i * NU];
    const double *u1 = first_order_hold ? &U[(i + 1) * NU] : u0;
    const double t_intermediate = std::fmod(t, time_step) / time_step;
    for (int j = 0; j < NU; j++)
        u[j] = u0[j] + (u1[j] - u0[j]) * t_intermediate;
}

struct SimResult
{
    std::vector<double> X_sim, U_sim; // [steps][NX], [steps][NU]
    std::vector<double> t_plan;       // planned final time of every solve
    std::vector<int> sc_iterations;
    int steps = 0;                    // simulate() calls made
    int reached_end = 0;
    int solver_failed = 0;
};

// SC_sim.cpp:28-66.  `x` aliases model->p.x_init exactly as in the reference (SC_sim.cpp:36): the plant state IS
// the next solve's initial-state constraint.
template <class M>
SimResult scSim(SCAlgorithm<M> &solver, double time_step, int max_steps)
{
    constexpr int NX = M::NX, NU = M::NU;
    SimResult r;
    M *model = solver.model;
    double *x = model->p.x_init;
    int sim_step = 0;
    while (sim_step < max_steps)
    {
        const bool warm_start = sim_step > 0;
        solver.solve(warm_start);
        if (solver.solver_failed)
        {
            r.solver_failed = 1;
            break;
        }
        const TrajectoryData &td = solver.td; // getSolution
        const int K = td.K;
        double u0[NU], u1[NU];
        for (int j = 0; j < NU; j++)
            u0[j] = td.U[j];
        const bool first_order_hold = td.interpolatedInput();
        interpolatedInput<NU>(td.U, K, time_step, td.t, first_order_hold, u1);
        simulate(*model, time_step, u0, u1, x);
        r.X_sim.insert(r.X_sim.end(), x, x + NX);
        r.U_sim.insert(r.U_sim.end(), u0, u0 + NU);
        r.t_plan.push_back(td.t);
        r.sc_iterations.push_back(solver.iterations);
        r.steps++;
        double d2 = 0.;
        for (int j = 0; j < NX; j++)
            d2 += (x[j] - model->p.x_final[j]) * (x[j] - model->p.x_final[j]);
        const bool reached_end = std::sqrt(d2) < 0.02 || td.t < 0.25;
        if (reached_end)
        {
            r.reached_end = 1;
            break;
        }
        sim_step++;
    }
    return r;
}

} // namespace oracle
