// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the reference multiple-shooting discretization:
//   ODE<FOH,VT>::operator()          scpp_core/include/discretizationImplementation.hpp:38-120
//   multipleShootingImplementation   scpp_core/include/discretizationImplementation.hpp:122-181
//   multipleShooting dispatcher      scpp_core/src/discretization.cpp:42-55
//   simulate                         scpp_core/src/simulation.cpp:25-42
// Follows the reference formulation literally: augmented state V = [x | Phi | V_B | V_C | V_s | V_z],
// Phi^-1 by partial-pivot LU each RHS evaluation (Eigen's fixed-size inverse), RKF78 x 5 steps,
// post-multiplication by Phi(dt).
// Parity status: PINNED by tests/golden/rocketquat_dd_K15/K50.npz (scipy DOP853, rtol 1e-13, independent ODE formulation).
#pragma once
#include <cmath>
#include <vector>

#include "models.hpp"
#include "rkf78.hpp"

namespace oracle
{

// dense inverse by LU with partial pivoting; a, inv row-major n x n
inline void invertMatrix(int n, const double *a, double *inv)
{
    std::vector<double> lu(a, a + n * n);
    std::vector<int> piv(n);
    for (int i = 0; i < n; i++)
        piv[i] = i;
    for (int k = 0; k < n; k++)
    {
        int pr = k;
        double best = std::fabs(lu[k * n + k]);
        for (int i = k + 1; i < n; i++)
            if (std::fabs(lu[i * n + k]) > best)
            {
                best = std::fabs(lu[i * n + k]);
                pr = i;
            }
        if (pr != k)
        {
            for (int j = 0; j < n; j++)
                std::swap(lu[k * n + j], lu[pr * n + j]);
            std::swap(piv[k], piv[pr]);
        }
        for (int i = k + 1; i < n; i++)
        {
            lu[i * n + k] /= lu[k * n + k];
            const double f = lu[i * n + k];
            for (int j = k + 1; j < n; j++)
                lu[i * n + j] -= f * lu[k * n + j];
        }
    }
    for (int c = 0; c < n; c++)
    {
        std::vector<double> y(n);
        for (int i = 0; i < n; i++)
        {
            double v = (piv[i] == c) ? 1. : 0.;
            for (int j = 0; j < i; j++)
                v -= lu[i * n + j] * y[j];
            y[i] = v;
        }
        for (int i = n - 1; i >= 0; i--)
        {
            double v = y[i];
            for (int j = i + 1; j < n; j++)
                v -= lu[i * n + j] * inv[j * n + c];
            inv[i * n + c] = v / lu[i * n + i];
        }
    }
}

template <class Model>
void multipleShooting(const Model &model, const TrajectoryData &td, DiscretizationData &dd)
{
    constexpr int NX = Model::NX, NU = Model::NU;
    const bool FOH = dd.interpolatedInput(), VT = dd.variableTime();
    const int K = td.K;
    const int ncols = 1 + NX + NU + (FOH ? NU : 0) + (VT ? 1 : 0) + 1;

    double dt = 1. / double(K - 1);
    if (!VT)
        dt *= td.t;
    const double time = td.t;

    for (int k = 0; k < K - 1; k++)
    {
        // V column-major: V[c*NX + r]
        std::vector<double> V(size_t(NX) * ncols, 0.);
        for (int i = 0; i < NX; i++)
        {
            V[i] = td.x(k)[i];
            V[(1 + i) * NX + i] = 1.;
        }
        const double *u0 = td.u(k);
        const double *u1 = FOH ? td.u(k + 1) : u0;

        auto ode = [&](const std::vector<double> &Vc, std::vector<double> &dV, const double t) {
            const double *x = &Vc[0];
            double u[NU];
            for (int i = 0; i < NU; i++)
                u[i] = FOH ? u0[i] + t / dt * (u1[i] - u0[i]) : u0[i];
            double f[NX], A[NX * NX], B[NX * NU];
            model.computef(x, u, f);
            model.computeJacobians(x, u, A, B);
            if (VT)
            {
                for (double &v : A)
                    v *= time;
                for (double &v : B)
                    v *= time;
            }
            double Phi[NX * NX], Pinv[NX * NX];
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NX; j++)
                    Phi[i * NX + j] = Vc[(1 + j) * NX + i];
            invertMatrix(NX, Phi, Pinv);

            int cols = 0;
            for (int i = 0; i < NX; i++)
                dV[i] = VT ? time * f[i] : f[i];
            cols += 1;
            // A * Phi
            for (int j = 0; j < NX; j++)
                for (int i = 0; i < NX; i++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NX; q++)
                        acc += A[i * NX + q] * Phi[q * NX + j];
                    dV[(cols + j) * NX + i] = acc;
                }
            cols += NX;
            // Phi^-1 B
            double PB[NX * NU];
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NU; j++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NX; q++)
                        acc += Pinv[i * NX + q] * B[q * NU + j];
                    PB[i * NU + j] = acc;
                }
            if (FOH)
            {
                const double alpha = (dt - t) / dt;
                for (int j = 0; j < NU; j++)
                    for (int i = 0; i < NX; i++)
                        dV[(cols + j) * NX + i] = PB[i * NU + j] * alpha;
                cols += NU;
                const double beta = t / dt;
                for (int j = 0; j < NU; j++)
                    for (int i = 0; i < NX; i++)
                        dV[(cols + j) * NX + i] = PB[i * NU + j] * beta;
                cols += NU;
            }
            else
            {
                for (int j = 0; j < NU; j++)
                    for (int i = 0; i < NX; i++)
                        dV[(cols + j) * NX + i] = PB[i * NU + j];
                cols += NU;
            }
            // -A x - B u (with the sigma-scaled A,B when VT)
            double r[NX];
            for (int i = 0; i < NX; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NX; q++)
                    acc -= A[i * NX + q] * x[q];
                for (int q = 0; q < NU; q++)
                    acc -= B[i * NU + q] * u[q];
                r[i] = acc;
            }
            if (VT)
            {
                for (int i = 0; i < NX; i++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NX; q++)
                        acc += Pinv[i * NX + q] * f[q];
                    dV[cols * NX + i] = acc;
                }
                cols += 1;
                for (int i = 0; i < NX; i++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NX; q++)
                        acc += Pinv[i * NX + q] * r[q];
                    dV[cols * NX + i] = acc;
                }
                cols += 1;
            }
            else
            {
                for (int i = 0; i < NX; i++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NX; q++)
                        acc += Pinv[i * NX + q] * (f[q] + r[q]);
                    dV[cols * NX + i] = acc;
                }
                cols += 1;
            }
        };

        integrateRKF78(ode, V, dt, 5);

        double *Ak = &dd.A[size_t(k) * NX * NX];
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NX; j++)
                Ak[i * NX + j] = V[(1 + j) * NX + i];
        int cols = 1 + NX;
        auto mulPhi = [&](int ncol, double *out, int ld) {
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < ncol; j++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NX; q++)
                        acc += Ak[i * NX + q] * V[(cols + j) * NX + q];
                    out[i * ld + j] = acc;
                }
            cols += ncol;
        };
        mulPhi(NU, &dd.B[size_t(k) * NX * NU], NU);
        if (FOH)
            mulPhi(NU, &dd.C[size_t(k) * NX * NU], NU);
        if (VT)
            mulPhi(1, &dd.s[size_t(k) * NX], 1);
        mulPhi(1, &dd.z[size_t(k) * NX], 1);
    }
}

// simulation.cpp:25-42
template <class Model>
void simulate(const Model &model, double dt, const double *u0, const double *u1, double *x)
{
    constexpr int NX = Model::NX, NU = Model::NU;
    std::vector<double> y(x, x + NX);
    auto ode = [&](const std::vector<double> &xs, std::vector<double> &dx, const double t) {
        double u[NU];
        for (int i = 0; i < NU; i++)
            u[i] = u0[i] + t / dt * (u1[i] - u0[i]);
        model.computef(&xs[0], u, &dx[0]);
    };
    integrateRKF78(ode, y, dt, 20);
    for (int i = 0; i < NX; i++)
        x[i] = y[i];
}

} // namespace oracle
