// ORACLE (test infrastructure, NOT product code).
// Runge-Kutta-Fehlberg 7(8), 13 stages (Fehlberg, NASA TR R-287, 1968), propagated with the
// 8th-order weights -- the stepper the reference uses:
//   boost::numeric::odeint::runge_kutta_fehlberg78  (Boost absent here; tableau restated from
//   the published paper, unit-tested against the order conditions in tests/test_oracle_rkf78.py)
// driven like integrate_adaptive(stepper, ode, V, 0., dt, dt/N) with a NON-controlled stepper,
// i.e. N fixed steps of dt/N:
//   scpp_core/include/discretizationImplementation.hpp:141,154 (N=5)
//   scpp_core/src/simulation.cpp:37,41 (N=20)
// Parity status: PINNED by the Fehlberg 7(8) order conditions and an 8th-order convergence test (tests/test_oracle_rkf78.py);
// Boost.Odeint itself is absent.
#pragma once
#include <vector>

namespace oracle
{

struct RKF78Tableau
{
    static constexpr int S = 13;
    double c[S];
    double a[S][S];
    double b[S];
    RKF78Tableau()
    {
        for (int i = 0; i < S; i++)
        {
            c[i] = 0.;
            b[i] = 0.;
            for (int j = 0; j < S; j++)
                a[i][j] = 0.;
        }
        c[1] = 2. / 27.;
        c[2] = 1. / 9.;
        c[3] = 1. / 6.;
        c[4] = 5. / 12.;
        c[5] = 1. / 2.;
        c[6] = 5. / 6.;
        c[7] = 1. / 6.;
        c[8] = 2. / 3.;
        c[9] = 1. / 3.;
        c[10] = 1.;
        c[11] = 0.;
        c[12] = 1.;
        a[1][0] = 2. / 27.;
        a[2][0] = 1. / 36.;
        a[2][1] = 1. / 12.;
        a[3][0] = 1. / 24.;
        a[3][2] = 1. / 8.;
        a[4][0] = 5. / 12.;
        a[4][2] = -25. / 16.;
        a[4][3] = 25. / 16.;
        a[5][0] = 1. / 20.;
        a[5][3] = 1. / 4.;
        a[5][4] = 1. / 5.;
        a[6][0] = -25. / 108.;
        a[6][3] = 125. / 108.;
        a[6][4] = -65. / 27.;
        a[6][5] = 125. / 54.;
        a[7][0] = 31. / 300.;
        a[7][4] = 61. / 225.;
        a[7][5] = -2. / 9.;
        a[7][6] = 13. / 900.;
        a[8][0] = 2.;
        a[8][3] = -53. / 6.;
        a[8][4] = 704. / 45.;
        a[8][5] = -107. / 9.;
        a[8][6] = 67. / 90.;
        a[8][7] = 3.;
        a[9][0] = -91. / 108.;
        a[9][3] = 23. / 108.;
        a[9][4] = -976. / 135.;
        a[9][5] = 311. / 54.;
        a[9][6] = -19. / 60.;
        a[9][7] = 17. / 6.;
        a[9][8] = -1. / 12.;
        a[10][0] = 2383. / 4100.;
        a[10][3] = -341. / 164.;
        a[10][4] = 4496. / 1025.;
        a[10][5] = -301. / 82.;
        a[10][6] = 2133. / 4100.;
        a[10][7] = 45. / 82.;
        a[10][8] = 45. / 164.;
        a[10][9] = 18. / 41.;
        a[11][0] = 3. / 205.;
        a[11][5] = -6. / 41.;
        a[11][6] = -3. / 205.;
        a[11][7] = -3. / 41.;
        a[11][8] = 3. / 41.;
        a[11][9] = 6. / 41.;
        a[12][0] = -1777. / 4100.;
        a[12][3] = -341. / 164.;
        a[12][4] = 4496. / 1025.;
        a[12][5] = -289. / 82.;
        a[12][6] = 2193. / 4100.;
        a[12][7] = 51. / 82.;
        a[12][8] = 33. / 164.;
        a[12][9] = 12. / 41.;
        a[12][11] = 1.;
        // 8th-order weights
        b[5] = 34. / 105.;
        b[6] = 9. / 35.;
        b[7] = 9. / 35.;
        b[8] = 9. / 280.;
        b[9] = 9. / 280.;
        b[11] = 41. / 840.;
        b[12] = 41. / 840.;
    }
};

inline const RKF78Tableau &rkf78()
{
    static const RKF78Tableau t;
    return t;
}

// N fixed RKF78 steps over [0, dt]; ode(y, dydt, t)
template <class ODE>
void integrateRKF78(ODE &ode, std::vector<double> &y, double dt, int N)
{
    const RKF78Tableau &T = rkf78();
    const size_t n = y.size();
    std::vector<std::vector<double>> k(RKF78Tableau::S, std::vector<double>(n));
    std::vector<double> ys(n);
    const double h = dt / double(N);
    for (int step = 0; step < N; step++)
    {
        const double t0 = double(step) * h;
        for (int s = 0; s < RKF78Tableau::S; s++)
        {
            for (size_t i = 0; i < n; i++)
            {
                double acc = 0.;
                for (int j = 0; j < s; j++)
                    if (T.a[s][j] != 0.)
                        acc += T.a[s][j] * k[j][i];
                ys[i] = y[i] + h * acc;
            }
            ode(ys, k[s], t0 + T.c[s] * h);
        }
        for (size_t i = 0; i < n; i++)
        {
            double acc = 0.;
            for (int s = 0; s < RKF78Tableau::S; s++)
                if (T.b[s] != 0.)
                    acc += T.b[s] * k[s][i];
            y[i] += h * acc;
        }
    }
}

} // namespace oracle
