// ORACLE (test infrastructure, NOT product code).
// Forward-mode dual numbers: stand-in for the CppAD tape + CppADCodeGen Jacobian of
// scpp_core/include/systemDynamics.hpp:109-168,206-235 (exact derivatives, so any
// correct AD agrees to round-off).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may link this.
#pragma once
#include <cmath>

namespace oracle
{

template <int N>
struct Dual
{
    double v;
    double d[N];

    Dual() : v(0.)
    {
        for (int i = 0; i < N; i++)
            d[i] = 0.;
    }
    Dual(double c) : v(c)
    {
        for (int i = 0; i < N; i++)
            d[i] = 0.;
    }
};

template <int N>
Dual<N> operator+(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    r.v = a.v + b.v;
    for (int i = 0; i < N; i++)
        r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int N>
Dual<N> operator-(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    r.v = a.v - b.v;
    for (int i = 0; i < N; i++)
        r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int N>
Dual<N> operator-(const Dual<N> &a)
{
    Dual<N> r;
    r.v = -a.v;
    for (int i = 0; i < N; i++)
        r.d[i] = -a.d[i];
    return r;
}
template <int N>
Dual<N> operator*(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    r.v = a.v * b.v;
    for (int i = 0; i < N; i++)
        r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <int N>
Dual<N> operator/(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    const double inv = 1. / b.v;
    r.v = a.v * inv;
    for (int i = 0; i < N; i++)
        r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N>
Dual<N> operator+(const Dual<N> &a, double b) { return a + Dual<N>(b); }
template <int N>
Dual<N> operator+(double a, const Dual<N> &b) { return Dual<N>(a) + b; }
template <int N>
Dual<N> operator-(const Dual<N> &a, double b) { return a - Dual<N>(b); }
template <int N>
Dual<N> operator-(double a, const Dual<N> &b) { return Dual<N>(a) - b; }
template <int N>
Dual<N> operator*(const Dual<N> &a, double b) { return a * Dual<N>(b); }
template <int N>
Dual<N> operator*(double a, const Dual<N> &b) { return Dual<N>(a) * b; }
template <int N>
Dual<N> operator/(const Dual<N> &a, double b) { return a / Dual<N>(b); }
template <int N>
Dual<N> operator/(double a, const Dual<N> &b) { return Dual<N>(a) / b; }

template <int N>
Dual<N> sqrt(const Dual<N> &a)
{
    Dual<N> r;
    r.v = std::sqrt(a.v);
    const double s = 0.5 / r.v;
    for (int i = 0; i < N; i++)
        r.d[i] = a.d[i] * s;
    return r;
}
template <int N>
Dual<N> sin(const Dual<N> &a)
{
    Dual<N> r;
    r.v = std::sin(a.v);
    const double c = std::cos(a.v);
    for (int i = 0; i < N; i++)
        r.d[i] = a.d[i] * c;
    return r;
}
template <int N>
Dual<N> cos(const Dual<N> &a)
{
    Dual<N> r;
    r.v = std::cos(a.v);
    const double s = -std::sin(a.v);
    for (int i = 0; i < N; i++)
        r.d[i] = a.d[i] * s;
    return r;
}

inline double sqrt(double a) { return std::sqrt(a); }
inline double sin(double a) { return std::sin(a); }
inline double cos(double a) { return std::cos(a); }

} // namespace oracle
