// Second-order-cone / Nesterov-Todd scaling primitives for the batched interior-point kernel.
// Conventions (ECOS / CVXOPT): cone Q^d = {(s0,s1): s0 >= ||s1||}, J = diag(1,-1,..,-1),
//   wbar = (sbar + J zbar) / (2 gamma), gamma = sqrt((1 + sbar'zbar)/2), eta = (||s||_J / ||z||_J)^(1/2)
//   W = eta [ w0 , w1' ; w1 , I + w1 w1'/(1+w0) ] ,  W z = W^-1 s = lambda ,  W^-2 = (2 vt vt' - J)/eta^2, vt = J wbar.
// All functions work on small per-lane arrays (pointers may be global memory).
#pragma once
#include "common.h"

namespace scpp
{
namespace cone
{

// returns false if s or z left the cone interior
template <class AS, class AZ, class AW>
__device__ inline bool nt_scaling(AS s, AZ z, int d, double &eta, AW w)
{
    double s1 = 0., z1 = 0.;
    for (int i = 1; i < d; i++)
    {
        s1 += s[i] * s[i];
        z1 += z[i] * z[i];
    }
    const double sres = s[0] * s[0] - s1, zres = z[0] * z[0] - z1;
    if (!(sres > 0.) || !(zres > 0.))
        return false;
    const double sn = sqrt(sres), zn = sqrt(zres);
    double sz = 0.;
    for (int i = 0; i < d; i++)
        sz += (s[i] / sn) * (z[i] / zn);
    const double gamma = sqrt(0.5 * (1. + sz));
    const double a = 0.5 / gamma;
    w[0] = a * (s[0] / sn + z[0] / zn);
    for (int i = 1; i < d; i++)
        w[i] = a * (s[i] / sn - z[i] / zn);
    eta = sqrt(sn / zn);
    return true;
}
template <class AW, class AV, class AO>
__device__ inline void applyW(double eta, AW w, int d, AV v, AO out)
{
    double zeta = 0.;
    for (int i = 1; i < d; i++)
        zeta += w[i] * v[i];
    const double v0 = v[0];
    const double f = v0 + zeta / (1. + w[0]);
    for (int i = 1; i < d; i++)
        out[i] = eta * (v[i] + f * w[i]);
    out[0] = eta * (w[0] * v0 + zeta);
}
template <class AW, class AV, class AO>
__device__ inline void applyWinv(double eta, AW w, int d, AV v, AO out)
{
    double zeta = 0.;
    for (int i = 1; i < d; i++)
        zeta += w[i] * v[i];
    const double v0 = v[0];
    const double f = -v0 + zeta / (1. + w[0]);
    for (int i = 1; i < d; i++)
        out[i] = (v[i] + f * w[i]) / eta;
    out[0] = (w[0] * v0 - zeta) / eta;
}
template <class AW, class AV, class AO>
__device__ inline void applyWinv2(double eta, AW w, int d, AV v, AO out)
{
    double tv = w[0] * v[0];
    for (int i = 1; i < d; i++)
        tv -= w[i] * v[i];
    const double e2 = 1. / (eta * eta);
    const double v0 = v[0];
    for (int i = 1; i < d; i++)
        out[i] = e2 * (-2. * w[i] * tv + v[i]);
    out[0] = e2 * (2. * w[0] * tv - v0);
}
// out = u o v   (out may not alias u or v)
template <class AU, class AV, class AO>
__device__ inline void conicProduct(int d, AU u, AV v, AO out)
{
    double s0 = 0.;
    for (int i = 0; i < d; i++)
        s0 += u[i] * v[i];
    for (int i = 1; i < d; i++)
        out[i] = u[0] * v[i] + v[0] * u[i];
    out[0] = s0;
}
// solve lam o out = dd  (out may alias dd)
template <class AL, class AD, class AO>
__device__ inline void conicDivision(int d, AL lam, AD dd, AO out)
{
    double l1d1 = 0., l1l1 = 0.;
    for (int i = 1; i < d; i++)
    {
        l1d1 += lam[i] * dd[i];
        l1l1 += lam[i] * lam[i];
    }
    const double rho = lam[0] * lam[0] - l1l1;
    const double u0 = (lam[0] * dd[0] - l1d1) / rho;
    for (int i = 1; i < d; i++)
        out[i] = (dd[i] - u0 * lam[i]) / lam[0];
    out[0] = u0;
}
// 1/alpha_max such that lam + alpha v stays in the cone (ECOS lineSearch)
template <class AL, class AV>
__device__ inline double stepInv(int d, AL lam, AV v)
{
    double l1 = 0.;
    for (int i = 1; i < d; i++)
        l1 += lam[i] * lam[i];
    const double ln = sqrt(lam[0] * lam[0] - l1);
    double lbJv = lam[0] * v[0];
    for (int i = 1; i < d; i++)
        lbJv -= lam[i] * v[i];
    lbJv /= ln;
    const double rho0 = lbJv / ln;
    const double f = (lbJv + v[0]) / (lam[0] / ln + 1.);
    double r1 = 0.;
    for (int i = 1; i < d; i++)
    {
        const double ri = (v[i] - f * lam[i] / ln) / ln;
        r1 += ri * ri;
    }
    return sqrt(r1) - rho0;
}

// ---- compile-time-dimension versions on register arrays (fully unrolled: all loads of a cone are issued
// together, no scratch indexing) ----
template <int D>
__device__ inline bool nt_scalingS(const double (&s)[D], const double (&z)[D], double &eta, double (&w)[D])
{
    double s1 = 0., z1 = 0.;
#pragma unroll
    for (int i = 1; i < D; i++)
    {
        s1 += s[i] * s[i];
        z1 += z[i] * z[i];
    }
    const double sres = s[0] * s[0] - s1, zres = z[0] * z[0] - z1;
    if (!(sres > 0.) || !(zres > 0.))
        return false;
    const double sn = sqrt(sres), zn = sqrt(zres);
    double sz = 0.;
#pragma unroll
    for (int i = 0; i < D; i++)
        sz += (s[i] / sn) * (z[i] / zn);
    const double gamma = sqrt(0.5 * (1. + sz));
    const double a = 0.5 / gamma;
    w[0] = a * (s[0] / sn + z[0] / zn);
#pragma unroll
    for (int i = 1; i < D; i++)
        w[i] = a * (s[i] / sn - z[i] / zn);
    eta = sqrt(sn / zn);
    return true;
}
template <int D>
__device__ inline void applyWS(double eta, const double (&w)[D], const double (&v)[D], double (&out)[D])
{
    double zeta = 0.;
#pragma unroll
    for (int i = 1; i < D; i++)
        zeta += w[i] * v[i];
    const double v0 = v[0];
    const double f = v0 + zeta / (1. + w[0]);
#pragma unroll
    for (int i = 1; i < D; i++)
        out[i] = eta * (v[i] + f * w[i]);
    out[0] = eta * (w[0] * v0 + zeta);
}
template <int D>
__device__ inline void applyWinvS(double eta, const double (&w)[D], const double (&v)[D], double (&out)[D])
{
    double zeta = 0.;
#pragma unroll
    for (int i = 1; i < D; i++)
        zeta += w[i] * v[i];
    const double v0 = v[0];
    const double f = -v0 + zeta / (1. + w[0]);
#pragma unroll
    for (int i = 1; i < D; i++)
        out[i] = (v[i] + f * w[i]) / eta;
    out[0] = (w[0] * v0 - zeta) / eta;
}
template <int D>
__device__ inline void applyWinv2S(double eta, const double (&w)[D], const double (&v)[D], double (&out)[D])
{
    double tv = w[0] * v[0];
#pragma unroll
    for (int i = 1; i < D; i++)
        tv -= w[i] * v[i];
    const double e2 = 1. / (eta * eta);
    const double v0 = v[0];
#pragma unroll
    for (int i = 1; i < D; i++)
        out[i] = e2 * (-2. * w[i] * tv + v[i]);
    out[0] = e2 * (2. * w[0] * tv - v0);
}
template <int D>
__device__ inline void conicProductS(const double (&u)[D], const double (&v)[D], double (&out)[D])
{
    double s0 = 0.;
#pragma unroll
    for (int i = 0; i < D; i++)
        s0 += u[i] * v[i];
#pragma unroll
    for (int i = 1; i < D; i++)
        out[i] = u[0] * v[i] + v[0] * u[i];
    out[0] = s0;
}
template <int D>
__device__ inline void conicDivisionS(const double (&lam)[D], double (&dd)[D])
{
    double l1d1 = 0., l1l1 = 0.;
#pragma unroll
    for (int i = 1; i < D; i++)
    {
        l1d1 += lam[i] * dd[i];
        l1l1 += lam[i] * lam[i];
    }
    const double rho = lam[0] * lam[0] - l1l1;
    const double u0 = (lam[0] * dd[0] - l1d1) / rho;
#pragma unroll
    for (int i = 1; i < D; i++)
        dd[i] = (dd[i] - u0 * lam[i]) / lam[0];
    dd[0] = u0;
}
template <int D>
__device__ inline double stepInvS(const double (&lam)[D], const double (&v)[D])
{
    double l1 = 0.;
#pragma unroll
    for (int i = 1; i < D; i++)
        l1 += lam[i] * lam[i];
    const double ln = sqrt(lam[0] * lam[0] - l1);
    double lbJv = lam[0] * v[0];
#pragma unroll
    for (int i = 1; i < D; i++)
        lbJv -= lam[i] * v[i];
    lbJv /= ln;
    const double rho0 = lbJv / ln;
    const double f = (lbJv + v[0]) / (lam[0] / ln + 1.);
    double r1 = 0.;
#pragma unroll
    for (int i = 1; i < D; i++)
    {
        const double ri = (v[i] - f * lam[i] / ln) / ln;
        r1 += ri * ri;
    }
    return sqrt(r1) - rho0;
}

} // namespace cone
} // namespace scpp
