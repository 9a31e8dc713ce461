// Batched multiple-shooting discretization: replaces
//   scpp::discretization::multipleShooting   scpp_core/src/discretization.cpp:42-55
//   multipleShootingImplementation<FOH,VT>   scpp_core/include/discretizationImplementation.hpp:122-181
//   ODE<FOH,VT>::operator()                  scpp_core/include/discretizationImplementation.hpp:38-120
// for ALL K-1 segments of B problem instances in one launch.
//
// Mapping (gfx950): one 64-lane wavefront (= one workgroup) per (instance, segment). The augmented
// state V = [x | Phi | Psi_B | Psi_C | psi_s | psi_z] (NX x NCOLS, 14x25 = 350 entries for RocketQuat)
// is spread over the lanes (<= 6 entries per lane, register resident incl. the 13 RKF78 stage
// derivatives); stage values and the per-stage Jacobian tile [sigma*A | sigma*B] (NX x (NX+NU)) are
// staged in LDS.  The Jacobian is produced SIMT-style by forward-mode AD: lane j < NX+NU evaluates the
// model plugin's systemFlowMap<Dual1> with seed e_j (one Jacobian column per lane, no divergence).
//
// Integrator: the reference's RKF78 with 5 fixed steps per segment, applied to the
// forward-sensitivity form  Psi' = A Psi + forcing  (algebraically identical to the reference's
// Phi(dt) * int Phi^-1 ... formulation, SURVEY.md §7, but needs no 14x14 inverse per RHS evaluation):
//   Phi'   = A Phi                  Phi(0) = I          -> A_k = Phi(dt)
//   Psi_B' = A Psi_B + B (dt-t)/dt  Psi_B(0) = 0        -> B_k
//   Psi_C' = A Psi_C + B t/dt                           -> C_k
//   psi_s' = A psi_s + f                                 -> s_k      (A,B scaled by sigma; f unscaled)
//   psi_z' = A psi_z - A x - B u                         -> z_k      (fixed time: + f)
// Algorithmic HBM traffic per instance-call (RocketQuat, K=50): read 7,288 B, write 131,712 B.
#pragma once
#include "common.h"
#include <utility>

namespace scpp
{

// f(integral_constant<int,0>) ... f(integral_constant<int,RK_S-1>) : compile-time loop over the RK stages
template <class F, int... S>
__device__ inline void forEachStageImpl(F &&f, std::integer_sequence<int, S...>)
{
    (f(std::integral_constant<int, S>{}), ...);
}
template <class F>
__device__ inline void forEachStage(F &&f)
{
    forEachStageImpl(f, std::make_integer_sequence<int, RK_S>{});
}

#ifndef DISC_WAVES_PER_SIMD
#define DISC_WAVES_PER_SIMD 2
#endif

template <class Model, bool FOH, bool VT>
struct DiscLayout
{
    static constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP;
    static constexpr int NJ = NX + NU;                                  // Jacobian columns
    static constexpr int NCOLS = 1 + NX + NU + (FOH ? NU : 0) + (VT ? 1 : 0) + 1;
    static constexpr int NENT = NX * NCOLS;
    static constexpr int EPL = (NENT + WAVE - 1) / WAVE;                // entries per lane
    static constexpr int COL_PHI = 1, COL_B = 1 + NX, COL_C = COL_B + NU;
    static constexpr int COL_S = COL_C + (FOH ? NU : 0);
    static constexpr int COL_Z = COL_S + (VT ? 1 : 0);
};

// X [B][K][NX], U [B][K][NU] (FOH) , sigma [B], par [B][NP]  ->  A [B][K-1][NX][NX], Bm, C [B][K-1][NX][NU],
// S, Z [B][K-1][NX]   (row-major blocks).  active[B] (may be null): skip instances with active == 0.
template <class Model, bool FOH, bool VT>
__global__ void __launch_bounds__(WAVE, DISC_WAVES_PER_SIMD)
    discretize_kernel(int B, int K, const double *__restrict__ X, const double *__restrict__ U,
                      const double *__restrict__ sigma, const double *__restrict__ par, int par_stride,
                      const int *__restrict__ active,
                      double *__restrict__ Aout, double *__restrict__ Bout, double *__restrict__ Cout,
                      double *__restrict__ Sout, double *__restrict__ Zout)
{
    using L = DiscLayout<Model, FOH, VT>;
    constexpr int NX = L::NX, NU = L::NU, NP = L::NP, NJ = L::NJ, NCOLS = L::NCOLS, NENT = L::NENT, EPL = L::EPL;

    __shared__ double Ys[NENT];      // stage values of V (column-major: col*NX + row)
    __shared__ double Jm[NX * NJ];   // [sigma*A | sigma*B] row-major
    __shared__ double fv[NX];        // f(x,u) (unscaled)

    const int lane = threadIdx.x;
    // XCD-aware block -> (instance, segment) map: blocks b, b+8, b+16.. share an XCD (and its L2), so
    // give one XCD all K-1 segments of an instance (they re-read the same X/U/par lines).
    const int nseg = K - 1;
    const long b = blockIdx.x;
    const long xcd = b & 7, g = b >> 3;
    const long inst = (g / nseg) * 8 + xcd;
    const int k = int(g % nseg);
    if (inst >= B)
        return;
    if (active && active[inst] == 0)
        return;

    const double sg = sigma[inst];
    const double dt = VT ? 1. / double(K - 1) : sg / double(K - 1);
    const double tscale = VT ? sg : 1.;
    double p[NP];
#pragma unroll
    for (int i = 0; i < NP; i++)
        p[i] = par[inst * par_stride + i];
    double u0[NU], u1[NU];
#pragma unroll
    for (int i = 0; i < NU; i++)
    {
        u0[i] = U[(inst * K + k) * NU + i];
        u1[i] = FOH ? U[(inst * K + k + 1) * NU + i] : u0[i];
    }

    // entries owned by this lane
    int erow[EPL], ecol[EPL];
    double y[EPL];
#pragma unroll
    for (int m = 0; m < EPL; m++)
    {
        const int e = lane + WAVE * m;
        const int ee = e < NENT ? e : 0;
        ecol[m] = ee / NX;
        erow[m] = ee - ecol[m] * NX;
        double v = 0.;
        if (e < NENT)
        {
            if (ecol[m] == 0)
                v = X[(inst * K + k) * NX + erow[m]];
            else if (ecol[m] >= L::COL_PHI && ecol[m] < L::COL_PHI + NX)
                v = (ecol[m] - L::COL_PHI == erow[m]) ? 1. : 0.;
        }
        y[m] = v;
    }

    double kk[RK_S][EPL];
    const double h = dt / 5.;

    for (int step = 0; step < 5; step++)
    {
        const double t0 = double(step) * h;
        // The 13 stages are instantiated with a COMPILE-TIME stage index: the tableau entries become
        // immediates, zero coefficients vanish and kk[][] stays in registers (a rolled stage loop indexes it
        // dynamically, i.e. from scratch memory, and tests the coefficients at run time).
        forEachStage([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // ---- stage value ys = y + h * sum_j a_sj k_j ----
            const double ts = t0 + RK_C[s] * h;
#pragma unroll
            for (int m = 0; m < EPL; m++)
            {
                double acc = 0.;
#pragma unroll
                for (int j = 0; j < s; j++)
                    if (RK_A[s][j] != 0.)
                        acc += RK_A[s][j] * kk[j][m];
                const double ys = y[m] + h * acc;
                const int e = lane + WAVE * m;
                if (e < NENT)
                    Ys[e] = ys;
            }
            WAVE_SYNC();
            // ---- Jacobian tile by forward-mode AD, one seed direction per lane ----
            const double frac = FOH ? ts / dt : 0.;
            if (lane < NJ)
            {
                Dual1 xd[NX], ud[NU], fd[NX];
#pragma unroll
                for (int i = 0; i < NX; i++)
                    xd[i] = Dual1(Ys[i], (lane == i) ? 1. : 0.);
#pragma unroll
                for (int i = 0; i < NU; i++)
                    ud[i] = Dual1(u0[i] + frac * (u1[i] - u0[i]), (lane == NX + i) ? 1. : 0.);
                Model::template systemFlowMap<Dual1>(xd, ud, p, fd);
#pragma unroll
                for (int i = 0; i < NX; i++)
                    Jm[i * NJ + lane] = tscale * fd[i].d;
                if (lane == 0)
                {
#pragma unroll
                    for (int i = 0; i < NX; i++)
                        fv[i] = fd[i].v;
                }
            }
            WAVE_SYNC();
            // ---- derivative of the owned entries ----
#pragma unroll
            for (int m = 0; m < EPL; m++)
            {
                const int r = erow[m], c = ecol[m];
                double d;
                if (c == 0)
                {
                    d = tscale * fv[r];
                }
                else
                {
                    double acc = 0.;
#pragma unroll
                    for (int j = 0; j < NX; j++)
                        acc += Jm[r * NJ + j] * Ys[c * NX + j];
                    if (c >= L::COL_B && c < L::COL_B + NU)
                    {
                        const double alpha = FOH ? (1. - frac) : 1.;
                        acc += Jm[r * NJ + NX + (c - L::COL_B)] * alpha;
                    }
                    else if (FOH && c >= L::COL_C && c < L::COL_C + NU)
                    {
                        acc += Jm[r * NJ + NX + (c - L::COL_C)] * frac;
                    }
                    else if (VT && c == L::COL_S)
                    {
                        acc += fv[r];
                    }
                    else if (c == L::COL_Z)
                    {
                        double q = VT ? 0. : fv[r];
#pragma unroll
                        for (int j = 0; j < NX; j++)
                            q -= Jm[r * NJ + j] * Ys[j];
#pragma unroll
                        for (int j = 0; j < NU; j++)
                            q -= Jm[r * NJ + NX + j] * (u0[j] + frac * (u1[j] - u0[j]));
                        acc += q;
                    }
                    d = acc;
                }
                kk[s][m] = d;
            }
            WAVE_SYNC();
        });
        // ---- y += h * sum_s b_s k_s ----
#pragma unroll
        for (int m = 0; m < EPL; m++)
        {
            double acc = 0.;
#pragma unroll
            for (int s = 0; s < RK_S; s++)
                if (RK_B[s] != 0.)
                    acc += RK_B[s] * kk[s][m];
            y[m] += h * acc;
        }
    }

    // ---- write A_k, B_k, C_k, s_k, z_k ----
    const long seg = inst * nseg + k;
#pragma unroll
    for (int m = 0; m < EPL; m++)
    {
        const int e = lane + WAVE * m;
        if (e >= NENT)
            continue;
        const int r = erow[m], c = ecol[m];
        if (c >= L::COL_PHI && c < L::COL_PHI + NX)
            Aout[seg * NX * NX + r * NX + (c - L::COL_PHI)] = y[m];
        else if (c >= L::COL_B && c < L::COL_B + NU)
            Bout[seg * NX * NU + r * NU + (c - L::COL_B)] = y[m];
        else if (FOH && c >= L::COL_C && c < L::COL_C + NU)
            Cout[seg * NX * NU + r * NU + (c - L::COL_C)] = y[m];
        else if (VT && c == L::COL_S)
            Sout[seg * NX + r] = y[m];
        else if (c == L::COL_Z)
            Zout[seg * NX + r] = y[m];
    }
}

// Batched nonlinear propagation  x <- x(dt)  under first-order-hold input: replaces
//   scpp::simulate  scpp_core/src/simulation.cpp:25-42   (RKF78, 20 fixed steps)
// One thread per instance (14 states + 13 stage slopes fit in registers).
template <class Model>
__global__ void simulate_kernel(int B, const double *__restrict__ par, int par_stride, const double *__restrict__ dtv,
                                const double *__restrict__ u0v, const double *__restrict__ u1v, double *__restrict__ x,
                                const int *__restrict__ active)
{
    constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP;
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= B)
        return;
    if (active && active[i] == 0)
        return;
    double p[NP], u0[NU], u1[NU], y[NX], kk[RK_S][NX];
    for (int j = 0; j < NP; j++)
        p[j] = par[i * par_stride + j];
    for (int j = 0; j < NU; j++)
    {
        u0[j] = u0v[i * NU + j];
        u1[j] = u1v[i * NU + j];
    }
    for (int j = 0; j < NX; j++)
        y[j] = x[i * NX + j];
    const double dt = dtv[i];
    const double h = dt / 20.;
    for (int step = 0; step < 20; step++)
    {
        const double t0 = double(step) * h;
#pragma unroll
        for (int s = 0; s < RK_S; s++)
        {
            double ys[NX], u[NU];
            const double ts = t0 + RK_C[s] * h;
            for (int j = 0; j < NX; j++)
            {
                double acc = 0.;
#pragma unroll
                for (int q = 0; q < s; q++)
                    if (RK_A[s][q] != 0.)
                        acc += RK_A[s][q] * kk[q][j];
                ys[j] = y[j] + h * acc;
            }
            for (int j = 0; j < NU; j++)
                u[j] = u0[j] + ts / dt * (u1[j] - u0[j]);
            Model::template systemFlowMap<double>(ys, u, p, kk[s]);
        }
        for (int j = 0; j < NX; j++)
        {
            double acc = 0.;
#pragma unroll
            for (int s = 0; s < RK_S; s++)
                if (RK_B[s] != 0.)
                    acc += RK_B[s] * kk[s][j];
            y[j] += h * acc;
        }
    }
    for (int j = 0; j < NX; j++)
        x[i * NX + j] = y[j];
}

} // namespace scpp
