// Model plugin: RocketQuat (6-DoF rocket, 14 states / 4 inputs / 10 flow-map parameters).
// Keeps the reference's plugin contract -- a scalar-templated systemFlowMap(x,u,par,f)
// (scpp_core/include/systemModel.hpp:64-140, scpp_models/src/rocketQuat.cpp:7-37) -- but as
// compile-time __host__ __device__ code so it inlines into the HIP kernels (the reference
// JIT-compiles a CppAD-generated C file instead: systemDynamics.hpp:132-144).
// Reference quirks kept on purpose: un-normalised quaternion rotation matrix
// (Eigen toRotationMatrix), gyroscopic term w x w == 0.
//
// ONE source of truth per model (round 6): this flow map is the only place a model's dynamics are written.  The analytic Jacobian rows / table the
// discretisation kernel uses (model_jacobian_rows.h) are GENERATED from it: tools/flowmap_symbolic.cpp instantiates systemFlowMap with a scalar type
// that records expressions (T = PT = Sym), tools/gen_model_jacobian.py parses what it prints with sympy, differentiates and emits the header -- the
// reference's CppAD tape -> CppADCodeGen -> C step (systemDynamics.hpp:109-168) at build time.  Hence the two template parameters: T = scalar of states,
// inputs and outputs (double, Dual1, Sym), PT = scalar of the parameters (double everywhere on the device, Sym in the generator).
#pragma once
#include "common.h"
#ifndef SCPP_FLOWMAP_ONLY // (the generator's host tool compiles the flow maps without the header it is about to generate)
#include "model_jacobian_rows.h"
#else
namespace scpp
{
struct RocketQuatJacobianRows;
struct RocketQuatJacobianTable;
struct Rocket2dJacobianRows;
struct Rocket2dJacobianTable;
} // namespace scpp
#endif

namespace scpp
{

struct RocketQuatModel
{
    static constexpr int NX = 14, NU = 4, NP = 10;
    static constexpr int MODEL_ID = 0;
    // optional: analytic rows of [df/dx | df/du], generated from this flow map by tools/gen_model_jacobian.py (the
    // reference's CppADCodeGen step at build time); kernels fall back to forward-mode AD of systemFlowMap<Dual1> without it
    using JacobianRows = RocketQuatJacobianRows;
    using JacobianTable = RocketQuatJacobianTable; // the same, one output per lane (discretize_kernel)

    // par = [alpha_m, g_I(3), J_B(3), r_T_B(3)]   rocketQuat.cpp:168-173
    template <class T, class PT = double>
    __host__ __device__ static void systemFlowMap(const T *x, const T *u, const PT *par, T *f)
    {
        const PT alpha_m = par[0];
        const T m = x[0];
        const T qw = x[7], qx = x[8], qy = x[9], qz = x[10];
        const T wx = x[11], wy = x[12], wz = x[13];
        const T Tx = u[0], Ty = u[1], Tz = u[2];
        const T tx = 2. * qx, ty = 2. * qy, tz = 2. * qz;
        const T twx = tx * qw, twy = ty * qw, twz = tz * qw;
        const T txx = tx * qx, txy = ty * qx, txz = tz * qx;
        const T tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        const T R00 = 1. - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
        const T R10 = txy + twz, R11 = 1. - (txx + tzz), R12 = tyz - twx;
        const T R20 = txz - twy, R21 = tyz + twx, R22 = 1. - (txx + tyy);
        f[0] = -alpha_m * dsqrt(Tx * Tx + Ty * Ty + Tz * Tz);
        f[1] = x[4];
        f[2] = x[5];
        f[3] = x[6];
        const T im = 1. / m;
        f[4] = (im * R00) * Tx + (im * R01) * Ty + (im * R02) * Tz + par[1];
        f[5] = (im * R10) * Tx + (im * R11) * Ty + (im * R12) * Tz + par[2];
        f[6] = (im * R20) * Tx + (im * R21) * Ty + (im * R22) * Tz + par[3];
        f[7] = 0.5 * (-wx * qx - wy * qy - wz * qz);
        f[8] = 0.5 * (wx * qw + wz * qy - wy * qz);
        f[9] = 0.5 * (wy * qw - wz * qx + wx * qz);
        f[10] = 0.5 * (wz * qw + wy * qx - wx * qy);
        const PT rx = par[7], ry = par[8], rz = par[9];
        const T cxr = ry * Tz - rz * Ty, cyr = rz * Tx - rx * Tz, czr = rx * Ty - ry * Tx;
        f[11] = (1. / par[4]) * cxr - (wy * wz - wz * wy);
        f[12] = (1. / par[5]) * cyr - (wz * wx - wx * wz);
        f[13] = (1. / par[6]) * (czr + u[3]) - (wx * wy - wy * wx);
    }
};

// Model plugin: Rocket2d (planar rocket, 6/2/6), scpp_models/src/rocket2d.cpp:7-38
struct Rocket2dModel
{
    static constexpr int NX = 6, NU = 2, NP = 6;
    static constexpr int MODEL_ID = 1;
    using JacobianRows = Rocket2dJacobianRows;
    using JacobianTable = Rocket2dJacobianTable;
    template <class T, class PT = double>
    __host__ __device__ static void systemFlowMap(const T *x, const T *u, const PT *par, T *f)
    {
        const PT m = par[0], J_B = par[1];
        const T eta = x[4], w = x[5];
        const T angle = u[0], magnitude = u[1];
        const T TBx = dcos(angle) * 0. - dsin(angle) * magnitude;
        const T TBy = dsin(angle) * 0. + dcos(angle) * magnitude;
        const T ce = dcos(eta), se = dsin(eta);
        const T RTx = ce * TBx - se * TBy;
        const T RTy = se * TBx + ce * TBy;
        f[0] = x[2];
        f[1] = x[3];
        f[2] = (1. / m) * RTx + par[2];
        f[3] = (1. / m) * RTy + par[3];
        f[4] = w;
        f[5] = (1. / J_B) * (par[4] * TBy - par[5] * TBx);
    }
};

} // namespace scpp
