// Model plugin, first half: Lander3dof -- a point-mass powered-descent vehicle (7 states / 3 inputs / 4 flow-map parameters).
// NOT a model of the reference: it is the THIRD model of this repository, added in round 6 to show what a user of the reference's plugin interface
// (a class derived from SystemModel: systemFlowMap + addApplicationConstraints + the parameter hooks, scpp_core/include/systemModel.hpp:64-82) writes
// here -- this flow map, one constraint table (constraint_table.h: Lander3dofSC), one plugin struct (sc_kernels.h: Lander3dofPlugin, registered in
// `Plugins`), a parameter struct + three entry points in the C ABI.  Everything else (discretisation, simulation, the interior-point solver, the SC and
// SCvx loops, both streaming engines) is instantiated from those by the registry.
//
// State [m, r_I(3), v_I(3)], input T_I(3) (thrust in the inertial frame), par = [alpha_m, g_I(3)]:
//   m' = -alpha_m ||T||,  r' = v,  v' = T / m + g            (RocketQuat, rocketQuat.cpp:7-37, without the attitude states)
// The analytic Jacobian rows the discretisation uses are GENERATED from this function (tools/flowmap_symbolic.cpp + tools/gen_model_jacobian.py), as
// for the reference's two models.
#pragma once
#include "common.h"
#ifndef SCPP_FLOWMAP_ONLY
#include "model_jacobian_rows.h"
#else
namespace scpp
{
struct Lander3dofJacobianRows;
struct Lander3dofJacobianTable;
} // namespace scpp
#endif

namespace scpp
{

struct Lander3dofModel
{
    static constexpr int NX = 7, NU = 3, NP = 4;
    static constexpr int MODEL_ID = 2;
    using JacobianRows = Lander3dofJacobianRows;
    using JacobianTable = Lander3dofJacobianTable;
    template <class T, class PT = double>
    __host__ __device__ static void systemFlowMap(const T *x, const T *u, const PT *par, T *f)
    {
        const PT alpha_m = par[0];
        const T im = 1. / x[0];
        f[0] = -alpha_m * dsqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        f[1] = x[4];
        f[2] = x[5];
        f[3] = x[6];
        f[4] = im * u[0] + par[1];
        f[5] = im * u[1] + par[2];
        f[6] = im * u[2] + par[3];
    }
};

} // namespace scpp
