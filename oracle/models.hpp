// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the reference model plugins:
//   RocketQuat : scpp_models/src/rocketQuat.cpp:7-332, scpp_models/include/common.hpp:30-38,125-134
//   Rocket2d   : scpp_models/src/rocket2d.cpp:7-232
// and of the model boundary scpp_core/include/systemDynamics.hpp:181-235 (computef /
// computeJacobians; CppAD replaced by forward duals) and systemModel.hpp:64-158.
// Reference quirks are reproduced on purpose: w.cross(w)==0, k/K interpolation,
// (T_max-T_min)/2 initial thrust for RocketQuat, un-normalised quaternion rotation matrix,
// thrust_const refreshed once per solve().
// Parity status: PINNED for the flow maps and their Jacobians by tests/golden/*_jacobians.npz (sympy, independent of this
// code, 1e-13); configuration loading and the quirks are restated from the source only.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "dual.hpp"
#include "info_parser.hpp"
#include "socp.hpp"

namespace oracle
{

// trajectoryData.hpp:8-32
struct TrajectoryData
{
    int nx = 0, nu = 0;
    int K = 0, nU = 0;
    std::vector<double> X; // [K][nx]
    std::vector<double> U; // [nU][nu]
    double t = 0.;
    void initialize(int nx_, int nu_, int K_, bool interpolate_input)
    {
        nx = nx_;
        nu = nu_;
        K = K_;
        nU = interpolate_input ? K : K - 1;
        X.assign(size_t(K) * nx, 0.);
        U.assign(size_t(nU) * nu, 0.);
        t = 0.;
    }
    bool interpolatedInput() const { return nU == K; }
    double *x(int k) { return &X[size_t(k) * nx]; }
    double *u(int k) { return &U[size_t(k) * nu]; }
    const double *x(int k) const { return &X[size_t(k) * nx]; }
    const double *u(int k) const { return &U[size_t(k) * nu]; }
};

// discretizationData.hpp:8-53 ; matrices stored ROW-major here: A[k][i*nx+j]
struct DiscretizationData
{
    int nx = 0, nu = 0, K = 0;
    bool foh = false, vt = false;
    std::vector<double> A, B, C, s, z;
    void initialize(int nx_, int nu_, int K_, bool interpolate_input, bool free_final_time)
    {
        nx = nx_;
        nu = nu_;
        K = K_;
        foh = interpolate_input;
        vt = free_final_time;
        A.assign(size_t(K - 1) * nx * nx, 0.);
        B.assign(size_t(K - 1) * nx * nu, 0.);
        C.assign(foh ? size_t(K - 1) * nx * nu : 0, 0.);
        s.assign(vt ? size_t(K - 1) * nx : 0, 0.);
        z.assign(size_t(K - 1) * nx, 0.);
    }
    bool interpolatedInput() const { return foh; }
    bool variableTime() const { return vt; }
};

// counter-based RNG shared by oracle and product: SplitMix64 keyed by
// (seed, instance, draw) -> uniform double in [-1,1)
inline double counterUniform(uint64_t seed, uint64_t instance, uint64_t draw)
{
    uint64_t z = seed + (instance * 8ull + draw + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const double u01 = double(z >> 11) * (1.0 / 9007199254740992.0);
    return 2. * u01 - 1.;
}

// q = Rx(phi) Ry(theta) Rz(psi), (w,x,y,z)   common.hpp:30-38
inline void eulerToQuaternionXYZ(const double eta[3], double q[4])
{
    const double cx = std::cos(0.5 * eta[0]), sx = std::sin(0.5 * eta[0]);
    const double cy = std::cos(0.5 * eta[1]), sy = std::sin(0.5 * eta[1]);
    const double cz = std::cos(0.5 * eta[2]), sz = std::sin(0.5 * eta[2]);
    // qx*qy
    const double aw = cx * cy, ax = sx * cy, ay = cx * sy, az = sx * sy;
    // (a)*qz, qz = (cz,0,0,sz)
    q[0] = aw * cz - az * sz;
    q[1] = ax * cz + ay * sz;
    q[2] = ay * cz - ax * sz;
    q[3] = aw * sz + az * cz;
}

template <class M, int NX_, int NU_, int NP_>
struct ModelBase
{
    // current flow-map parameters (systemDynamics.hpp:170-179)
    double par[NP_];
    // systemDynamics.hpp:181-203
    void computef(const double *x, const double *u, double *f) const
    {
        M::template flow<double>(x, u, par, f);
    }
    // systemDynamics.hpp:206-235 : A = df/dx, B = df/du (row-major out)
    void computeJacobians(const double *x, const double *u, double *A, double *B) const
    {
        constexpr int NX = NX_, NU = NU_, NP = NP_, ND = NX + NU;
        using D = Dual<ND>;
        D xd[NX], ud[NU], pd[NP], fd[NX];
        for (int i = 0; i < NX; i++)
        {
            xd[i] = D(x[i]);
            xd[i].d[i] = 1.;
        }
        for (int i = 0; i < NU; i++)
        {
            ud[i] = D(u[i]);
            ud[i].d[NX + i] = 1.;
        }
        for (int i = 0; i < NP; i++)
            pd[i] = D(par[i]);
        M::template flow<D>(xd, ud, pd, fd);
        for (int i = 0; i < NX; i++)
        {
            for (int j = 0; j < NX; j++)
                A[i * NX + j] = fd[i].d[j];
            for (int j = 0; j < NU; j++)
                B[i * NU + j] = fd[i].d[NX + j];
        }
    }
};

// ---------------------------------------------------------------------------------------------
struct RocketQuat : ModelBase<RocketQuat, 14, 4, 10>
{
    static constexpr int NX = 14, NU = 4, NP = 10;
    static const char *modelName() { return "RocketQuat"; }

    // rocketQuat.cpp:7-37
    template <class T>
    static void flow(const T *x, const T *u, const T *par, T *f)
    {
        const T alpha_m = par[0];
        const T m = x[0];
        const T qw = x[7], qx = x[8], qy = x[9], qz = x[10];
        const T wx = x[11], wy = x[12], wz = x[13];
        const T Tx = u[0], Ty = u[1], Tz = u[2];
        // Eigen::Quaternion(w,x,y,z).toRotationMatrix(), no normalisation
        const T tx = 2. * qx, ty = 2. * qy, tz = 2. * qz;
        const T twx = tx * qw, twy = ty * qw, twz = tz * qw;
        const T txx = tx * qx, txy = ty * qx, txz = tz * qx;
        const T tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        const T R00 = 1. - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
        const T R10 = txy + twz, R11 = 1. - (txx + tzz), R12 = tyz - twx;
        const T R20 = txz - twy, R21 = tyz + twx, R22 = 1. - (txx + tyy);

        f[0] = -alpha_m * sqrt(Tx * Tx + Ty * Ty + Tz * Tz);
        f[1] = x[4];
        f[2] = x[5];
        f[3] = x[6];
        const T im = 1. / m;
        f[4] = (im * R00) * Tx + (im * R01) * Ty + (im * R02) * Tz + par[1];
        f[5] = (im * R10) * Tx + (im * R11) * Ty + (im * R12) * Tz + par[2];
        f[6] = (im * R20) * Tx + (im * R21) * Ty + (im * R22) * Tz + par[3];
        // 0.5 * Omega(w) * q   common.hpp:125-134
        f[7] = 0.5 * (-wx * qx - wy * qy - wz * qz);
        f[8] = 0.5 * (wx * qw + wz * qy - wy * qz);
        f[9] = 0.5 * (wy * qw - wz * qx + wx * qz);
        f[10] = 0.5 * (wz * qw + wy * qx - wx * qy);
        // J^-1 (r_T x T + torque) - w x w
        const T rx = par[7], ry = par[8], rz = par[9];
        const T cxr = ry * Tz - rz * Ty, cyr = rz * Tx - rx * Tz, czr = rx * Ty - ry * Tx;
        f[11] = (1. / par[4]) * cxr - (wy * wz - wz * wy);
        f[12] = (1. / par[5]) * cyr - (wz * wx - wx * wz);
        f[13] = (1. / par[6]) * (czr + u[3]) - (wx * wy - wy * wx);
    }

    struct Parameters
    {
        bool exact_minimum_thrust = true, enable_roll_control = false, random_initial_state = false;
        double g_I[3], J_B[3], r_T_B[3];
        double alpha_m, T_min, T_max, t_max;
        double gimbal_max, theta_max, gamma_gs, w_B_max;
        double x_init[14], x_final[14];
        double rpy_init[3];
        double final_time;
        double m_scale = 1., r_scale = 1.;

        // rocketQuat.cpp:234-289
        void loadFromFile(const std::string &path)
        {
            ParameterServer param(path);
            double I_sp, m_init, m_dry;
            double r_init[3], v_init[3], w_init[3], r_final[3], v_final[3], rpy_final[3], w_final[3];
            param.loadVector("g_I", g_I, 3);
            param.loadVector("J_B", J_B, 3);
            param.loadVector("r_T_B", r_T_B, 3);
            param.loadScalar("m_init", m_init);
            param.loadVector("r_init", r_init, 3);
            param.loadVector("v_init", v_init, 3);
            param.loadVector("rpy_init", rpy_init, 3);
            param.loadVector("w_init", w_init, 3);
            param.loadVector("w_final", w_final, 3);
            param.loadScalar("m_dry", m_dry);
            param.loadVector("r_final", r_final, 3);
            param.loadVector("v_final", v_final, 3);
            param.loadVector("rpy_final", rpy_final, 3);
            param.loadScalar("T_min", T_min);
            param.loadScalar("T_max", T_max);
            param.loadScalar("t_max", t_max);
            param.loadScalar("I_sp", I_sp);
            param.loadScalar("gimbal_max", gimbal_max);
            param.loadScalar("theta_max", theta_max);
            param.loadScalar("gamma_gs", gamma_gs);
            param.loadScalar("w_B_max", w_B_max);
            param.loadScalar("random_initial_state", random_initial_state);
            param.loadScalar("final_time", final_time);
            param.loadScalar("exact_minimum_thrust", exact_minimum_thrust);
            param.loadScalar("enable_roll_control", enable_roll_control);
            const double d2r = M_PI / 180.;
            gimbal_max *= d2r;
            theta_max *= d2r;
            gamma_gs *= d2r;
            w_B_max *= d2r;
            for (int i = 0; i < 3; i++)
            {
                rpy_init[i] *= d2r;
                rpy_final[i] *= d2r;
                w_init[i] *= d2r;
                w_final[i] *= d2r;
            }
            alpha_m = 1. / (I_sp * std::fabs(g_I[2]));
            double q_init[4], q_final[4];
            eulerToQuaternionXYZ(rpy_init, q_init);
            eulerToQuaternionXYZ(rpy_final, q_final);
            x_init[0] = m_init;
            x_final[0] = m_dry;
            for (int i = 0; i < 3; i++)
            {
                x_init[1 + i] = r_init[i];
                x_init[4 + i] = v_init[i];
                x_init[11 + i] = w_init[i];
                x_final[1 + i] = r_final[i];
                x_final[4 + i] = v_final[i];
                x_final[11 + i] = w_final[i];
            }
            for (int i = 0; i < 4; i++)
            {
                x_init[7 + i] = q_init[i];
                x_final[7 + i] = q_final[i];
            }
        }

        // The reference's (commented-out) recipe rocketQuat.cpp:203-227, made deterministic
        // per SURVEY.md §8(d): 7 counter-based uniforms per instance; all 4 quaternion
        // components are written.
        void randomizeInitialState(uint64_t seed, uint64_t instance)
        {
            x_init[1] *= counterUniform(seed, instance, 0);
            x_init[2] *= counterUniform(seed, instance, 1);
            x_init[4] *= counterUniform(seed, instance, 2);
            x_init[5] *= counterUniform(seed, instance, 3);
            x_init[6] *= 1. + 0.2 * counterUniform(seed, instance, 4);
            double euler[3];
            euler[0] = counterUniform(seed, instance, 5) * rpy_init[0];
            euler[1] = counterUniform(seed, instance, 6) * rpy_init[1];
            euler[2] = rpy_init[2];
            eulerToQuaternionXYZ(euler, &x_init[7]);
        }

        // rocketQuat.cpp:291-312
        void nondimensionalize()
        {
            m_scale = x_init[0];
            r_scale = std::sqrt(x_init[1] * x_init[1] + x_init[2] * x_init[2] + x_init[3] * x_init[3]);
            alpha_m *= r_scale;
            for (int i = 0; i < 3; i++)
            {
                r_T_B[i] /= r_scale;
                g_I[i] /= r_scale;
                J_B[i] /= m_scale * r_scale * r_scale;
            }
            x_init[0] /= m_scale;
            x_final[0] /= m_scale;
            for (int i = 1; i < 7; i++)
            {
                x_init[i] /= r_scale;
                x_final[i] /= r_scale;
            }
            T_min /= m_scale * r_scale;
            T_max /= m_scale * r_scale;
            t_max /= m_scale * r_scale * r_scale;
        }
        // rocketQuat.cpp:314-332
        void redimensionalize()
        {
            alpha_m /= r_scale;
            for (int i = 0; i < 3; i++)
            {
                r_T_B[i] *= r_scale;
                g_I[i] *= r_scale;
                J_B[i] *= m_scale * r_scale * r_scale;
            }
            x_init[0] *= m_scale;
            x_final[0] *= m_scale;
            for (int i = 1; i < 7; i++)
            {
                x_init[i] *= r_scale;
                x_final[i] *= r_scale;
            }
            T_min *= m_scale * r_scale;
            T_max *= m_scale * r_scale;
            t_max *= m_scale * r_scale * r_scale;
        }
    } p;

    struct DynamicParameters
    {
        double tilt_const = 0., gs_const = 0., gimbal_const = 0.;
        std::vector<double> thrust_const; // [K][3]
    } p_dyn;

    void loadParameters(const std::string &folder) { p.loadFromFile(folder + "/model.info"); }
    void nondimensionalize() { p.nondimensionalize(); }
    void redimensionalize() { p.redimensionalize(); }

    // rocketQuat.cpp:39-68
    void getInitializedTrajectory(TrajectoryData &td) const
    {
        const int K = td.K;
        for (int k = 0; k < K; k++)
        {
            const double alpha1 = double(K - k) / K;
            const double alpha2 = double(k) / K;
            double *x = td.x(k);
            x[0] = alpha1 * p.x_init[0] + alpha2 * p.x_final[0];
            for (int i = 1; i < 7; i++)
                x[i] = alpha1 * p.x_init[i] + alpha2 * p.x_final[i];
            // Eigen slerp(alpha2)
            const double *q0 = &p.x_init[7], *q1 = &p.x_final[7];
            const double one = 1. - 2.220446049250313e-16;
            const double d = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
            const double absD = std::fabs(d);
            double scale0, scale1;
            if (absD >= one)
            {
                scale0 = 1. - alpha2;
                scale1 = alpha2;
            }
            else
            {
                const double theta = std::acos(absD);
                const double sinTheta = std::sin(theta);
                scale0 = std::sin((1. - alpha2) * theta) / sinTheta;
                scale1 = std::sin(alpha2 * theta) / sinTheta;
            }
            if (d < 0.)
                scale1 = -scale1;
            for (int i = 0; i < 4; i++)
                x[7 + i] = scale0 * q0[i] + scale1 * q1[i];
            for (int i = 11; i < 14; i++)
                x[i] = alpha1 * p.x_init[i] + alpha2 * p.x_final[i];
        }
        for (int k = 0; k < td.nU; k++)
        {
            double *u = td.u(k);
            u[0] = 0.;
            u[1] = 0.;
            u[2] = (p.T_max - p.T_min) / 2.;
            u[3] = 0.;
        }
        td.t = p.final_time;
    }

    // rocketQuat.cpp:156-173 ; U0 is the trajectory the SOCP was bound to (td.U)
    void getNewModelParameters(const TrajectoryData &td)
    {
        par[0] = p.alpha_m;
        for (int i = 0; i < 3; i++)
        {
            par[1 + i] = p.g_I[i];
            par[4 + i] = p.J_B[i];
            par[7 + i] = p.r_T_B[i];
        }
        p_dyn.gimbal_const = std::tan(p.gimbal_max);
        p_dyn.gs_const = std::tan(p.gamma_gs);
        p_dyn.tilt_const = std::sqrt((1. - std::cos(p.theta_max)) / 2.);
        if (p.exact_minimum_thrust)
        {
            p_dyn.thrust_const.assign(size_t(td.nU) * 3, 0.);
            for (int k = 0; k < td.nU; k++)
            {
                const double *u = td.u(k);
                const double z = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
                const double s = z > 0. ? 1. / std::sqrt(z) : 1.; // Eigen normalized()
                for (int i = 0; i < 3; i++)
                    p_dyn.thrust_const[size_t(k) * 3 + i] = u[i] * s;
            }
        }
    }

    // rocketQuat.cpp:175-201
    void nondimensionalizeTrajectory(TrajectoryData &td) const
    {
        for (int k = 0; k < td.K; k++)
        {
            double *x = td.x(k);
            x[0] /= p.m_scale;
            for (int i = 1; i < 7; i++)
                x[i] /= p.r_scale;
        }
        for (int k = 0; k < td.nU; k++)
        {
            double *u = td.u(k);
            for (int i = 0; i < 3; i++)
                u[i] /= p.m_scale * p.r_scale;
            u[3] /= p.m_scale * p.r_scale * p.r_scale;
        }
    }
    void redimensionalizeTrajectory(TrajectoryData &td) const
    {
        for (int k = 0; k < td.K; k++)
        {
            double *x = td.x(k);
            x[0] *= p.m_scale;
            for (int i = 1; i < 7; i++)
                x[i] *= p.r_scale;
        }
        for (int k = 0; k < td.nU; k++)
        {
            double *u = td.u(k);
            for (int i = 0; i < 3; i++)
                u[i] *= p.m_scale * p.r_scale;
            u[3] *= p.m_scale * p.r_scale * p.r_scale;
        }
    }

    // rocketQuat.cpp:70-144.  vX(i,k), vU(i,k) give variable indices; keys: see sc.hpp
    template <class FX, class FU, class KeyFn>
    void addApplicationConstraints(Socp &socp, int K, int nU, FX vX, FU vU, KeyFn key) const
    {
        // Initial state
        for (int i = 0; i < NX; i++)
            socp.addEq(Aff(-p.x_init[i]).add(vX(i, 0), 1.), key.stageEq(0));
        // Final state: mass and roll free
        for (int i : {1, 2, 3, 4, 5, 6, 8, 9, 11, 12, 13})
            socp.addEq(Aff(-p.x_final[i]).add(vX(i, K - 1), 1.), key.stageEq(K - 1));
        // Mass
        for (int k = 0; k < K; k++)
            socp.addGe0(Aff(-p.x_final[0]).add(vX(0, k), 1.), key.stageCone(k));
        // Glide slope  || X[1:3,k] || <= gs_const * X[3,k]
        for (int k = 0; k < K; k++)
            socp.addSoc({Aff().add(vX(3, k), p_dyn.gs_const), Aff().add(vX(1, k), 1.), Aff().add(vX(2, k), 1.)},
                        key.stageCone(k));
        // Max tilt
        for (int k = 0; k < K; k++)
            socp.addSoc({Aff(p_dyn.tilt_const), Aff().add(vX(8, k), 1.), Aff().add(vX(9, k), 1.)}, key.stageCone(k));
        // Max rotation velocity
        for (int k = 0; k < K; k++)
            socp.addSoc({Aff(p.w_B_max), Aff().add(vX(11, k), 1.), Aff().add(vX(12, k), 1.), Aff().add(vX(13, k), 1.)},
                        key.stageCone(k));
        // Final input
        for (int i : {0, 1, 3})
            socp.addEq(Aff().add(vU(i, nU - 1), 1.), key.stageEq(nU - 1));
        if (p.exact_minimum_thrust)
        {
            for (int k = 0; k < nU; k++)
            {
                Aff e(-p.T_min);
                for (int i = 0; i < 3; i++)
                    e.add(vU(i, k), p_dyn.thrust_const[size_t(k) * 3 + i]);
                socp.addGe0(e, key.stageCone(k));
            }
        }
        else
        {
            for (int k = 0; k < nU; k++)
                socp.addGe0(Aff(-p.T_min).add(vU(2, k), 1.), key.stageCone(k));
        }
        // Max thrust
        for (int k = 0; k < nU; k++)
            socp.addSoc({Aff(p.T_max), Aff().add(vU(0, k), 1.), Aff().add(vU(1, k), 1.), Aff().add(vU(2, k), 1.)},
                        key.stageCone(k));
        // Max gimbal
        for (int k = 0; k < nU; k++)
            socp.addSoc({Aff().add(vU(2, k), p_dyn.gimbal_const), Aff().add(vU(0, k), 1.), Aff().add(vU(1, k), 1.)},
                        key.stageCone(k));
        if (p.enable_roll_control)
        {
            for (int k = 0; k < nU; k++)
            {
                socp.addGe0(Aff(p.t_max).add(vU(3, k), 1.), key.stageCone(k));
                socp.addGe0(Aff(p.t_max).add(vU(3, k), -1.), key.stageCone(k));
            }
        }
        else
        {
            for (int k = 0; k < K; k++)
                socp.addEq(Aff().add(vX(13, k), 1.), key.stageEq(k));
            for (int k = 0; k < nU; k++)
                socp.addEq(Aff().add(vU(3, k), 1.), key.stageEq(k));
        }
    }
};

// ---------------------------------------------------------------------------------------------
struct Rocket2d : ModelBase<Rocket2d, 6, 2, 6>
{
    static constexpr int NX = 6, NU = 2, NP = 6;
    static const char *modelName() { return "Rocket2D"; }

    // rocket2d.cpp:7-38
    template <class T>
    static void flow(const T *x, const T *u, const T *par, T *f)
    {
        const T m = par[0], J_B = par[1];
        const T eta = x[4], w = x[5];
        const T angle = u[0], magnitude = u[1];
        // Rotation2D(angle) * (0, magnitude)
        const T TBx = cos(angle) * 0. - sin(angle) * magnitude;
        const T TBy = sin(angle) * 0. + cos(angle) * magnitude;
        const T ce = cos(eta), se = sin(eta);
        const T RTx = ce * TBx - se * TBy;
        const T RTy = se * TBx + ce * TBy;
        f[0] = x[2];
        f[1] = x[3];
        f[2] = (1. / m) * RTx + par[2];
        f[3] = (1. / m) * RTy + par[3];
        f[4] = w;
        f[5] = (1. / J_B) * (par[4] * TBy - par[5] * TBx);
    }

    struct Parameters
    {
        double m, J_B, g_I[2], r_T_B[2];
        double x_init[6], x_final[6];
        double eta_init, eta_final;
        double final_time;
        double T_min, T_max, gamma_gs, gimbal_max, theta_max, w_B_max;
        double tan_gamma_gs = 0.;
        bool constrain_initial_final = true, add_slack_variables = false;
        double m_scale = 1., r_scale = 1.;

        // rocket2d.cpp:150-198
        void loadFromFile(const std::string &path)
        {
            ParameterServer param(path);
            double r_init[2], v_init[2], r_final[2], v_final[2], w_init, w_final;
            param.loadVector("g_I", g_I, 2);
            param.loadScalar("J_B", J_B);
            param.loadVector("r_T_B", r_T_B, 2);
            param.loadVector("r_init", r_init, 2);
            param.loadVector("v_init", v_init, 2);
            param.loadScalar("eta_init", eta_init);
            param.loadScalar("w_init", w_init);
            param.loadVector("r_final", r_final, 2);
            param.loadVector("v_final", v_final, 2);
            param.loadScalar("eta_final", eta_final);
            param.loadScalar("w_final", w_final);
            param.loadScalar("final_time", final_time);
            param.loadScalar("m", m);
            param.loadScalar("T_min", T_min);
            param.loadScalar("T_max", T_max);
            param.loadScalar("gamma_gs", gamma_gs);
            param.loadScalar("gimbal_max", gimbal_max);
            param.loadScalar("theta_max", theta_max);
            param.loadScalar("w_B_max", w_B_max);
            param.loadScalar("constrain_initial_final", constrain_initial_final);
            param.loadScalar("add_slack_variables", add_slack_variables);
            const double d2r = M_PI / 180.;
            gimbal_max *= d2r;
            theta_max *= d2r;
            gamma_gs *= d2r;
            w_B_max *= d2r;
            w_init *= d2r;
            w_final *= d2r;
            eta_init *= d2r;
            eta_final *= d2r;
            x_init[0] = r_init[0];
            x_init[1] = r_init[1];
            x_init[2] = v_init[0];
            x_init[3] = v_init[1];
            x_init[4] = eta_init;
            x_init[5] = w_init;
            x_final[0] = r_final[0];
            x_final[1] = r_final[1];
            x_final[2] = v_final[0];
            x_final[3] = v_final[1];
            x_final[4] = eta_final;
            x_final[5] = w_final;
        }
        // rocket2d.cpp:200-214
        void nondimensionalize()
        {
            r_scale = std::sqrt(x_init[0] * x_init[0] + x_init[1] * x_init[1]);
            m_scale = m;
            m /= m_scale;
            for (int i = 0; i < 2; i++)
            {
                r_T_B[i] /= r_scale;
                g_I[i] /= r_scale;
            }
            J_B /= m_scale * r_scale * r_scale;
            for (int i = 0; i < 4; i++)
            {
                x_init[i] /= r_scale;
                x_final[i] /= r_scale;
            }
            T_min /= m_scale * r_scale;
            T_max /= m_scale * r_scale;
        }
        // rocket2d.cpp:216-231
        void redimensionalize()
        {
            m *= m_scale;
            for (int i = 0; i < 2; i++)
            {
                r_T_B[i] *= r_scale;
                g_I[i] *= r_scale;
            }
            J_B *= m_scale * r_scale * r_scale;
            for (int i = 0; i < 4; i++)
            {
                x_init[i] *= r_scale;
                x_final[i] *= r_scale;
            }
            T_min *= m_scale * r_scale;
            T_max *= m_scale * r_scale;
        }
    } p;

    void loadParameters(const std::string &folder) { p.loadFromFile(folder + "/model.info"); }
    void nondimensionalize() { p.nondimensionalize(); }
    void redimensionalize() { p.redimensionalize(); }

    // rocket2d.cpp:120-136
    void getInitializedTrajectory(TrajectoryData &td) const
    {
        const int K = td.K;
        for (int k = 0; k < K; k++)
        {
            const double alpha1 = double(K - k) / K;
            const double alpha2 = double(k) / K;
            for (int i = 0; i < NX; i++)
                td.x(k)[i] = alpha1 * p.x_init[i] + alpha2 * p.x_final[i];
        }
        for (int k = 0; k < td.nU; k++)
        {
            td.u(k)[0] = 0.;
            td.u(k)[1] = (p.T_max + p.T_min) / 2;
        }
        td.t = p.final_time;
    }

    // rocket2d.cpp:143-148
    void getNewModelParameters(const TrajectoryData &)
    {
        par[0] = p.m;
        par[1] = p.J_B;
        par[2] = p.g_I[0];
        par[3] = p.g_I[1];
        par[4] = p.r_T_B[0];
        par[5] = p.r_T_B[1];
        p.tan_gamma_gs = std::tan(p.gamma_gs);
    }

    // rocket2d.cpp:92-118
    void nondimensionalizeTrajectory(TrajectoryData &td) const
    {
        for (int k = 0; k < td.K; k++)
            for (int i = 0; i < 4; i++)
                td.x(k)[i] /= p.r_scale;
        for (int k = 0; k < td.nU; k++)
            td.u(k)[1] /= p.m_scale * p.r_scale;
    }
    void redimensionalizeTrajectory(TrajectoryData &td) const
    {
        for (int k = 0; k < td.K; k++)
            for (int i = 0; i < 4; i++)
                td.x(k)[i] *= p.r_scale;
        for (int k = 0; k < td.nU; k++)
            td.u(k)[1] *= p.m_scale * p.r_scale;
    }

    // rocket2d.cpp:46-84
    template <class FX, class FU, class KeyFn>
    void addApplicationConstraints(Socp &socp, int K, int nU, FX vX, FU vU, KeyFn key) const
    {
        if (p.constrain_initial_final)
        {
            for (int i = 0; i < NX; i++)
                socp.addEq(Aff(-p.x_init[i]).add(vX(i, 0), 1.), key.stageEq(0));
            for (int i = 0; i < NX; i++)
                socp.addEq(Aff(-p.x_final[i]).add(vX(i, K - 1), 1.), key.stageEq(K - 1));
            socp.addEq(Aff().add(vU(0, nU - 1), 1.), key.stageEq(nU - 1));
        }
        // Glideslope: || X[0,k] || <= tan_gamma_gs * X[1,k]
        for (int k = 0; k < K; k++)
            socp.addSoc({Aff().add(vX(1, k), p.tan_gamma_gs), Aff().add(vX(0, k), 1.)}, key.stageCone(k));
        // boxes
        for (int k = 0; k < K; k++)
        {
            socp.addGe0(Aff(p.theta_max).add(vX(4, k), 1.), key.stageCone(k));
            socp.addGe0(Aff(p.theta_max).add(vX(4, k), -1.), key.stageCone(k));
        }
        for (int k = 0; k < K; k++)
        {
            socp.addGe0(Aff(p.w_B_max).add(vX(5, k), 1.), key.stageCone(k));
            socp.addGe0(Aff(p.w_B_max).add(vX(5, k), -1.), key.stageCone(k));
        }
        for (int k = 0; k < nU; k++)
        {
            socp.addGe0(Aff(p.gimbal_max).add(vU(0, k), 1.), key.stageCone(k));
            socp.addGe0(Aff(p.gimbal_max).add(vU(0, k), -1.), key.stageCone(k));
        }
        for (int k = 0; k < nU; k++)
        {
            socp.addGe0(Aff(-p.T_min).add(vU(1, k), 1.), key.stageCone(k));
            socp.addGe0(Aff(p.T_max).add(vU(1, k), -1.), key.stageCone(k));
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Lander3dof: NOT a model of the reference -- the third model of this repository (csrc/model_lander3dof.h: a point-mass powered-descent
// vehicle, RocketQuat without the attitude states), added in round 6 to exercise the model-plugin path end to end.  This is the checker's
// restatement of it in the reference's own plugin shape (systemModel.hpp:64-82): flow map, parameters with nondimensionalize /
// redimensionalize, getInitializedTrajectory, getNewModelParameters, addApplicationConstraints on the literal problem.
struct Lander3dof : ModelBase<Lander3dof, 7, 3, 4>
{
    static constexpr int NX = 7, NU = 3, NP = 4;
    static const char *modelName() { return "Lander3dof"; }

    template <class T>
    static void flow(const T *x, const T *u, const T *par, T *f)
    {
        const T m = x[0];
        f[0] = -par[0] * sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        f[1] = x[4];
        f[2] = x[5];
        f[3] = x[6];
        f[4] = u[0] / m + par[1];
        f[5] = u[1] / m + par[2];
        f[6] = u[2] / m + par[3];
    }

    struct Parameters
    {
        bool exact_minimum_thrust = true;
        double g_I[3];
        double alpha_m, T_min, T_max, pointing_max, gamma_gs;
        double x_init[7], x_final[7];
        double final_time;
        double m_scale = 1., r_scale = 1.;

        void loadFromFile(const std::string &path)
        {
            ParameterServer param(path);
            double I_sp, m_init, m_dry, r_init[3], v_init[3], r_final[3], v_final[3];
            param.loadVector("g_I", g_I, 3);
            param.loadScalar("m_init", m_init);
            param.loadVector("r_init", r_init, 3);
            param.loadVector("v_init", v_init, 3);
            param.loadScalar("final_time", final_time);
            param.loadScalar("m_dry", m_dry);
            param.loadVector("r_final", r_final, 3);
            param.loadVector("v_final", v_final, 3);
            param.loadScalar("exact_minimum_thrust", exact_minimum_thrust);
            param.loadScalar("I_sp", I_sp);
            param.loadScalar("T_min", T_min);
            param.loadScalar("T_max", T_max);
            param.loadScalar("pointing_max", pointing_max);
            param.loadScalar("gamma_gs", gamma_gs);
            const double d2r = M_PI / 180.;
            pointing_max *= d2r;
            gamma_gs *= d2r;
            alpha_m = 1. / (I_sp * std::fabs(g_I[2]));
            x_init[0] = m_init;
            x_final[0] = m_dry;
            for (int i = 0; i < 3; i++)
            {
                x_init[1 + i] = r_init[i];
                x_init[4 + i] = v_init[i];
                x_final[1 + i] = r_final[i];
                x_final[4 + i] = v_final[i];
            }
        }
        void nondimensionalize()
        {
            m_scale = x_init[0];
            r_scale = std::sqrt(x_init[1] * x_init[1] + x_init[2] * x_init[2] + x_init[3] * x_init[3]);
            alpha_m *= r_scale;
            for (int i = 0; i < 3; i++)
                g_I[i] /= r_scale;
            x_init[0] /= m_scale;
            x_final[0] /= m_scale;
            for (int i = 1; i < 7; i++)
            {
                x_init[i] /= r_scale;
                x_final[i] /= r_scale;
            }
            T_min /= m_scale * r_scale;
            T_max /= m_scale * r_scale;
        }
        void redimensionalize()
        {
            alpha_m /= r_scale;
            for (int i = 0; i < 3; i++)
                g_I[i] *= r_scale;
            x_init[0] *= m_scale;
            x_final[0] *= m_scale;
            for (int i = 1; i < 7; i++)
            {
                x_init[i] *= r_scale;
                x_final[i] *= r_scale;
            }
            T_min *= m_scale * r_scale;
            T_max *= m_scale * r_scale;
        }
    } p;

    struct DynamicParameters
    {
        double gs_const = 0., pointing_const = 0.;
        std::vector<double> thrust_const; // [K][3]
    } p_dyn;

    void loadParameters(const std::string &folder) { p.loadFromFile(folder + "/model.info"); }
    void nondimensionalize() { p.nondimensionalize(); }
    void redimensionalize() { p.redimensionalize(); }

    void getInitializedTrajectory(TrajectoryData &td) const
    {
        const int K = td.K;
        for (int k = 0; k < K; k++)
        {
            const double alpha1 = double(K - k) / K, alpha2 = double(k) / K; // (the reference models' k / K interpolation)
            for (int i = 0; i < NX; i++)
                td.x(k)[i] = alpha1 * p.x_init[i] + alpha2 * p.x_final[i];
        }
        for (int k = 0; k < td.nU; k++)
        {
            td.u(k)[0] = 0.;
            td.u(k)[1] = 0.;
            td.u(k)[2] = (p.T_max + p.T_min) / 2.;
        }
        td.t = p.final_time;
    }

    void getNewModelParameters(const TrajectoryData &td)
    {
        par[0] = p.alpha_m;
        for (int i = 0; i < 3; i++)
            par[1 + i] = p.g_I[i];
        p_dyn.gs_const = std::tan(p.gamma_gs);
        p_dyn.pointing_const = std::tan(p.pointing_max);
        if (p.exact_minimum_thrust)
        {
            p_dyn.thrust_const.assign(size_t(td.nU) * 3, 0.);
            for (int k = 0; k < td.nU; k++)
            {
                const double *u = td.u(k);
                const double z = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
                const double s = z > 0. ? 1. / std::sqrt(z) : 1.;
                for (int i = 0; i < 3; i++)
                    p_dyn.thrust_const[size_t(k) * 3 + i] = u[i] * s;
            }
        }
    }

    void nondimensionalizeTrajectory(TrajectoryData &td) const
    {
        for (int k = 0; k < td.K; k++)
        {
            td.x(k)[0] /= p.m_scale;
            for (int i = 1; i < 7; i++)
                td.x(k)[i] /= p.r_scale;
        }
        for (int k = 0; k < td.nU; k++)
            for (int i = 0; i < 3; i++)
                td.u(k)[i] /= p.m_scale * p.r_scale;
    }
    void redimensionalizeTrajectory(TrajectoryData &td) const
    {
        for (int k = 0; k < td.K; k++)
        {
            td.x(k)[0] *= p.m_scale;
            for (int i = 1; i < 7; i++)
                td.x(k)[i] *= p.r_scale;
        }
        for (int k = 0; k < td.nU; k++)
            for (int i = 0; i < 3; i++)
                td.u(k)[i] *= p.m_scale * p.r_scale;
    }

    template <class FX, class FU, class KeyFn>
    void addApplicationConstraints(Socp &socp, int K, int nU, FX vX, FU vU, KeyFn key) const
    {
        // initial state; final position and velocity (the final mass is free)
        for (int i = 0; i < NX; i++)
            socp.addEq(Aff(-p.x_init[i]).add(vX(i, 0), 1.), key.stageEq(0));
        for (int i = 1; i < 7; i++)
            socp.addEq(Aff(-p.x_final[i]).add(vX(i, K - 1), 1.), key.stageEq(K - 1));
        // the last input points straight up
        for (int i : {0, 1})
            socp.addEq(Aff().add(vU(i, nU - 1), 1.), key.stageEq(nU - 1));
        // mass >= m_dry
        for (int k = 0; k < K; k++)
            socp.addGe0(Aff(-p.x_final[0]).add(vX(0, k), 1.), key.stageCone(k));
        // glide slope || r_xy || <= tan(gamma_gs) r_z
        for (int k = 0; k < K; k++)
            socp.addSoc({Aff().add(vX(3, k), p_dyn.gs_const), Aff().add(vX(1, k), 1.), Aff().add(vX(2, k), 1.)}, key.stageCone(k));
        // minimum thrust: linearised at the trajectory the solve started from, or T_z >= T_min
        for (int k = 0; k < nU; k++)
        {
            if (p.exact_minimum_thrust)
            {
                Aff e(-p.T_min);
                for (int i = 0; i < 3; i++)
                    e.add(vU(i, k), p_dyn.thrust_const[size_t(k) * 3 + i]);
                socp.addGe0(e, key.stageCone(k));
            }
            else
                socp.addGe0(Aff(-p.T_min).add(vU(2, k), 1.), key.stageCone(k));
        }
        // maximum thrust || T || <= T_max
        for (int k = 0; k < nU; k++)
            socp.addSoc({Aff(p.T_max), Aff().add(vU(0, k), 1.), Aff().add(vU(1, k), 1.), Aff().add(vU(2, k), 1.)}, key.stageCone(k));
        // thrust pointing || T_xy || <= tan(pointing_max) T_z
        for (int k = 0; k < nU; k++)
            socp.addSoc({Aff().add(vU(2, k), p_dyn.pointing_const), Aff().add(vU(0, k), 1.), Aff().add(vU(1, k), 1.)}, key.stageCone(k));
    }
};

} // namespace oracle
