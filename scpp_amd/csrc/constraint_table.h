// Model plugin, second half: the APPLICATION CONSTRAINTS of a model as a compile-time table.
//
// The reference keeps the solver model-agnostic through SystemModel::addApplicationConstraints
// (scpp_core/include/systemModel.hpp:76-82), which records constraints on the Epigraph variables X and U:
//   RocketQuat   scpp_models/src/rocketQuat.cpp:70-144      Rocket2d   scpp_models/src/rocket2d.cpp:46-84
// Here the same information is a constexpr table per model, written in the reference's own row kinds
//   equalTo(X(:,0), x_init) / equalTo(X(i,K-1), x_final(i)) / equalTo(U(j,K-1), 0) / equalTo(X.row(i), 0)   -> fixed sets
//   lessThan(colwise norm of rows, par * row | par)                                                         -> SOC rows
//   greaterThan(row, par) / lessThan(row, par) / box(-par, row, par) / linearised minimum thrust           -> LP rows
// and everything the batched interior-point kernel needs -- which variables are presolved away at which node, which cones
// are active where, the affine slack map s = h - G w and its transpose, the sparsity pattern of the cones' Hessian
// blocks -- is DERIVED from the table at compile time (ipm_kernel.h).  A new model = its systemFlowMap<T> (model_*.h) +
// one table here; nothing in the solver changes.
//
// Stage variables.  Node k carries w_k = (x_k[XMAP...], u_k[UMAP...]) in one 16-wide tile (NXV + NUV <= 16, the operand
// size of v_mfma_f64_16x16x4_f64); states / inputs a model pins for the whole horizon (RocketQuat with roll control off:
// X(13,:) = 0, U(3,:) = 0) are not variables at all.
#pragma once
#include "common.h"

namespace scpp
{
namespace ipm
{

constexpr int NV = 16; // stage-variable tile

// ---- per-instance parameter block (doubles): the dynpar values the table refers to ----
enum InstPar
{
    IP_XINIT = 0,   // [14] nondimensional x_init
    IP_XFINAL = 14, // [14]
    IP_C0 = 28,     // model constants c0..c6 (RocketQuat: tan gamma_gs, tilt_const, w_B_max, T_min, T_max, tan gimbal_max, m_dry;
    IP_GS = 28,     //                         Rocket2d:   tan gamma_gs, theta_max,  w_B_max, T_min, T_max, gimbal_max)
    IP_TILT,
    IP_WMAX,
    IP_TMIN,
    IP_TMAX,
    IP_GIM,
    IP_MDRY,
    IP_WT = 35,
    IP_WTRT,
    IP_WTRX,
    IP_WVC,
    IP_PAR = 39, // [10] flow-map parameters
    IP_MSCALE = 49,
    IP_RSCALE,
    IP_FINALTIME,
    // SCvx mode (SCvxProblem.cpp:6-71) inside the same structure -- see oracle/structured_ipm.hpp (RQSocpInput::scvx):
    // delta_k is the constant trust_region, the state rows of the trust cone are zero padding, S = 0 decouples sigma
    IP_SCVX = 52,
    IP_TR = 53,
    IP_FIXEDT = 54, // != 0: SCAlgorithm with free_final_time false (SCProblem.cpp:33-35,78-100): sigma is not a variable
    IP_N = 56
};

// coefficient of a term: 1, a parameter of the instance block, or a component of the per-node thrust_const
// (rocketQuat.cpp:113-121: thrust_const(:,k) = normalised previous input, refreshed at solve() start)
constexpr int CF_ONE = -1, CF_UHAT0 = -10, CF_UHAT1 = -11, CF_UHAT2 = -12;
struct Term
{
    int var;    // stage variable index, -1: unused
    int coef;   // CF_ONE, CF_UHATj or an InstPar index
    double mul; // constant factor (sign)
};
// one row of the slack vector:  s_row = hmul * ip[hpar] (hpar < 0: no constant)  +  sum_t  mul_t * coef_t * w[var_t]
struct Row
{
    int hpar;
    double hmul;
    Term t[3];
};
constexpr Term NOTERM{-1, CF_ONE, 0.};
constexpr Row rowConst(int hpar) { return Row{hpar, 1., {NOTERM, NOTERM, NOTERM}}; }
constexpr Row rowVar(int var) { return Row{-1, 0., {Term{var, CF_ONE, 1.}, NOTERM, NOTERM}}; }
constexpr Row rowParVar(int par, int var) { return Row{-1, 0., {Term{var, par, 1.}, NOTERM, NOTERM}}; }
constexpr Row rowLower(int var, int par) { return Row{par, -1., {Term{var, CF_ONE, 1.}, NOTERM, NOTERM}}; }  // var >= par
constexpr Row rowUpper(int var, int par) { return Row{par, 1., {Term{var, CF_ONE, -1.}, NOTERM, NOTERM}}; }  // var <= par
constexpr Row rowBoxLo(int var, int par) { return Row{par, 1., {Term{var, CF_ONE, 1.}, NOTERM, NOTERM}}; }   // var >= -par

constexpr int MAXCONES = 6, MAXDIM = 4, MAXLP = 8;
struct Cone
{
    int dim;
    Row r[MAXDIM]; // r[0] >= || r[1..dim-1] ||
};

// ------------------------------------------------------------------------------------------------------------
// RocketQuat (rocketQuat.cpp:70-144), enable_roll_control = false.  w = (x0..x12, u0..u2).
// ------------------------------------------------------------------------------------------------------------
struct RocketQuatSC
{
    static constexpr int MODEL_ID = 0;
    static constexpr int NX = 14, NU = 4, NXV = 13, NUV = 3;
    // state / input index of every stage variable; XINV / UINV: stage variable of a state / input, -1 = pinned to 0
    static constexpr int XMAP[NXV] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
    static constexpr int UMAP[NUV] = {0, 1, 2};
    // equalTo(X.col(0), x_init) :79 ; equalTo(X(i, K-1), x_final(i)) for i in {1..6, 8, 9, 11, 12, 13} :83-89 ;
    // equalTo(U({0,1,3}, K-1), 0) :109-111 ; equalTo(X.row(13), 0), equalTo(U.row(3), 0) :141-142 (pinned, not variables)
    static constexpr unsigned FIXED_FIRST = 0x1FFFu;
    static constexpr unsigned FIXED_LAST = (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 8) | (1u << 9) |
                                           (1u << 11) | (1u << 12) | (1u << 13) | (1u << 14);
    static constexpr int NCONE = 5; // application cones (the trust-region cone of SCProblem.cpp:103-126 is the solver's own)
    static constexpr Cone CONES[NCONE] = {
        {3, {rowParVar(IP_GS, 3), rowVar(1), rowVar(2), {}}},          // glide slope  ||r_xy|| <= tan(gamma) r_z          :96-97
        {3, {rowConst(IP_TILT), rowVar(8), rowVar(9), {}}},            // tilt         ||q_xy|| <= sqrt((1-cos theta)/2)    :100-101
        {3, {rowConst(IP_WMAX), rowVar(11), rowVar(12), {}}},          // rate         ||w|| <= w_B_max (w_z pinned)        :104-105
        {4, {rowConst(IP_TMAX), rowVar(13), rowVar(14), rowVar(15)}},  // max thrust   ||T|| <= T_max                       :129
        {3, {rowParVar(IP_GIM, 15), rowVar(13), rowVar(14), {}}},      // gimbal       ||T_xy|| <= tan(delta) T_z           :132-133
    };
    static constexpr int NLP = 2;
    static constexpr Row LPS[NLP] = {
        rowLower(0, IP_MDRY),                                                                             // mass >= m_dry        :93
        Row{IP_TMIN, -1., {Term{13, CF_UHAT0, 1.}, Term{14, CF_UHAT1, 1.}, Term{15, CF_UHAT2, 1.}}},     // linearised T_min     :113-125
    };
};

// ------------------------------------------------------------------------------------------------------------
// Rocket2d (rocket2d.cpp:46-84), constrain_initial_final = true (the SC configuration, model.info:55-56).
// w = (r_x, r_y, v_x, v_y, eta, omega, gimbal angle, thrust).
// ------------------------------------------------------------------------------------------------------------
struct Rocket2dSC
{
    static constexpr int MODEL_ID = 1;
    static constexpr int NX = 6, NU = 2, NXV = 6, NUV = 2;
    static constexpr int XMAP[NXV] = {0, 1, 2, 3, 4, 5};
    static constexpr int UMAP[NUV] = {0, 1};
    // equalTo(x_init, X.col(0)), equalTo(x_final, X.rightCols(1)), equalTo(U(0, K-1), 0)   :54-59
    static constexpr unsigned FIXED_FIRST = 0x3Fu;
    static constexpr unsigned FIXED_LAST = 0x3Fu | (1u << 6);
    static constexpr int NCONE = 1;
    static constexpr Cone CONES[NCONE] = {
        {2, {rowParVar(IP_GS, 1), rowVar(0), {}, {}}}, // glide slope  norm(r_x) <= tan(gamma) r_y   :62-64
    };
    static constexpr int NLP = 8;
    static constexpr Row LPS[NLP] = {
        rowUpper(4, IP_TILT), rowBoxLo(4, IP_TILT), // box(-theta_max, eta, theta_max)        :66-68
        rowUpper(5, IP_WMAX), rowBoxLo(5, IP_WMAX), // box(-w_B_max, omega, w_B_max)          :70-72
        rowUpper(6, IP_GIM), rowBoxLo(6, IP_GIM),   // box(-gimbal_max, gimbal, gimbal_max)   :76-78
        rowLower(7, IP_TMIN), rowUpper(7, IP_TMAX), // box(T_min, thrust, T_max)              :80-82
    };
};

// ------------------------------------------------------------------------------------------------------------
// Lander3dof (csrc/model_lander3dof.h): NOT a model of the reference -- this repository's third model, written the way a user's
// addApplicationConstraints would be (the RocketQuat rows without the attitude states).  w = (m, r_x, r_y, r_z, v_x, v_y, v_z, T_x, T_y, T_z).
// ------------------------------------------------------------------------------------------------------------
struct Lander3dofSC
{
    static constexpr int MODEL_ID = 2;
    static constexpr int NX = 7, NU = 3, NXV = 7, NUV = 3;
    static constexpr int XMAP[NXV] = {0, 1, 2, 3, 4, 5, 6};
    static constexpr int UMAP[NUV] = {0, 1, 2};
    // equalTo(X.col(0), x_init); equalTo(X(i, K-1), x_final(i)) for position and velocity (the final mass is free); equalTo(U({0,1}, K-1), 0)
    static constexpr unsigned FIXED_FIRST = 0x7Fu;
    static constexpr unsigned FIXED_LAST = (0x3Fu << 1) | (1u << 7) | (1u << 8);
    static constexpr int NCONE = 3;
    static constexpr Cone CONES[NCONE] = {
        {3, {rowParVar(IP_GS, 3), rowVar(1), rowVar(2), {}}},       // glide slope      ||r_xy|| <= tan(gamma_gs) r_z
        {4, {rowConst(IP_TMAX), rowVar(7), rowVar(8), rowVar(9)}},  // maximum thrust   ||T|| <= T_max
        {3, {rowParVar(IP_GIM, 9), rowVar(7), rowVar(8), {}}},      // thrust pointing  ||T_xy|| <= tan(pointing_max) T_z
    };
    static constexpr int NLP = 2;
    static constexpr Row LPS[NLP] = {
        rowLower(0, IP_MDRY),                                                                         // mass >= m_dry
        Row{IP_TMIN, -1., {Term{7, CF_UHAT0, 1.}, Term{8, CF_UHAT1, 1.}, Term{9, CF_UHAT2, 1.}}},    // uhat . T >= T_min (uhat = (0,0,1) without exact_minimum_thrust)
    };
};

// Zero-order-hold variant of a table (SCProblem.cpp:37-59,116-120 with td.interpolatedInput() == false: K-1 inputs).  The stage
// structure keeps an input slot at every node; at the LAST node the inputs do not exist -- they are pinned (to 0, which is
// also what the device stores in U[K-1]) so that every cone / LP row on them drops out and the trust-region cone of that node
// carries its state part only -- and the table's final-input equalities (`v_U.cols() - 1`) act on node K-2.  C = 0 in the
// dynamics (the zero-order-hold discretisation writes no C).
template <class P>
struct ZeroOrderHold : P
{
    static constexpr bool ZOH = true;
};
// Policy variant of a table (same problem, same record layout): the three most-travelled segment fields stay in the workspace instead of
// living in LDS for the duration of a solve (ipm_kernel.h: SegLdsSlot).  The kernels that run more than two wavefronts per SIMD use it --
// 12 wavefronts x (16.8 KB + 3 KB) do not fit the CU's 160 KB -- and so does every kernel that does not stay resident for a whole solve.
template <class P>
struct SegFieldsInWorkspace : P
{
    static constexpr bool SEG_FIELDS_IN_WORKSPACE = true;
};
template <class P, class = void>
struct SegInLds
{
    static constexpr bool value = true;
};
template <class P>
struct SegInLds<P, decltype(void(P::SEG_FIELDS_IN_WORKSPACE))>
{
    static constexpr bool value = !P::SEG_FIELDS_IN_WORKSPACE;
};
template <class P, class = void>
struct IsZoh
{
    static constexpr bool value = false;
};
template <class P>
struct IsZoh<P, decltype(void(P::ZOH))>
{
    static constexpr bool value = P::ZOH;
};

// ---------------- quantities derived from a table (all constexpr) ----------------
template <class P>
struct Derived
{
    static constexpr bool ZOH = IsZoh<P>::value;
    static constexpr int NVU = P::NXV + P::NUV; // used stage variables; the rest of the tile is identity padding
    static_assert(NVU <= NV, "a node's free states + inputs must fit one 16-wide MFMA tile");
    static_assert(P::NCONE <= MAXCONES && P::NLP <= MAXLP, "table too large");
    static constexpr int TD = NV + 1;           // trust-region cone (delta_k ; wbar - w), zero rows for padding variables
    static constexpr int C1 = 0;
    static constexpr int coneDim(int c) { return c == 0 ? TD : P::CONES[c - 1].dim; } // cone 0 = trust region
    static constexpr int coneOff(int c)
    {
        int o = 0;
        for (int i = 0; i < c; i++)
            o += coneDim(i);
        return o;
    }
    static constexpr int NCONES = P::NCONE + 1;
    static constexpr int LP0 = coneOff(NCONES); // first LP row
    static constexpr int NS = LP0 + P::NLP;     // slack entries per node
    static constexpr unsigned PAD_MASK = (NVU >= NV) ? 0u : (((1u << NV) - 1u) & ~((1u << NVU) - 1u));
    static constexpr int NL = P::NX;            // dynamics rows per segment

    static constexpr int xinv(int i)
    {
        for (int j = 0; j < P::NXV; j++)
            if (P::XMAP[j] == i)
                return j;
        return -1;
    }
    static constexpr int uinv(int i)
    {
        for (int j = 0; j < P::NUV; j++)
            if (P::UMAP[j] == i)
                return P::NXV + j;
        return -1;
    }
    // XMAP / UMAP are the identity for both shipped models: a stage variable's A / B column is then plain arithmetic
    static constexpr bool mapsAreIdentity()
    {
        for (int j = 0; j < P::NXV; j++)
            if (P::XMAP[j] != j)
                return false;
        for (int j = 0; j < P::NUV; j++)
            if (P::UMAP[j] != j)
                return false;
        return true;
    }
    static constexpr bool IDENTITY_MAPS = mapsAreIdentity();
    // the same lookups as constant tables (device code indexes them with unrolled loop counters)
    struct Arr16
    {
        int v[16];
    };
    static constexpr Arr16 mkXinv()
    {
        Arr16 a{};
        for (int i = 0; i < 16; i++)
            a.v[i] = i < P::NX ? xinv(i) : -1;
        return a;
    }
    static constexpr Arr16 mkUinv()
    {
        Arr16 a{};
        for (int i = 0; i < 16; i++)
            a.v[i] = i < P::NU ? uinv(i) : -1;
        return a;
    }
    static constexpr Arr16 mkConeOff()
    {
        Arr16 a{};
        for (int i = 0; i < 16; i++)
            a.v[i] = i <= NCONES ? coneOff(i) : 0;
        return a;
    }
    static constexpr Arr16 mkConeDim()
    {
        Arr16 a{};
        for (int i = 0; i < 16; i++)
            a.v[i] = i < NCONES ? coneDim(i) : 0;
        return a;
    }
    static constexpr Arr16 XINV = mkXinv(), UINV = mkUinv(), CONE_OFF = mkConeOff(), CONE_DIM = mkConeDim();
    // presolved (constant) stage variables at node k
    static constexpr unsigned INPUT_MASK = ((1u << NVU) - 1u) & ~((1u << P::NXV) - 1u);
    // what the table pins at the last node (first-order hold) / at the last node and the last INPUT node K-2 (zero-order hold)
    static constexpr unsigned FIX_LAST = ZOH ? ((P::FIXED_LAST & ~INPUT_MASK) | INPUT_MASK) : P::FIXED_LAST;
    static constexpr unsigned FIX_PRE = ZOH ? (P::FIXED_LAST & INPUT_MASK) : 0u;
    __host__ __device__ static constexpr unsigned fixedMask(int k, int K)
    {
        return PAD_MASK | (k == 0 ? P::FIXED_FIRST : 0u) | (k == K - 1 ? FIX_LAST : 0u) | (k == K - 2 ? FIX_PRE : 0u);
    }
    static constexpr bool rowFree(const Row &r, unsigned fm)
    {
        for (int t = 0; t < 3; t++)
            if (r.t[t].var >= 0 && !(fm & (1u << r.t[t].var)))
                return true;
        return false;
    }
    // bit c: cone c (0 = trust region) is active; bit NCONES + l: LP row l.  A cone all of whose variables are presolved at a
    // node is a constant there and is dropped (oracle/structured_ipm.hpp: activeMask).
    static constexpr unsigned activeFor(unsigned fm)
    {
        unsigned a = 1u;
        for (int c = 0; c < P::NCONE; c++)
        {
            bool fr = false;
            for (int i = 0; i < P::CONES[c].dim; i++)
                fr = fr || rowFree(P::CONES[c].r[i], fm);
            if (fr)
                a |= 1u << (c + 1);
        }
        for (int l = 0; l < P::NLP; l++)
            if (rowFree(P::LPS[l], fm))
                a |= 1u << (NCONES + l);
        return a;
    }
    static constexpr unsigned ACT_FIRST = activeFor(PAD_MASK | P::FIXED_FIRST), ACT_LAST = activeFor(PAD_MASK | FIX_LAST),
                              ACT_MID = activeFor(PAD_MASK), ACT_BOTH = activeFor(PAD_MASK | P::FIXED_FIRST | FIX_LAST),
                              ACT_PRE = activeFor(PAD_MASK | FIX_PRE), ACT_FIRST_PRE = activeFor(PAD_MASK | P::FIXED_FIRST | FIX_PRE);
    __host__ __device__ static constexpr unsigned activeMask(int k, int K)
    {
        return (k == 0 && k == K - 1) ? ACT_BOTH
               : k == 0               ? (K == 2 ? ACT_FIRST_PRE : ACT_FIRST)
               : k == K - 1           ? ACT_LAST
               : k == K - 2           ? ACT_PRE
                                      : ACT_MID;
    }
    // number of active cone-program rows "D" (degree of the product cone) contributed by node k: one per cone, one per LP row
    static constexpr int popcount(unsigned v)
    {
        int n = 0;
        for (; v; v >>= 1)
            n += int(v & 1u);
        return n;
    }

    // sparsity pattern of  sum_c L_c' W_c^-2 L_c  over the application cones and LP rows: index of entry (a, b) in the
    // node's small-block record, -1 outside the pattern
    struct Pattern
    {
        int idx[NV][NV];
        int n;
    };
    static constexpr Pattern buildPattern()
    {
        Pattern p{};
        bool m[NV][NV] = {};
        for (int c = 0; c < P::NCONE; c++)
            for (int a = 0; a < P::CONES[c].dim; a++)
                for (int b = 0; b < P::CONES[c].dim; b++)
                    for (int ta = 0; ta < 3; ta++)
                        for (int tb = 0; tb < 3; tb++)
                        {
                            const int va = P::CONES[c].r[a].t[ta].var, vb = P::CONES[c].r[b].t[tb].var;
                            if (va >= 0 && vb >= 0)
                                m[va][vb] = true;
                        }
        for (int l = 0; l < P::NLP; l++)
            for (int ta = 0; ta < 3; ta++)
                for (int tb = 0; tb < 3; tb++)
                {
                    const int va = P::LPS[l].t[ta].var, vb = P::LPS[l].t[tb].var;
                    if (va >= 0 && vb >= 0)
                        m[va][vb] = true;
                }
        p.n = 0;
        for (int a = 0; a < NV; a++)
            for (int b = 0; b < NV; b++)
                p.idx[a][b] = m[a][b] ? p.n++ : -1;
        return p;
    }
    static constexpr Pattern PAT = buildPattern();
    static constexpr int HS_N = PAT.n;
};

} // namespace ipm
} // namespace scpp
