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
// staged in LDS.  Every lane owns entries of ONE row of V, so it needs one row of the Jacobian per stage: by default each
// lane evaluates the non-zeros of its row from the model's generated analytic rows (Model::JacobianRows, the build-time
// analogue of the reference's CppADCodeGen step; 61 non-zeros of 252 for RocketQuat).  -DDISC_AD_JACOBIAN selects the
// generic path instead: forward-mode AD, lane j < NX+NU evaluates systemFlowMap<Dual1> with seed e_j (one Jacobian
// column per lane) and the tile is exchanged through LDS.
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

// RKF78 steps per segment.  The reference takes exactly 5 whatever the segment length (integrate_adaptive(stepper, ode, V, 0., dt,
// dt / 5.) with an uncontrolled stepper, discretizationImplementation.hpp:141,154): at its own shipped RocketQuat configuration
// (K = 15, 12 s) that is a step of 0.171 s and leaves ~1e-13 relative in A .. z.  At K = 50 the same 5 steps are 0.049 s long -- an
// order-8 scheme then integrates to round-off four times over.  The kernel can therefore take (OPT-IN since round 4: the default
// of a context is the reference's 5 -- SCvx accept / reject decisions hinge on the sign of a dJ of ~1e-10, and a 1e-13 change of
// A .. z flips a handful of them)
//      n = clamp(ceil(segment length / DISC_MAX_STEP), 1, DISC_STEPS_MAX)     (wave-uniform, per instance and call)
// steps: never more than the reference's 5, and never a step longer than the reference's own at the configuration it ships --
// K = 50, 12 s: 2 steps of 0.122 s, A .. z within 1e-13 of the 5-step result (measured: 3 steps 3.5e-15, 2 steps 1.0e-13, 1 step
// 2.4e-11; K = 15: 5 steps as the reference); test-enforced against the DOP853 goldens at 1e-9 like before, and against the 5-step
// kernel at 1e-11.  40 % of the stage evaluations at K = 50: discretize_kernel 4.2 -> 1.9 ms per launch, headline +4.2 % (same box).
// scpp_hip_set_discretization_steps(ctx, n) sets the count at run time (n = 5: the reference's scheme literally, the default; 0 = this rule);
// -DDISC_STEPS=n pins it at compile time.
#ifndef DISC_STEPS
#define DISC_STEPS 0
#endif
#ifndef DISC_MAX_STEP
#define DISC_MAX_STEP (12. / (14. * 5.)) // seconds: shipped RocketQuat SC.info (K = 15) on the 12 s scenario, 5 steps per segment
#endif
constexpr int DISC_STEPS_MAX = 5;
#ifndef DISC_WAVES_PER_SIMD
#define DISC_WAVES_PER_SIMD 2
#endif

template <class Model, bool FOH, bool VT>
struct DiscLayout
{
    static constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP;
    static constexpr int NJ = NX + NU;                                  // Jacobian columns
    // integrated columns [x | Phi | Psi_B | Psi_C | psi_s]; psi_z is NOT integrated: Runge-Kutta methods
    // commute with affine maps of the state, so  z = x(dt) - A x0 - B u0 - C u1 - s sigma  reproduces the
    // integrated psi_z to round-off (the defect ODE of that combination is exactly the reference's z ODE)
    static constexpr int NCOLS = 1 + NX + NU + (FOH ? NU : 0) + (VT ? 1 : 0);
    static constexpr int NENT = NX * NCOLS;
    static constexpr int NG = WAVE / 16;                                // lane groups: lane = g * 16 + row (MFMA operand rows)
    static constexpr int EPL = (NCOLS + NG - 1) / NG;                   // columns per lane: col = m * NG + g
    // column order [x | psi_s | Phi | Psi_B | Psi_C]: with NG = 4 groups (RocketQuat) the B and the C columns each
    // fill one column slot m exactly, so their forcing terms need no per-lane selection
    static constexpr int COL_S = 1, COL_PHI = 1 + (VT ? 1 : 0), COL_B = COL_PHI + NX, COL_C = COL_B + NU;
    static constexpr int NJP = (NJ + 1) & ~1;                           // row pitch of the Jacobian tile (16-byte rows)
};

// X [B][K][NX], U [B][K][NU] (FOH) , sigma [B], par [B][NP]  ->  A [B][K-1][NX][NX], Bm, C [B][K-1][NX][NU],
// S, Z [B][K-1][NX]   (row-major blocks).  active[B] (may be null): skip instances with active == 0.
// LDS of one segment integration (one wavefront): stage values, the per-stage-time table of the input, the operand vector of the Jacobian
// table, the wave-uniform constants.  A struct, so that its home can be chosen by the caller: a __shared__ object of its own in
// discretize_kernel, the dynamic LDS region it shares with the solver's LDS-resident segment fields in the persistent SCvx kernel (the two are
// never live at the same time; 9.9 + 16.8 KB next to each other would cost that kernel three of its eight wavefronts per CU).
template <class Model, bool FOH, bool VT>
struct DiscLds
{
    using L = DiscLayout<Model, FOH, VT>;
    __attribute__((aligned(16))) double Ys[32 * L::NX + 2];
#ifdef DISC_AD_JACOBIAN
    __attribute__((aligned(16))) double Jm[L::NX * L::NJP]; // [sigma*A | sigma*B] row-major
    double fv[L::NX];                                       // f(x,u) (unscaled)
    double cst[L::NP + 2 * L::NU + 1];
#else
    double uh[DISC_STEPS_MAX * RK_S * (Model::JacobianRows::NUAUX + 1 + L::NU)];
    __attribute__((aligned(16))) double Wt[Model::JacobianTable::NW];
    double cst[L::NP + 2 * L::NU + Model::JacobianRows::NAUX + 1];
#endif
};

// The integration of ONE segment by one wavefront: the body of discretize_kernel, and of the discretisation step of the persistent SCvx
// kernel (scvx_persistent.h), where one wavefront walks through the K - 1 segments of its instance.
template <class Model, bool FOH, bool VT>
__device__ __forceinline__ void discretizeSegment(int B, int K, const double *__restrict__ X, const double *__restrict__ U,
                                                  const double *__restrict__ sigma, const double *__restrict__ par, int par_stride,
                                                  const int *__restrict__ active, double *__restrict__ Aout, double *__restrict__ Bout,
                                                  double *__restrict__ Cout, double *__restrict__ Sout, double *__restrict__ Zout, int steps_opt,
                                                  const long inst, const int k, DiscLds<Model, FOH, VT> *lds)
{
    using L = DiscLayout<Model, FOH, VT>;
    constexpr int NX = L::NX, NU = L::NU, NP = L::NP, NJ = L::NJ, NJP = L::NJP, NCOLS = L::NCOLS, NG = L::NG, EPL = L::EPL;

    // (J V)' = V' J' on the matrix core: V' is the A operand of v_mfma_f64_16x16x4_f64, read straight from Ys (two column
    // tiles; the contraction index runs to 16, so Ys is zero-filled once and padded to 32 columns), J' the B operand, taken
    // from the Jacobian row each lane holds in registers.  No staging, no extra synchronisation.
    static_assert(NX <= 16 && NCOLS + NG <= 32 && NG == 4, "one 16-row tile of states, two 16-column tiles of V");
    auto &Ys = lds->Ys;
    // flow-map parameters and the segment's two input nodes: wave-uniform values that the 13x5 row evaluations need.
    // Kept in LDS and re-read inside every evaluation: held in registers across the stage loop they were
    // spilled to scratch, and their serialised reloads (11 round trips per evaluation) dominated the evaluation phase.
#ifndef DISC_AD_JACOBIAN
    constexpr int NAUX = Model::JacobianRows::NAUX;   // parameter-only sub-expressions of the analytic rows
    constexpr int NUAUX = Model::JacobianRows::NUAUX; // input-only sub-expressions, tabulated per (step, stage)
    constexpr int UHP = NUAUX + 1 + NU;           // per stage time: input-only sub-expressions, t / dt, u(t)
    auto &uh = lds->uh;
#define DISC_TABLE_ROWS 1
    // Jacobian entries as a lane-parallel table (Model::JacobianTable): one output per lane per pass instead of one divergent
    // `case` per row; W holds the operands, the partial sums and the outputs [J | f]
    using TB = typename Model::JacobianTable;
    auto &Wt = lds->Wt;
#else
    constexpr int NAUX = 0;
    auto &Jm = lds->Jm;
    auto &fv = lds->fv;
#endif
    auto &cst = lds->cst;

    const int lane = threadIdx.x;
    for (int i = lane; i < 32 * NX + 2; i += WAVE)
        Ys[i] = 0.;
    const int nseg = K - 1;
    if (inst >= B)
        return;
    if (active && active[inst] == 0)
        return;

    const double sg = sigma[inst];
    const double dt = VT ? 1. / double(K - 1) : sg / double(K - 1);
    const double tscale = VT ? sg : 1.;
    // steps of this segment (see DISC_STEPS above); the segment lasts sg / (K - 1) seconds in both time parametrisations
    int nsteps = DISC_STEPS > 0 ? DISC_STEPS : steps_opt; // scpp_hip_set_discretization_steps: 0 = the rule, 1 .. 5 = pinned
    if (nsteps <= 0)
    {
        const double seg_seconds = fabs(sg) / double(K - 1);
        nsteps = int(ceil(seg_seconds / DISC_MAX_STEP));
        nsteps = nsteps < 1 ? 1 : (nsteps > DISC_STEPS_MAX ? DISC_STEPS_MAX : nsteps);
    }
    nsteps = uniformInt(nsteps);
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

    if (lane == 0)
    {
#pragma unroll
        for (int i = 0; i < NP; i++)
            cst[i] = p[i];
#pragma unroll
        for (int i = 0; i < NU; i++)
        {
            cst[NP + i] = u0[i];
            cst[NP + NU + i] = u1[i];
        }
#ifndef DISC_AD_JACOBIAN
        double aux0[NAUX];
        Model::JacobianRows::prepare(p, aux0);
#pragma unroll
        for (int i = 0; i < NAUX; i++)
            cst[NP + 2 * NU + i] = aux0[i];
#endif
    }
#ifndef DISC_AD_JACOBIAN
    {
        // the input is a known function of time: u(t) at each of the nsteps x 13 stage times and what the rows need of it alone
        // (|T|, 1/|T| for RocketQuat) are computed once here, one stage time per lane
        const double hh = dt / double(nsteps);
        for (int e = lane; e < nsteps * RK_S; e += WAVE)
        {
            const double tse = double(e / RK_S) * hh + RK_C[e % RK_S] * hh;
            const double fre = FOH ? tse / dt : 0.;
            double ue[NU], ua[NUAUX];
#pragma unroll
            for (int i = 0; i < NU; i++)
                ue[i] = u0[i] + fre * (u1[i] - u0[i]);
            Model::JacobianRows::prepareInput(ue, p, ua);
#pragma unroll
            for (int i = 0; i < NUAUX; i++)
                uh[e * UHP + i] = ua[i];
            uh[e * UHP + NUAUX] = fre;
#pragma unroll
            for (int i = 0; i < NU; i++)
                uh[e * UHP + NUAUX + 1 + i] = ue[i];
        }
    }
#endif
#ifdef DISC_TABLE_ROWS
    for (int i = lane; i < TB::NW; i += WAVE)
        Wt[i] = (i == TB::W_ONE) ? 1. : 0.;
#endif
    WAVE_SYNC();
#ifdef DISC_TABLE_ROWS
    if (lane == 0)
    {
#pragma unroll
        for (int i = 0; i < NP; i++)
            Wt[TB::W_PAR + i] = p[i];
#pragma unroll
        for (int i = 0; i < NAUX; i++)
            Wt[TB::W_AUX + i] = cst[NP + 2 * NU + i];
    }
    // this lane's two table slots: coefficients and the LDS byte offsets of the factors (two 16-bit offsets per word)
    double tcf[2][TB::MAXMON];
    unsigned tof[2][TB::MAXMON][2];
    int ttg[2];
#pragma unroll
    for (int ps = 0; ps < 2; ps++)
    {
        const int slot = ps * 64 + lane;
        ttg[ps] = TB::target(slot);
#pragma unroll
        for (int q = 0; q < TB::MAXMON; q++)
        {
            tcf[ps][q] = TB::coef(slot, q);
            tof[ps][q][0] = unsigned(TB::factor(slot, q, 0) * 8) | (unsigned(TB::factor(slot, q, 1) * 8) << 16);
            tof[ps][q][1] = unsigned(TB::factor(slot, q, 2) * 8) | (unsigned(TB::factor(slot, q, 3) * 8) << 16);
        }
    }
    WAVE_SYNC();
#endif
    // Lane (g, row) owns entries (row, col = m*NG + g), m = 0..EPL-1: it needs ONE row of the Jacobian tile per
    // stage (kept in registers for all its columns) and one column of V per entry.
    // lane = g * 16 + row: the layout in which v_mfma_f64_16x16x4_f64 delivers (J V)' -- lane (g, row), accumulator register
    // r holds the derivative of entry (row, col = g + 4 r), i.e. exactly the entries m = r this lane integrates
    const int row = lane & 15, g = lane >> 4;
    const bool lane_on = row < NX;
#ifndef SCPP_HIP_EMU
    __builtin_assume(g >= 0 && g < NG);
#endif
    const double x0r = lane_on ? X[(inst * K + k) * NX + row] : 0.; // (rows >= NX would read past the node: past the buffer at the last one)
    double y[EPL];
    bool eon[EPL];
#pragma unroll
    for (int m = 0; m < EPL; m++)
    {
        const int c = m * NG + g;
        eon[m] = lane_on && c < NCOLS;
        double v = 0.;
        if (c == 0)
            v = x0r;
        else if (c >= L::COL_PHI && c < L::COL_PHI + NX)
            v = (c - L::COL_PHI == row) ? 1. : 0.;
        y[m] = eon[m] ? v : 0.;
    }

    double kk[RK_S][EPL];
    const double h = dt / double(nsteps);
#ifdef DISC_PROFILE
    long long tA = 0, tB = 0, tC = 0;
    const long long tk0 = clock64();
#endif

    for (int step = 0; step < nsteps; step++)
    {
        const double t0 = double(step) * h;
        // The 13 stages are instantiated with a COMPILE-TIME stage index: the tableau entries become
        // immediates, zero coefficients vanish and kk[][] stays in registers (a rolled stage loop indexes it
        // dynamically, i.e. from scratch memory, and tests the coefficients at run time).
        forEachStage([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // ---- stage value ys = y + h * sum_j a_sj k_j ----
#ifdef DISC_PROFILE
            const long long p0 = clock64();
#endif
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
                if (eon[m])
                    Ys[(m * NG + g) * NX + row] = ys;
#ifdef DISC_TABLE_ROWS
                if (m == 0 && g == 0 && lane_on)
                    Wt[TB::W_X + row] = ys; // the state column is also operand x of the Jacobian table
#endif
            }
            WAVE_SYNC();
#ifdef DISC_PROFILE
            const long long p1 = clock64();
            tA += p1 - p0;
#endif
#ifndef DISC_AD_JACOBIAN
            const double frac = FOH ? uh[(step * RK_S + s) * UHP + NUAUX] : 0.; // t / dt, tabulated with the input
#else
            const double frac = FOH ? ts / dt : 0.;
#endif
            double jr[NJ]; // my row of [sigma*A | sigma*B]
            double fr;     // f[row] (unscaled)
#ifdef DISC_TABLE_ROWS
            double bsel[4];
            {
                const int e = step * RK_S + s;
                // operands that change with the stage: u(t), the input-only terms, the state-dependent terms
                if (lane < NU)
                    Wt[TB::W_U + lane] = uh[e * UHP + NUAUX + 1 + lane];
                else if (lane < NU + NUAUX)
                    Wt[TB::W_UAUX + lane - NU] = uh[e * UHP + lane - NU];
                {
                    double hh[TB::NH];
                    TB::evalHoists(Ys, uh + e * UHP + NUAUX + 1, cst, cst + NP + 2 * NU, uh + e * UHP, hh);
                    if (lane == 0)
                    {
#pragma unroll
                        for (int i = 0; i < TB::NH; i++)
                            Wt[TB::W_H + i] = hh[i];
                    }
                }
                WAVE_SYNC();
                const char *wb = reinterpret_cast<const char *>(Wt);
#pragma unroll
                for (int ps = 0; ps < 2; ps++)
                {
                    double val = 0.;
#pragma unroll
                    for (int q = 0; q < (ps == 0 ? TB::MAXMON : TB::MAXMON_B); q++)
                    {
                        const double f0 = *reinterpret_cast<const double *>(wb + (tof[ps][q][0] & 0xFFFFu));
                        const double f1 = *reinterpret_cast<const double *>(wb + (tof[ps][q][0] >> 16));
                        const double f2 = *reinterpret_cast<const double *>(wb + (tof[ps][q][1] & 0xFFFFu));
                        const double f3 = *reinterpret_cast<const double *>(wb + (tof[ps][q][1] >> 16));
                        val += ((tcf[ps][q] * f0) * f1) * (f2 * f3);
                    }
                    if (ttg[ps] >= 0)
                        Wt[ttg[ps]] = val;
                    WAVE_SYNC();
                }
                const int rowc = lane_on ? row : 0;
#pragma unroll
                for (int t = 0; t < 4; t++)
                {
                    const double v = Wt[TB::W_J + rowc * NJ + (4 * t + g < NX ? 4 * t + g : 0)];
                    bsel[t] = (4 * t + g < NX) ? v : 0.; // columns >= NX belong to the inputs
                }
                fr = Wt[TB::W_F + rowc];
            }
#elif !defined(DISC_AD_JACOBIAN)
            {
                // ---- analytic non-zeros of my Jacobian row at the stage point (no exchange through LDS) ----
                int zo = 0; // opaque zero offset: keeps the LDS reads of the wave-uniform constants inside the stage (see cst)
#ifndef SCPP_HIP_EMU
                asm volatile("" : "+v"(zo));
#endif
                const double *cv = cst + zo;
                double pl[NP], xs[NX], us[NU], ax[NAUX], ux[NUAUX];
#pragma unroll
                for (int i = 0; i < NUAUX; i++)
                    ux[i] = uh[(step * RK_S + s) * UHP + i];
#pragma unroll
                for (int i = 0; i < NP; i++)
                    pl[i] = cv[i];
#pragma unroll
                for (int i = 0; i < NAUX; i++)
                    ax[i] = cv[NP + 2 * NU + i];
#pragma unroll
                for (int i = 0; i < NX; i++)
                    xs[i] = Ys[i];
#pragma unroll
                for (int i = 0; i < NU; i++)
                {
                    const double a0 = cv[NP + i], a1 = cv[NP + NU + i];
                    us[i] = a0 + frac * (a1 - a0);
                }
                fr = Model::JacobianRows::row(row, xs, us, pl, ax, ux, jr);
            }
#else
            // ---- Jacobian tile by forward-mode AD, one seed direction per lane ----
            if (lane < NJ)
            {
                Dual1 xd[NX], ud[NU], fd[NX];
                // opaque zero offset: keeps the compiler from hoisting these LDS reads out of the stage loop (and
                // spilling the values again)
                int zo = 0;
#ifndef SCPP_HIP_EMU
                asm volatile("" : "+v"(zo));
#endif
                const double *cv = cst + zo;
                double pl[NP];
#pragma unroll
                for (int i = 0; i < NP; i++)
                    pl[i] = cv[i];
#pragma unroll
                for (int i = 0; i < NX; i++)
                    xd[i] = Dual1(Ys[i], (lane == i) ? 1. : 0.);
#pragma unroll
                for (int i = 0; i < NU; i++)
                {
                    const double a0 = cv[NP + i], a1 = cv[NP + NU + i];
                    ud[i] = Dual1(a0 + frac * (a1 - a0), (lane == NX + i) ? 1. : 0.);
                }
                Model::template systemFlowMap<Dual1>(xd, ud, pl, fd);
#pragma unroll
                for (int i = 0; i < NX; i++)
                    Jm[i * NJP + lane] = tscale * fd[i].d;
                if (lane == 0)
                {
#pragma unroll
                    for (int i = 0; i < NX; i++)
                        fv[i] = fd[i].v;
                }
            }
            WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < NJ; j++)
                jr[j] = Jm[row * NJP + j];
            fr = fv[row];
#endif
#ifdef DISC_PROFILE
            const long long p2 = clock64();
            tB += p2 - p1;
#endif
            // ---- derivative of the owned entries: d(row, c) = J[row,:] V[:,c] + forcing(row, c), branch-free ----
            {
                const double alphaB = FOH ? (1. - frac) : 1.;
                d4_t acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
#pragma unroll
                for (int t = 0; t < 4; t++)
                {
                    const int kx = g + 4 * t; // contraction index = state j
#ifdef DISC_TABLE_ROWS
                    double b = bsel[t];
#else
                    double b = (4 * t < NX) ? jr[4 * t < NX ? 4 * t : 0] : 0.; // J[row][kx]: pick jr[4t + g]; columns >= NX are the inputs' -> 0
#pragma unroll
                    for (int q = 1; q < 4; q++)
                        b = (g == q) ? ((4 * t + q < NX) ? jr[4 * t + q < NX ? 4 * t + q : 0] : 0.) : b;
#endif
#ifndef DISC_AD_JACOBIAN
                    b *= tscale; // sigma scaling of A applied to the four entries this lane contributes
#endif
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ys[row * NX + kx], b, acc0, 0, 0, 0);
                    if (NCOLS > 16)
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ys[(16 + row) * NX + kx], b, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < EPL; m++)
                {
                    const int c = m * NG + g;
                    const double acc = m < 4 ? acc0[m < 4 ? m : 0] : acc1[m >= 4 ? m - 4 : 0];
                    // forcing: B columns J[row, NX+j] * alpha ; C columns J[row, NX+j] * frac ; s column f ; x column: sigma f only.
                    // Which kinds can occur in slot m is a compile-time fact (m is a constant after unrolling).
                    double d = acc;
                    const bool slotB = m * NG < L::COL_B + NU && m * NG + NG > L::COL_B;
                    const bool slotC = FOH && m * NG < L::COL_C + NU && m * NG + NG > L::COL_C;
                    if (slotB || slotC)
                    {
                        const bool isB = c >= L::COL_B && c < L::COL_B + NU;
                        const bool isC = FOH && c >= L::COL_C && c < L::COL_C + NU;
                        const int jj = isB ? c - L::COL_B : isC ? c - L::COL_C : 0;
                        const double w = isB ? alphaB : isC ? frac : 0.;
#ifdef DISC_TABLE_ROWS
                        const double jb = Wt[TB::W_J + (lane_on ? row : 0) * NJ + NX + jj]; // J[row, NX + jj], jj is lane dependent
#else
                        double jb = jr[NX]; // J[row, NX + jj], jj is lane dependent
#pragma unroll
                        for (int q = 1; q < NU; q++)
                            jb = (jj == q) ? jr[NX + q] : jb;
#endif
#ifndef DISC_AD_JACOBIAN
                        d += (w * tscale) * jb; // sigma scaling of B
#else
                        d += w * jb;
#endif
                    }
                    if (VT && m * NG <= L::COL_S && m * NG + NG > L::COL_S)
                        d = (c == L::COL_S) ? d + fr : d;
                    if (m == 0)
                        d = (c == 0) ? tscale * fr : d;
                    kk[s][m] = d;
                }
            }
            WAVE_SYNC();
#ifdef DISC_PROFILE
            tC += clock64() - p2;
#endif
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

#ifdef DISC_PROFILE
    if (blockIdx.x == 4096 && lane == 0)
        printf("disc profile (cycles/segment): stage-value %lld  AD %lld  product %lld  total %lld\n", tA, tB, tC, (long long)(clock64() - tk0));
#endif
    // ---- write A_k, B_k, C_k, s_k ; z_k from the affine identity ----
    const long seg = inst * nseg + k;
#pragma unroll
    for (int m = 0; m < EPL; m++)
    {
        const int c = m * NG + g;
        if (!eon[m])
            continue;
        Ys[c * NX + row] = y[m];
        if (c >= L::COL_PHI && c < L::COL_PHI + NX)
            Aout[seg * NX * NX + row * NX + (c - L::COL_PHI)] = y[m];
        else if (c >= L::COL_B && c < L::COL_B + NU)
            Bout[seg * NX * NU + row * NU + (c - L::COL_B)] = y[m];
        else if (FOH && c >= L::COL_C && c < L::COL_C + NU)
            Cout[seg * NX * NU + row * NU + (c - L::COL_C)] = y[m];
        else if (VT && c == L::COL_S)
            Sout[seg * NX + row] = y[m];
    }
    // stash x0 behind the integrated columns for the identity
    if (lane < NX)
        Ys[NCOLS * NX + lane] = x0r;
    WAVE_SYNC();
    if (lane < NX)
    {
        double z = Ys[lane]; // x(dt)
#pragma unroll
        for (int j = 0; j < NX; j++)
            z -= Ys[(L::COL_PHI + j) * NX + lane] * Ys[NCOLS * NX + j];
#pragma unroll
        for (int j = 0; j < NU; j++)
        {
            z -= Ys[(L::COL_B + j) * NX + lane] * u0[j];
            if (FOH)
                z -= Ys[(L::COL_C + j) * NX + lane] * u1[j];
        }
        if (VT)
            z -= Ys[L::COL_S * NX + lane] * sg;
        Zout[seg * NX + lane] = z;
    }
}

template <class Model, bool FOH, bool VT>
__global__ void __launch_bounds__(WAVE, DISC_WAVES_PER_SIMD)
    discretize_kernel(int B, int K, const double *__restrict__ X, const double *__restrict__ U,
                      const double *__restrict__ sigma, const double *__restrict__ par, int par_stride,
                      const int *__restrict__ active,
                      double *__restrict__ Aout, double *__restrict__ Bout, double *__restrict__ Cout,
                      double *__restrict__ Sout, double *__restrict__ Zout, int steps_opt)
{
    // XCD-aware block -> (instance, segment) map: blocks b, b+8, b+16.. share an XCD (and its L2), so
    // give one XCD all K-1 segments of an instance (they re-read the same X/U/par lines).
    const int nseg = K - 1;
    const long b = blockIdx.x;
    const long xcd = b & 7, gb = b >> 3;
    const long inst = (gb / nseg) * 8 + xcd;
    const int k = int(gb % nseg);
    __shared__ DiscLds<Model, FOH, VT> lds;
    discretizeSegment<Model, FOH, VT>(B, K, X, U, sigma, par, par_stride, active, Aout, Bout, Cout, Sout, Zout, steps_opt, inst, k, &lds);
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
