// Batched linear MPC solve: ONE wavefront per controller instance.
//
// Reference path: MPCAlgorithm::solve -> ECOSSolver::solve on buildMPCProblem + Rocket2d::addApplicationConstraints
// (scpp_core/src/MPCAlgorithm.cpp:95-139, MPCProblem.cpp:6-87, scpp_models/src/rocket2d.cpp:46-84).  With
// constant_dynamics (MPC.info:5) the linear model x+ = A x + B u + z is the same for every solve, so the states are
// eliminated once on the host (mpc_setup.h) and each solve is an inequality-only cone program in
//   v = [u_0 .. u_{N-1} | input_cost | error_cost] / D          (nv = 2N + 2 <= 16 variables, N = K - 1 <= 7)
//   min c'v   s.t.  s = h - G v in K,   h = c0 + P x_init + Q x_final
// whose normal matrix H = G' W^-2 G is a single 16 x 16 FP64 tile: Mehrotra predictor-corrector with Nesterov-Todd
// scaling, one tile Cholesky (tile_engine.h: invCholFactor) and two tile solves per iteration.
//
// Data layout inside the wavefront: constraint row r lives in lane r & 63, slot r >> 6.
//   slot 0: the 8N box rows (per stage +-tilt, +-rate; per input +-gimbal, thrust lo/hi)            -> pure LP arithmetic
//   slot 1: lanes 2c, 2c+1 glide-slope cone of stage c+1 | lanes 16..22 error cone | lanes 32..32+2N input cone
//           (every cone inside one group of 16 lanes: cone sums are 4 xor-shuffles)
// G rows stay in registers (2 x 16 doubles per lane); vectors of variables travel through 16 doubles of LDS; W^-1 G is
// staged through LDS (64 rows at a time) into MFMA operand layout for H (24-28 x v_mfma_f64_16x16x4_f64).  No HBM traffic besides
// x_init in / U, X out.  Scalar twin: oracle/mpc.hpp (MpcCondensedIpm).
#pragma once
#include "tile_engine.h"

#ifndef MPC_SPLIT_STEPS
#define MPC_SPLIT_STEPS 1 // primal and dual step lengths of their own (round 6); 0: ECOS's common one
#endif

namespace scpp
{
namespace mpc
{

constexpr int NX = 6, NU = 2, NV = 16, KMAX = 8, NMAX = 7, ROWS = 128, LROWS = 112, GP = 17;
constexpr int ERR_LANE = 16, INP_LANE = 32;

struct MpcConst
{
    int K, N, nv, nlp;
    int maxit, pad0, pad1, pad2;
    double tan_gs, theta_max, w_max;
    double feastol, abstol, reltol;
    double D[NV], c[NV];
    double G[ROWS][NV];
    double P[ROWS][NX], Q[ROWS][NX], c0[ROWS];
    double Li0[NV * NV], Li0T[NV * NV]; // inverse Cholesky factor of G'G (identity padded), row-major, and its transpose
    double Phi[KMAX][NX][NX], Gam[KMAX][NMAX][NX][NU], zeta[KMAX][NX];
    double A[NX * NX], B[NX * NU], z[NX]; // exactLinearDiscretization output (parity tests)
};

struct Shared
{
    ipm::TileShared ts;
    double gt[64 * GP]; // W^-1 G rows of one slot at a time
    double vec[NV];
    double bk[NV];
    double red[NV];
};

using ipm::Tile;

// sum over the lanes of my cone (slot 1); v must be 0 on lanes that hold no row.  Glide-slope pairs sit in the first row
// of 16 lanes, the error and input cones own a row each.
__device__ __forceinline__ double csum(double v, int lane)
{
    const double p1 = v + rowXor1(v);
    double r = p1 + rowXor2(p1);
    r += rowHalfMirror(r);
    r += rowMirror(r);
    return lane < 16 ? p1 : r; // symmetric butterfly: bitwise the same sum in every lane of the cone
}

// 16 per-lane partial sums -> the wave total of entry j ends up in lanes 4j .. 4j+3 (17 shuffles instead of 96)
__device__ inline double waveReduce16(const double (&p)[NV], int lane)
{
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    double a[8], b[4], c[2];
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        const double mine = b5 ? p[8 + q] : p[q], send = b5 ? p[q] : p[8 + q];
        a[q] = mine + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const double mine = b4 ? a[4 + q] : a[q], send = b4 ? a[q] : a[4 + q];
        b[q] = mine + __shfl_xor(send, 16);
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
    {
        const double mine = b3 ? b[2 + q] : b[q], send = b3 ? b[q] : b[2 + q];
        c[q] = mine + rowRor8(send); // rotation by 8 in a row of 16 = partner lane ^ 8
    }
    const double mine = b2 ? c[1] : c[0], send = b2 ? c[0] : c[1];
    double d = mine + __shfl_xor(send, 4);
    d += rowXor2(d);
    d += rowXor1(d);
    return d;
}

// NC = number of variable columns actually carried (14 for the shipped K = 7, else the whole tile)
template <int NC>
struct Rows
{
    double G0[NC], G1[NC];
    bool act0, act1, head;
    int hd, lane;
};

// G' (t0, t1) -> sh.vec
template <int NC>
__device__ inline void mulGT(const Rows<NC> &R, double t0, double t1, Shared &sh)
{
    double p[NV] = {};
#pragma unroll
    for (int j = 0; j < NC; j++)
        p[j] = R.G0[j] * t0 + R.G1[j] * t1;
    const double d = waveReduce16(p, R.lane);
    WAVE_SYNC();
    if ((R.lane & 3) == 0)
        sh.vec[R.lane >> 2] = d;
    WAVE_SYNC();
}
// G sh.vec -> (o0, o1)
template <int NC>
__device__ inline void mulG(const Rows<NC> &R, const Shared &sh, double &o0, double &o1)
{
    double a0 = 0., a1 = 0.;
#pragma unroll
    for (int j = 0; j < NC; j++)
    {
        const double x = sh.vec[j];
        a0 += R.G0[j] * x;
        a1 += R.G1[j] * x;
    }
    o0 = a0;
    o1 = a1;
}

// sh.vec <- H^-1 sh.vec with H^-1 = Li' Li
__device__ inline void tileSolve(const Tile &Li, const Tile &LiT, Shared &sh, int lane)
{
    const int g = lane >> 4;
    Tile Bt;
#pragma unroll
    for (int r = 0; r < 4; r++)
        Bt.v[r] = sh.vec[g + 4 * r];
    const Tile a = ipm::mm(LiT, Bt);
    const Tile x = ipm::mm(Li, a);
    WAVE_SYNC();
    if ((lane & 15) == 0)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
            sh.vec[g + 4 * r] = x.v[r];
    }
    WAVE_SYNC();
}

// Nesterov-Todd scaling of the slot-1 cones, one component per lane.  Lanes that hold no row are their own "head" with
// zero data: they contribute 0 to every cone sum and every helper returns 0 for them.
// Reciprocals are SHARED: one correctly rounded division per cone / row and iteration with multiplications at the uses (measured on
// 256 states against the twin: identical statuses and iteration counts, 6.5 M -> 8.0 M solves/s).  Two variants were measured
// and dropped: the same treatment of the NT scaling itself (8 of 256 iteration counts then differ from the twin) and hardware
// seed + Newton reciprocals instead of IEEE divisions (+1 %).

struct ConeScal
{
    double w, w0, eta;
    double ieta, iw01; // 1/eta, 1/(1 + w0): shared reciprocals instead of one IEEE division per use
};
// scaled variable lambda = W z of my cone: what every step-length computation of an iteration shares
struct LamInfo
{
    double lam, lam0, iln, f_den, ln2; // component, head, 1/||lambda||_J, 1/(lam0/||lambda||_J + 1), ||lambda||_J^2
};
// value of my cone's head lane (pairs: the even lane; error cone: lane 16; input cone: lane 32)
__device__ inline double headv(double v, int lane)
{
    const double pr = pairHead(v), e = readLane(v, ERR_LANE), i = readLane(v, INP_LANE);
    return lane < 16 ? pr : lane < 32 ? e : lane < 48 ? i : v;
}

template <class RowsT>
__device__ inline double applyW(const ConeScal &c, const RowsT &R, double v)
{
    const double zeta = csum(R.head ? 0. : c.w * v, R.lane), v0 = headv(v, R.lane);
    const double f = v0 + zeta * c.iw01;
    return !R.act1 ? 0. : R.head ? c.eta * (c.w0 * v0 + zeta) : c.eta * (v + f * c.w);
}
template <class RowsT>
__device__ inline double applyWinv(const ConeScal &c, const RowsT &R, double v)
{
    const double zeta = csum(R.head ? 0. : c.w * v, R.lane), v0 = headv(v, R.lane);
    const double f = -v0 + zeta * c.iw01;
    return !R.act1 ? 0. : R.head ? (c.w0 * v0 - zeta) * c.ieta : (v + f * c.w) * c.ieta;
}
template <class RowsT>
__device__ inline double applyWinv2(const ConeScal &c, const RowsT &R, double v)
{
    const double tv = csum(R.head ? c.w * v : -c.w * v, R.lane), v0 = headv(v, R.lane);
    const double e2 = c.ieta * c.ieta;
    return !R.act1 ? 0. : R.head ? e2 * (2. * c.w0 * tv - v0) : e2 * (-2. * c.w * tv + v);
}
template <class RowsT>
__device__ inline double conicProduct(const RowsT &R, double u, double v)
{
    const double s0 = csum(u * v, R.lane), u0 = headv(u, R.lane), v0 = headv(v, R.lane);
    return !R.act1 ? 0. : R.head ? s0 : u0 * v + v0 * u;
}
template <class RowsT>
__device__ inline LamInfo lamInfo(const RowsT &R, double lam)
{
    LamInfo L;
    L.lam = lam;
    L.lam0 = headv(lam, R.lane);
    const double l1 = csum(R.head ? 0. : lam * lam, R.lane);
    const double ln2 = L.lam0 * L.lam0 - l1;
    L.ln2 = ln2;
    L.iln = 1. / sqrt(ln2);
    L.f_den = 1. / (L.lam0 * L.iln + 1.);
    return L;
}
// lam \ dd
template <class RowsT>
__device__ inline double conicDivision(const RowsT &R, const LamInfo &L, double dd)
{
    const double l1d1 = csum(R.head ? 0. : L.lam * dd, R.lane), dd0 = headv(dd, R.lane);
    // rho = lam0^2 - |lam1|^2 = 1 / iln^2
    const double u0 = (L.lam0 * dd0 - l1d1) * (L.iln * L.iln);
    return !R.act1 ? 0. : R.head ? u0 : (dd - u0 * L.lam) * (1. / L.lam0);
}
// 1 / (largest step keeping lam + alpha v in the cone)   (ECOS lineSearch)
template <class RowsT>
__device__ inline double stepInv(const RowsT &R, const LamInfo &L, double v)
{
    const double v0 = headv(v, R.lane);
    const double lbJv = csum(R.head ? L.lam * v : -L.lam * v, R.lane) * L.iln;
    const double rho0 = lbJv * L.iln;
    const double f = (lbJv + v0) * L.f_den;
    const double ri = R.head ? 0. : (v - f * L.lam * L.iln) * L.iln;
    const double r1 = csum(ri * ri, R.lane);
    return R.act1 ? sqrt(r1) - rho0 : 0.;
}

// ECOS bring2cone on (v0 | v1)
template <class RowsT>
__device__ inline void bring2cone(const RowsT &R, double &v0, double &v1)
{
    const double n2 = csum(R.head ? 0. : v1 * v1, R.lane);
    const double cand1 = sqrt(n2) - headv(v1, R.lane);
    double a = -0.99;
    a = R.act0 ? fmax(a, -v0) : a;
    a = R.act1 ? fmax(a, cand1) : a;
    a = waveMaxDpp(a);
    const double sh = 1. + a;
    v0 = R.act0 ? v0 + sh : 0.;
    v1 = (R.act1 && R.head) ? v1 + sh : v1;
}

// status: 0 optimal, 1 reduced accuracy (ECOS "close to optimal"), -1 iteration limit, -2 numerics,
//         -3 the given state violates its own (stage-0) constraints
#ifndef MPC_WAVES_PER_SIMD
#define MPC_WAVES_PER_SIMD 2 // measured (B = 32768): 1 -> 3.3 M solves/s, 2 -> 5.3 M, 3 (83 VGPRs spilled) -> 3.9 M, 4 -> 2.5 M
#endif
template <int NC>
__global__ void __launch_bounds__(64, MPC_WAVES_PER_SIMD) mpc_solve_kernel(const MpcConst *__restrict__ Cg, const double *__restrict__ x0g,
                                                        const double *__restrict__ xfg, double *__restrict__ Uout,
                                                        double *__restrict__ Xout, double *__restrict__ cost,
                                                        int *__restrict__ status, int *__restrict__ iters,
                                                        const int *__restrict__ active, int B)
{
    __shared__ Shared sh;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B)
        return;
    if (active && !active[b])
        return;
    const MpcConst &C = *Cg;
    const int N = C.N, nv = C.nv, nlp = C.nlp, K = C.K;
    double x0[NX], xf[NX];
#pragma unroll
    for (int i = 0; i < NX; i++)
    {
        x0[i] = x0g[size_t(b) * NX + i];
        xf[i] = xfg[size_t(b) * NX + i];
    }
    if (!(fabs(x0[0]) <= C.tan_gs * x0[1]) || !(fabs(x0[4]) <= C.theta_max) || !(fabs(x0[5]) <= C.w_max))
    {
        if (lane == 0)
        {
            status[b] = -3;
            iters[b] = 0;
        }
        return;
    }
    Rows<NC> R;
    R.lane = lane;
    R.act0 = lane < nlp;
    const bool glide = lane < 2 * N, err = lane >= ERR_LANE && lane < ERR_LANE + 1 + NX, inp = lane >= INP_LANE && lane < INP_LANE + 1 + NU * N;
    R.act1 = glide || err || inp;
    R.hd = glide ? (lane & ~1) : err ? ERR_LANE : inp ? INP_LANE : lane;
    R.head = lane == R.hd;
    double h0 = C.c0[lane], h1 = C.c0[64 + lane];
#pragma unroll
    for (int j = 0; j < NC; j++)
    {
        R.G0[j] = C.G[lane][j];
        R.G1[j] = C.G[64 + lane][j];
    }
#pragma unroll
    for (int i = 0; i < NX; i++)
    {
        h0 += C.P[lane][i] * x0[i] + C.Q[lane][i] * xf[i];
        h1 += C.P[64 + lane][i] * x0[i] + C.Q[64 + lane][i] * xf[i];
    }
    const int g = lane >> 4, li = lane & 15;
    // vectors of variables live one entry per lane in lanes 0..15 (0 elsewhere) and are broadcast through sh.vec
    const bool vl = lane < NV;
    const double cl = vl ? C.c[li] : 0.;
    // ---- initial point (ECOS init with W = I): x = argmin |Gx - h|, s = bring2cone(h - Gx); z = bring2cone(G x'), G'G x' = -c
    double s0, s1, z0, z1, xl;
    {
        Tile Li = ipm::loadTile(C.Li0, lane), LiT = ipm::loadTile(C.Li0T, lane);
        mulGT(R, h0, h1, sh);
        tileSolve(Li, LiT, sh, lane);
        double g0, g1;
        mulG(R, sh, g0, g1);
        s0 = h0 - g0;
        s1 = h1 - g1;
        bring2cone(R, s0, s1);
        xl = vl ? sh.vec[li] : 0.;
        WAVE_SYNC();
        if (vl)
            sh.vec[lane] = -cl;
        WAVE_SYNC();
        tileSolve(Li, LiT, sh, lane);
        mulG(R, sh, z0, z1);
        bring2cone(R, z0, z1);
    }
    double resz0, resx0;
    {
        double p[NV] = {};
        p[0] = h0 * h0 + h1 * h1;
        p[1] = cl * cl;
        const double d = waveReduce16(p, lane);
        WAVE_SYNC();
        if ((lane & 3) == 0)
            sh.red[lane >> 2] = d;
        WAVE_SYNC();
        resz0 = fmax(1., sqrt(sh.red[0]));
        resx0 = fmax(1., sqrt(sh.red[1]));
    }
    const double Ddeg = double(nlp + N + 2);
    bool bk_valid = false;
    double pres_prev = 0.;
    int st = -1, it = 0;
    for (int iter = 0;; iter++)
    {
        it = iter;
        // ---- residuals ----
        WAVE_SYNC();
        if (vl)
            sh.vec[lane] = xl;
        WAVE_SYNC();
        double g0, g1;
        mulG(R, sh, g0, g1);
        const double rz0 = R.act0 ? s0 + g0 - h0 : 0., rz1 = R.act1 ? s1 + g1 - h1 : 0.;
        mulGT(R, z0, z1, sh);
        const double rxl = vl ? sh.vec[li] + cl : 0.;
        double gap, nrz, nzz, nss, nrx, nxx, pcost;
        {
            // seven wave totals in one transposing reduction
            double p[NV] = {};
            p[0] = s0 * z0 + s1 * z1;
            p[1] = rz0 * rz0 + rz1 * rz1;
            p[2] = z0 * z0 + z1 * z1;
            p[3] = s0 * s0 + s1 * s1;
            p[4] = rxl * rxl;
            p[5] = xl * xl;
            p[6] = cl * xl;
            const double d = waveReduce16(p, lane);
            WAVE_SYNC();
            if ((lane & 3) == 0)
                sh.red[lane >> 2] = d;
            WAVE_SYNC();
            gap = sh.red[0];
            nrz = sh.red[1];
            nzz = sh.red[2];
            nss = sh.red[3];
            nrx = sh.red[4];
            nxx = sh.red[5];
            pcost = sh.red[6];
        }
        double pres = sqrt(nrz) / fmax(resz0 + sqrt(nxx) + sqrt(nss), 1.);
        double dres = sqrt(nrx) / fmax(resx0 + sqrt(nzz), 1.);
#ifdef SCPP_HIP_EMU
        // Test support (emulator build only): SCPP_EMU_INJECT_MPC_RES="n:pres:dres:gap" replaces the termination quantities of the n-th evaluation of the
        // process (per lane) -- the broken iterate the magnitude / negative-gap rule below exists for
        {
            static int calls[WAVE];
            static int inj_n = -2;
            static double inj_v[3];
            if (inj_n == -2)
            {
                const char *e = getenv("SCPP_EMU_INJECT_MPC_RES");
                inj_n = -1;
                if (e && sscanf(e, "%d:%lf:%lf:%lf", &inj_n, &inj_v[0], &inj_v[1], &inj_v[2]) != 4)
                    inj_n = -1;
            }
            if (inj_n >= 0 && calls[lane & (WAVE - 1)]++ == inj_n)
            {
                pres = inj_v[0];
                dres = inj_v[1];
                gap = inj_v[2];
            }
        }
#endif
        const double mu = gap / Ddeg;
        const double relgap = gap / fmax(fabs(pcost), 1e-300);
        // Broken iterate (round 6, the rule the big solver got in round 5 -- csrc/ipm_solve.h: IPM_BLOWN, IPM_NEG_GAP): every residual is RELATIVE to the
        // iterate's norm, so a blown-up point shows pres = 0, and any negative gap meets `gap < abstol`; until now that was caught only once a
        // fall-back iterate existed.  The problem is scaled to unit magnitudes at set-up (mpc_setup.h): 1e30 is never a value of a working iterate,
        // and a gap below -1e-6 means the point has left the cone.
        const bool finite = (pres - pres == 0.) && (dres - dres == 0.) && (gap - gap == 0.) && fabs(gap) <= 1e30 && fabs(pcost) <= 1e30 && !(gap < -1e-6);
        if (!finite || (bk_valid && (pres > 500. * pres_prev || gap < 0.)))
        {
            if (!bk_valid)
            {
                st = -2;
                break;
            }
            xl = vl ? sh.bk[li] : 0.;
            st = 1;
            break;
        }
        pres_prev = pres;
        if (pres < C.feastol && dres < C.feastol && (gap < C.abstol || relgap < C.reltol))
        {
            st = 0;
            break;
        }
        const bool inacc_ok = pres < 1e-4 && dres < 1e-4 && (gap < 5e-5 || relgap < 5e-5);
        if (inacc_ok)
        {
            if (vl)
                sh.bk[lane] = xl; // only read back by the same lane
            bk_valid = true;
        }
        if (iter >= C.maxit)
        {
            st = inacc_ok ? 1 : -1;
            break;
        }
        // ---- scalings ----
        bool ok = !R.act0 || (s0 > 0. && z0 > 0.);
        const double is0 = 1. / s0, iz0 = 1. / z0;
        const double zos = R.act0 ? z0 * is0 : 0.; // W^-2 of the LP rows
        ConeScal cs;
        {
            const double sh_ = headv(s1, lane), zh_ = headv(z1, lane);
            const double s2 = csum(R.head ? 0. : s1 * s1, lane), z2 = csum(R.head ? 0. : z1 * z1, lane);
            const double sres = sh_ * sh_ - s2, zres = zh_ * zh_ - z2;
            ok = ok && (!R.act1 || (sres > 0. && zres > 0.));
            const double sn = sqrt(sres), zn = sqrt(zres);
            const double sz = csum(s1 * z1, lane) / (sn * zn);
            const double gamma = sqrt(0.5 * (1. + sz));
            const double a = 0.5 / gamma;
            cs.w = R.act1 ? (R.head ? a * (s1 / sn + z1 / zn) : a * (s1 / sn - z1 / zn)) : 0.;
            cs.eta = R.act1 ? sqrt(sn / zn) : 1.;
            cs.ieta = 1. / cs.eta;
            const double hw = headv(cs.w, lane); // (shuffles are never issued under a lane-dependent condition)
            cs.w0 = R.act1 ? hw : 1.;
            cs.iw01 = 1. / (1. + cs.w0);
        }
        if (anyLane(!ok))
        {
            st = inacc_ok ? 1 : -2;
            break;
        }
        const LamInfo L1 = lamInfo(R, applyW(cs, R, z1));
        // ---- H = Gt' Gt, Gt = W^-1 G: rows staged through LDS into MFMA operand layout, 64 rows at a time ----
        Tile Li, LiT;
        {
            d4_t acc = {0., 0., 0., 0.};
            const double iw = R.act0 ? sqrt(zos) : 0.;
            WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < NC; j++)
                sh.gt[lane * GP + j] = R.G0[j] * iw;
            WAVE_SYNC();
#pragma unroll
            for (int t = 0; t < 4; t++)
            {
                if (16 * t >= nlp)
                    continue;
                double v[4];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    v[r] = sh.gt[(16 * t + g + 4 * r) * GP + li];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[r], v[r], acc, 0, 0, 0);
            }
            double gt1[NC];
#pragma unroll
            for (int j = 0; j < NC; j++)
                gt1[j] = applyWinv(cs, R, R.G1[j]);
            WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < NC; j++)
                sh.gt[lane * GP + j] = gt1[j];
            WAVE_SYNC();
#pragma unroll
            for (int t = 0; t < 3; t++)
            {
                double v[4];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    v[r] = sh.gt[(16 * t + g + 4 * r) * GP + li];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[r], v[r], acc, 0, 0, 0);
            }
            Tile H;
#pragma unroll
            for (int r = 0; r < 4; r++)
                H.v[r] = (li >= nv || g + 4 * r >= nv) ? (g + 4 * r == li ? 1. : 0.) : acc[r]; // identity on the padding
            Li = ipm::invCholFactor<NC>(H, sh.ts, lane);
            LiT = ipm::transposeTile(Li, sh.ts, lane);
        }
        // alpha: step length of the primal variables (x, s), alpha_d: of the multipliers z -- their own since round 6 (MPC_SPLIT_STEPS, the rule of
        // csrc/ipm_solve.h: IPM_SPLIT_STEPS; the centring parameter keeps ECOS's rule on the common affine step).  On the scalar twin, 256 controllers of the
        // shipped MPC.info: 14.99 -> 11.69 interior-point iterations per (cold-started) solve, all solved
        double sigma_c = 0., alpha = 1., alpha_d = 1., ds0 = 0., ds1 = 0., dz0 = 0., dz1 = 0., dsS1 = 0., dzS1 = 0., dxl = 0.;
        bool broke = false;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++)
        {
            const double om = 1. - sigma_c;
            // t = W^-2 (om rz) + [ -z | W^-1( lam \ (sigma mu e - dsS o dzS) - lam ) ]
            double t0, t1;
            {
                const double corr0 = (pass && R.act0) ? (sigma_c * mu - ds0 * dz0) * is0 : 0.;
                t0 = R.act0 ? zos * om * rz0 - z0 + corr0 : 0.;
                const double b2 = applyWinv2(cs, R, om * rz1);
                if (pass == 0)
                    t1 = b2 - z1;
                else
                {
                    double dsv = -conicProduct(R, dsS1, dzS1);
                    dsv += R.head ? sigma_c * mu : 0.;
                    double u = conicDivision(R, L1, dsv);
                    u -= L1.lam;
                    t1 = b2 + applyWinv(cs, R, u);
                }
                t1 = R.act1 ? t1 : 0.;
            }
            // H dx = -om rx - G't
            mulGT(R, t0, t1, sh);
            {
                const double bj = (vl && lane < nv) ? -om * rxl - sh.vec[li] : 0.;
                WAVE_SYNC();
                if (vl)
                    sh.vec[lane] = bj;
                WAVE_SYNC();
            }
            tileSolve(Li, LiT, sh, lane);
            dxl = vl ? sh.vec[li] : 0.;
            if (anyLane(!(dxl - dxl == 0.)))
            {
                broke = true;
                break;
            }
            double gd0, gd1;
            mulG(R, sh, gd0, gd1);
            // dz = W^-2 G dx + t ; ds = -om rz - G dx
            dz0 = R.act0 ? zos * gd0 + t0 : 0.;
            ds0 = R.act0 ? -om * rz0 - gd0 : 0.;
            double ainv = R.act0 ? -ds0 * is0 : 0., ainv_d = R.act0 ? -dz0 * iz0 : 0.; // 1 / alpha_max of the slacks' / the multipliers' direction
            ainv = fmax(ainv, 0.);
            ainv_d = fmax(ainv_d, 0.);
            const double w2g = applyWinv2(cs, R, R.act1 ? gd1 : 0.);
            dz1 = R.act1 ? w2g + t1 : 0.;
            ds1 = R.act1 ? -om * rz1 - gd1 : 0.;
            dsS1 = applyWinv(cs, R, ds1);
            dzS1 = applyW(cs, R, dz1);
            const double si = stepInv(R, L1, dsS1), zi = stepInv(R, L1, dzS1);
            ainv = R.act1 ? fmax(ainv, si) : ainv;
            ainv_d = R.act1 ? fmax(ainv_d, zi) : ainv_d;
            if (!MPC_SPLIT_STEPS || pass == 0)
            {
                ainv = waveMaxDpp(fmax(ainv, ainv_d));
                ainv_d = ainv;
            }
            else
            {
                ainv = waveMaxDpp(ainv);
                ainv_d = waveMaxDpp(ainv_d);
            }
            if (pass == 0)
            {
                const double alpha_a = ainv > 0. ? fmin(1. / ainv, 1.) : 1.;
                sigma_c = (1. - alpha_a) * (1. - alpha_a) * (1. - alpha_a);
                sigma_c = fmin(1., fmax(1e-4, sigma_c));
            }
            else
            {
                alpha = ainv > 0. ? fmin(0.99 / ainv, 1.) : 1.;
                alpha = fmin(alpha, 0.999);
                alpha = fmax(alpha, 1e-8);
                alpha_d = ainv_d > 0. ? fmin(0.99 / ainv_d, 1.) : 1.;
                alpha_d = fmax(fmin(alpha_d, 0.999), 1e-8);
            }
        }
        if (broke)
        {
            st = inacc_ok ? 1 : -2;
            break;
        }
        xl += alpha * dxl;
        s0 += alpha * ds0;
        z0 += alpha_d * dz0;
        s1 += alpha * ds1;
        z1 += alpha_d * dz1;
    }
    // ---- results ----
    if (lane == 0)
    {
        status[b] = st;
        iters[b] = it;
    }
    if (st < 0)
        return;
    WAVE_SYNC();
    if (vl)
        sh.vec[lane] = C.D[li] * xl;
    WAVE_SYNC();
    if (lane < NU * N)
        Uout[size_t(b) * NMAX * NU + lane] = sh.vec[lane];
    if (lane < 2)
        cost[size_t(b) * 2 + lane] = sh.vec[NU * N + lane];
    if (lane < NX * K)
    {
        const int k = lane / NX, i = lane % NX;
        double a = C.zeta[k][i];
        for (int j = 0; j < NX; j++)
            a += C.Phi[k][i][j] * x0[j];
        for (int j = 0; j < N; j++)
            for (int c = 0; c < NU; c++)
                a += C.Gam[k][j][i][c] * sh.vec[j * NU + c];
        Xout[size_t(b) * KMAX * NX + lane] = a;
    }
}

// Closed-loop bookkeeping of MPC_sim.cpp:49-86 for B loops (one lane per loop): after the plant step x <- simulate(x, u),
// take the new input from the solve made at the pre-step state (a failed solve holds the previous input), advance the
// clock, count, and retire the loop when it reaches x_final (|x - x_final| < stop_tol) or the simulated time is up.
__global__ void mpc_sim_advance_kernel(int B, const double *__restrict__ Usol, const int *__restrict__ status,
                                       const int *__restrict__ iters, const double *__restrict__ x, const double *__restrict__ xf,
                                       double *__restrict__ u_held, int *__restrict__ active,
                                       int *__restrict__ steps, int *__restrict__ failed, int *__restrict__ ipm_total,
                                       int *__restrict__ reached, double *__restrict__ t, double dt, double sim_time,
                                       double stop_tol, int *__restrict__ n_active)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || !active[b])
        return;
    if (status[b] >= 0)
    {
        u_held[b * NU + 0] = Usol[size_t(b) * NMAX * NU + 0];
        u_held[b * NU + 1] = Usol[size_t(b) * NMAX * NU + 1];
    }
    else
        failed[b]++;
    ipm_total[b] += iters[b];
    steps[b]++;
    t[b] += dt;
    double d2 = 0.;
    for (int i = 0; i < NX; i++)
    {
        const double d = x[size_t(b) * NX + i] - xf[size_t(b) * NX + i];
        d2 += d * d;
    }
    const bool hit = sqrt(d2) < stop_tol;
    if (hit)
        reached[b] = 1;
    if (hit || !(t[b] < sim_time))
        active[b] = 0;
    else
        atomicAdd(n_active, 1);
}

} // namespace mpc
} // namespace scpp
