// ORACLE (test infrastructure, NOT product code).
//
// CPU restatement of the solver the reference calls at
//   scpp_core/src/SCAlgorithm.cpp:63,78  (cvx::ecos::ECOSSolver::solve)
// The reference solves its sub-problems with ECOS through Epigraph; both are un-vendored
// submodules that are EMPTY in /root/reference (lib/Epigraph, pinned SHA unknown), so this
// file restates ECOS's *published* algorithm (Domahidi, Chu, Boyd, "ECOS: An SOCP solver for
// embedded systems", ECC 2013; same scheme as CVXOPT conelp):
//   - standard form  min c'x  s.t. Ax=b, Gx+s=h, s in K = R+^l x Q^{q1} x ... ;
//   - homogeneous self-dual embedding (tau, kappa), Nesterov-Todd scaling,
//     Mehrotra predictor-corrector, step factor 0.99, sigma=(1-alpha_aff)^3;
//   - ONE sparse LDL' of the regularised KKT matrix per iteration (ECOS: static reg 7e-8; here 1e-9 plus a sign-aware
//     dynamic regularisation),
//     iterative refinement against the un-regularised system;
//   - termination feastol=abstol=reltol=1e-8 (ECOS upstream defaults).
// Differences from ECOS that do not change the optimum: no Ruiz equilibration, dense W^2
// blocks instead of ECOS's sparse cone expansion, elimination order supplied by the
// problem builder (stage-interleaved) instead of AMD.
// PARITY UNPINNED at this boundary: no ECOS binary/golden vector exists to pin against.
// Parity status: UNPINNED -- no ECOS binary, source or reference output exists in this environment; pinned only against
// problems with known optima (tests/test_oracle_socp.py) and against the independent structured solver.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <utility>
#include <vector>

namespace oracle
{

// Affine expression sum_i coef_i x_i + c
struct Aff
{
    std::vector<std::pair<int, double>> t;
    double c = 0.;
    Aff() {}
    explicit Aff(double c_) : c(c_) {}
    Aff &add(int idx, double coef)
    {
        t.push_back({idx, coef});
        return *this;
    }
    Aff &addc(double v)
    {
        c += v;
        return *this;
    }
};

struct SocpRow
{
    std::vector<std::pair<int, double>> t; // coefficients on x
    double rhs;                            // eq: sum t x = rhs ; cone: s = rhs - sum t x
    int key;                               // elimination-order key
};

// Problem container + builder (the oracle's stand-in for Epigraph's OptimizationProblem;
// call sites restated: SCProblem.cpp:18-134, rocketQuat.cpp:74-142, rocket2d.cpp:57-83).
struct Socp
{
    int n = 0;
    std::vector<double> c;
    std::vector<int> var_key;
    std::vector<SocpRow> eq;
    std::vector<SocpRow> lp;
    std::vector<std::vector<SocpRow>> soc;

    int addVars(int count, int key)
    {
        const int first = n;
        n += count;
        c.resize(n, 0.);
        var_key.resize(n, key);
        return first;
    }
    void setKey(int var, int key) { var_key[var] = key; }
    // e == 0
    void addEq(const Aff &e, int key)
    {
        SocpRow r;
        r.t = e.t;
        r.rhs = -e.c;
        r.key = key;
        eq.push_back(r);
    }
    // e >= 0  -> s = e(x) = h - Gx  => G = -coef, h = const
    static SocpRow coneRow(const Aff &e, int key)
    {
        SocpRow r;
        r.t = e.t;
        for (auto &p : r.t)
            p.second = -p.second;
        r.rhs = e.c;
        r.key = key;
        return r;
    }
    void addGe0(const Aff &e, int key) { lp.push_back(coneRow(e, key)); }
    // e[0] >= || e[1:] ||_2
    void addSoc(const std::vector<Aff> &e, int key)
    {
        std::vector<SocpRow> rows;
        for (auto &a : e)
            rows.push_back(coneRow(a, key));
        soc.push_back(rows);
    }
    int numEq() const { return int(eq.size()); }
    int numLp() const { return int(lp.size()); }
    int numConeRows() const
    {
        int m = int(lp.size());
        for (auto &c_ : soc)
            m += int(c_.size());
        return m;
    }
};

struct SocpSettings
{
    double feastol = 1e-8;
    double abstol = 1e-8;
    double reltol = 1e-8;
    int maxit = 100;
    double gamma = 0.99;
    double delta_x = 1e-9;      // (1,1) block shift
    double delta_eq = 1e-9;  // (2,2) block shift
    double delta_cone = 1e-9; // (3,3) block shift: only guards cones pinned at their apex

    int nitref = 9;
    bool verbose = false;
};

struct SocpResult
{
    std::vector<double> x, y, z, s;
    int exitflag = -1; // 0 optimal, 10 optimal to reduced accuracy (ECOS_OPTIMAL + ECOS_INACC_OFFSET), 1 primal infeasible, 2 dual infeasible, -1 maxit, -2 numerics
    int iter = 0;
    double pcost = 0, dcost = 0, pres = 0, dres = 0, gap = 0, relgap = 0, mu = 0;
};

namespace detail
{

// Up-looking sparse LDL' (T. Davis' "LDL" algorithm, restated): input upper triangle in CSC.
struct SparseLDL
{
    int n = 0;
    std::vector<int> Lp, Li, Parent, Lnz, Flag, Pattern;
    std::vector<double> Lx, D, Y;

    void symbolic(int n_, const std::vector<int> &Ap, const std::vector<int> &Ai)
    {
        n = n_;
        Parent.assign(n, -1);
        Lnz.assign(n, 0);
        Flag.assign(n, -1);
        for (int k = 0; k < n; k++)
        {
            Flag[k] = k;
            for (int p = Ap[k]; p < Ap[k + 1]; p++)
            {
                int i = Ai[p];
                if (i < k)
                {
                    for (; Flag[i] != k; i = Parent[i])
                    {
                        if (Parent[i] == -1)
                            Parent[i] = k;
                        Lnz[i]++;
                        Flag[i] = k;
                    }
                }
            }
        }
        Lp.assign(n + 1, 0);
        for (int k = 0; k < n; k++)
            Lp[k + 1] = Lp[k] + Lnz[k];
        Li.assign(Lp[n], 0);
        Lx.assign(Lp[n], 0.);
        D.assign(n, 0.);
        Y.assign(n, 0.);
        Pattern.assign(n, 0);
    }

    // returns false on a zero pivot
    // Sign[k] = expected sign of pivot k; a pivot with the wrong sign or |D| <= eps is replaced by
    // Sign*delta_dyn (ECOS' dynamic regularisation, EPS=1e-13, DELTA=2e-7 upstream).
    int n_dynreg = 0;
    bool numeric(const std::vector<int> &Ap, const std::vector<int> &Ai, const std::vector<double> &Ax,
                 const std::vector<int> &Sign, double eps, double delta_dyn)
    {
        n_dynreg = 0;
        for (int k = 0; k < n; k++)
        {
            Y[k] = 0.;
            int top = n;
            Flag[k] = k;
            Lnz[k] = 0;
            for (int p = Ap[k]; p < Ap[k + 1]; p++)
            {
                int i = Ai[p];
                if (i <= k)
                {
                    Y[i] += Ax[p];
                    int len = 0;
                    for (; Flag[i] != k; i = Parent[i])
                    {
                        Pattern[len++] = i;
                        Flag[i] = k;
                    }
                    while (len > 0)
                        Pattern[--top] = Pattern[--len];
                }
            }
            D[k] = Y[k];
            Y[k] = 0.;
            for (; top < n; top++)
            {
                const int i = Pattern[top];
                const double yi = Y[i];
                Y[i] = 0.;
                const int p2 = Lp[i] + Lnz[i];
                for (int p = Lp[i]; p < p2; p++)
                    Y[Li[p]] -= Lx[p] * yi;
                const double l_ki = yi / D[i];
                D[k] -= l_ki * yi;
                if (std::getenv("ORACLE_DEBUG_IR") && !std::isfinite(D[k]))
                {
                    std::fprintf(stderr, "  k=%d i=%d yi=%.3e D[i]=%.3e\n", k, i, yi, D[i]);
                }
                Li[p2] = k;
                Lx[p2] = l_ki;
                Lnz[i]++;
            }
            if (!std::isfinite(D[k]))
            {
                if (std::getenv("ORACLE_DEBUG_IR"))
                    std::fprintf(stderr, "non-finite pivot at %d (sign %d)\n", k, Sign[k]);
                return false;
            }
            if (D[k] * Sign[k] <= eps)
            {
                if (std::getenv("ORACLE_DEBUG_IR"))
                    std::fprintf(stderr, "  dynreg k=%d D=%.3e sign %d\n", k, D[k], Sign[k]);
                D[k] = Sign[k] * delta_dyn;
                n_dynreg++;
            }
        }
        return true;
    }

    void solve(std::vector<double> &x) const
    {
        for (int j = 0; j < n; j++)
            for (int p = Lp[j]; p < Lp[j + 1]; p++)
                x[Li[p]] -= Lx[p] * x[j];
        for (int j = 0; j < n; j++)
            x[j] /= D[j];
        for (int j = n - 1; j >= 0; j--)
            for (int p = Lp[j]; p < Lp[j + 1]; p++)
                x[j] -= Lx[p] * x[Li[p]];
    }
};

} // namespace detail

class SocpSolver
{
  public:
    SocpSettings opt;

    explicit SocpSolver(const Socp &prob) { setup(prob); }

    SocpResult solve();

    int n, p, m, l;
    std::vector<int> qdim, qoff; // cone dims and offsets into the m cone rows
    long factor_nnz() const { return long(ldl.Lx.size()); }

  private:
    // static data
    std::vector<double> c, b, h;
    struct Entry
    {
        int row, col;
        double v;
    }; // A: row in [0,p), G: row in [0,m)
    std::vector<Entry> Aent, Gent;
    std::vector<int> perm, iperm; // perm[new] = old, iperm[old] = new ; old index space: [x | y | z]
    int N;
    // KKT in permuted upper CSC
    std::vector<int> Kp, Ki;
    std::vector<double> Kx;
    std::vector<int> posA, posG, posDiag; // CSC slots
    std::vector<std::vector<int>> posW;   // per SOC cone: dim*(dim+1)/2 slots (upper incl diag), LP handled via posDiag
    detail::SparseLDL ldl;
    std::vector<int> psign; // expected pivot signs in permuted order
    double delta_cur = 7e-8, eps_dyn = 1e-13, delta_dyn = 2e-7;
    double reg_scale = 1.;       // escalation factor of every regularisation (main loop safeguard)
    double worst_kkt_err = 0.;   // largest relative residual of the kktSolve calls since it was last reset

    // scaling state
    std::vector<double> wlp;                // LP: w_i = sqrt(s_i/z_i)
    std::vector<double> eta;                // per cone
    std::vector<std::vector<double>> wbar;  // per cone
    std::vector<double> lambda;

    void setup(const Socp &prob);
    void updateScalings(const std::vector<double> &s, const std::vector<double> &z);
    bool factor();
    // solve K [x;y;z] = [rx;ry;rz] (un-regularised K, refinement on the regularised factor)
    void kktSolve(const std::vector<double> &rx, const std::vector<double> &ry, const std::vector<double> &rz,
                  std::vector<double> &dx, std::vector<double> &dy, std::vector<double> &dz);
    void kktMul(const std::vector<double> &x, const std::vector<double> &y, const std::vector<double> &z,
                std::vector<double> &ox, std::vector<double> &oy, std::vector<double> &oz) const;
    // cone helpers
    void applyW(const std::vector<double> &v, std::vector<double> &out) const;    // W v
    void applyWinv(const std::vector<double> &v, std::vector<double> &out) const; // W^-1 v
    void applyW2(const std::vector<double> &v, std::vector<double> &out) const;   // W^2 v
    void conicProduct(const std::vector<double> &u, const std::vector<double> &v, std::vector<double> &out) const;
    void conicDivision(const std::vector<double> &lam, const std::vector<double> &d, std::vector<double> &out) const;
    void bring2cone(const std::vector<double> &r, std::vector<double> &s) const;
    double maxStep(const std::vector<double> &ds_scaled, const std::vector<double> &dz_scaled) const;
    void mulA(const std::vector<double> &x, std::vector<double> &out) const;
    void mulAT(const std::vector<double> &y, std::vector<double> &out) const;
    void mulG(const std::vector<double> &x, std::vector<double> &out) const;
    void mulGT(const std::vector<double> &z, std::vector<double> &out) const;
};

inline double dot(const std::vector<double> &a, const std::vector<double> &b)
{
    double r = 0.;
    for (size_t i = 0; i < a.size(); i++)
        r += a[i] * b[i];
    return r;
}
inline double norm2(const std::vector<double> &a) { return std::sqrt(dot(a, a)); }

inline void SocpSolver::setup(const Socp &prob)
{
    n = prob.n;
    p = prob.numEq();
    l = prob.numLp();
    m = prob.numConeRows();
    c = prob.c;
    b.resize(p);
    h.resize(m);
    std::vector<int> key(n + p + m);
    for (int j = 0; j < n; j++)
        key[j] = prob.var_key[j];
    for (int r = 0; r < p; r++)
    {
        b[r] = prob.eq[r].rhs;
        key[n + r] = prob.eq[r].key;
        for (auto &t : prob.eq[r].t)
            Aent.push_back({r, t.first, t.second});
    }
    int row = 0;
    for (int r = 0; r < l; r++, row++)
    {
        h[row] = prob.lp[r].rhs;
        key[n + p + row] = prob.lp[r].key;
        for (auto &t : prob.lp[r].t)
            Gent.push_back({row, t.first, t.second});
    }
    for (auto &cone : prob.soc)
    {
        qoff.push_back(row);
        qdim.push_back(int(cone.size()));
        for (auto &r : cone)
        {
            h[row] = r.rhs;
            key[n + p + row] = r.key;
            for (auto &t : r.t)
                Gent.push_back({row, t.first, t.second});
            row++;
        }
    }
    assert(row == m);
    N = n + p + m;
    perm.resize(N);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b_) { return key[a] < key[b_]; });
    iperm.resize(N);
    for (int i = 0; i < N; i++)
        iperm[perm[i]] = i;

    // assemble pattern: triplets (r,c) in permuted upper form
    struct Trip
    {
        int r, c, id;
    };
    std::vector<Trip> trips;
    int id = 0;
    auto push = [&](int oldr, int oldc) {
        int a = iperm[oldr], b_ = iperm[oldc];
        if (a > b_)
            std::swap(a, b_);
        trips.push_back({a, b_, id});
        return id++;
    };
    std::vector<int> idDiag(N), idA(Aent.size()), idG(Gent.size());
    for (int i = 0; i < N; i++)
        idDiag[i] = push(i, i);
    for (size_t e = 0; e < Aent.size(); e++)
        idA[e] = push(n + Aent[e].row, Aent[e].col);
    for (size_t e = 0; e < Gent.size(); e++)
        idG[e] = push(n + p + Gent[e].row, Gent[e].col);
    std::vector<std::vector<int>> idW(qdim.size());
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int d = qdim[k];
        for (int i = 0; i < d; i++)
            for (int j = i + 1; j < d; j++)
                idW[k].push_back(push(n + p + qoff[k] + i, n + p + qoff[k] + j));
    }
    // sort by (col,row); duplicates (same slot) are merged by accumulating values
    std::vector<int> order(trips.size());
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b_) {
        if (trips[a].c != trips[b_].c)
            return trips[a].c < trips[b_].c;
        return trips[a].r < trips[b_].r;
    });
    Kp.assign(N + 1, 0);
    std::vector<int> slotOf(trips.size());
    int lastr = -1, lastc = -1;
    for (int o : order)
    {
        const Trip &t = trips[o];
        if (t.r != lastr || t.c != lastc)
        {
            Ki.push_back(t.r);
            Kp[t.c + 1]++;
            lastr = t.r;
            lastc = t.c;
        }
        slotOf[t.id] = int(Ki.size()) - 1;
    }
    for (int i = 0; i < N; i++)
        Kp[i + 1] += Kp[i];
    Kx.assign(Ki.size(), 0.);
    posDiag.resize(N);
    for (int i = 0; i < N; i++)
        posDiag[i] = slotOf[idDiag[i]];
    posA.resize(Aent.size());
    for (size_t e = 0; e < Aent.size(); e++)
        posA[e] = slotOf[idA[e]];
    posG.resize(Gent.size());
    for (size_t e = 0; e < Gent.size(); e++)
        posG[e] = slotOf[idG[e]];
    posW.resize(qdim.size());
    for (size_t k = 0; k < qdim.size(); k++)
    {
        posW[k].resize(idW[k].size());
        for (size_t e = 0; e < idW[k].size(); e++)
            posW[k][e] = slotOf[idW[k][e]];
    }
    ldl.symbolic(N, Kp, Ki);
    psign.resize(N);
    for (int i = 0; i < N; i++)
        psign[i] = perm[i] < n ? 1 : -1;
    if (std::getenv("ORACLE_DEBUG_IR"))
        for (int i = 0; i < N; i++)
            std::fprintf(stderr, "perm %d -> old %d key %d\n", i, perm[i], key[perm[i]]);

    wlp.assign(l, 1.);
    eta.assign(qdim.size(), 1.);
    wbar.resize(qdim.size());
    for (size_t k = 0; k < qdim.size(); k++)
    {
        wbar[k].assign(qdim[k], 0.);
        wbar[k][0] = 1.;
    }
    lambda.assign(m, 0.);
}

inline void SocpSolver::mulA(const std::vector<double> &x, std::vector<double> &out) const
{
    out.assign(p, 0.);
    for (auto &e : Aent)
        out[e.row] += e.v * x[e.col];
}
inline void SocpSolver::mulAT(const std::vector<double> &y, std::vector<double> &out) const
{
    out.assign(n, 0.);
    for (auto &e : Aent)
        out[e.col] += e.v * y[e.row];
}
inline void SocpSolver::mulG(const std::vector<double> &x, std::vector<double> &out) const
{
    out.assign(m, 0.);
    for (auto &e : Gent)
        out[e.row] += e.v * x[e.col];
}
inline void SocpSolver::mulGT(const std::vector<double> &z, std::vector<double> &out) const
{
    out.assign(n, 0.);
    for (auto &e : Gent)
        out[e.col] += e.v * z[e.row];
}

// W = eta * [ w0, w1' ; w1, I + w1 w1'/(1+w0) ]
inline void SocpSolver::applyW(const std::vector<double> &v, std::vector<double> &out) const
{
    out.resize(m);
    for (int i = 0; i < l; i++)
        out[i] = wlp[i] * v[i];
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        const std::vector<double> &w = wbar[k];
        double zeta = 0.;
        for (int i = 1; i < d; i++)
            zeta += w[i] * v[o + i];
        out[o] = eta[k] * (w[0] * v[o] + zeta);
        const double f = v[o] + zeta / (1. + w[0]);
        for (int i = 1; i < d; i++)
            out[o + i] = eta[k] * (v[o + i] + f * w[i]);
    }
}
inline void SocpSolver::applyWinv(const std::vector<double> &v, std::vector<double> &out) const
{
    out.resize(m);
    for (int i = 0; i < l; i++)
        out[i] = v[i] / wlp[i];
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        const std::vector<double> &w = wbar[k];
        double zeta = 0.;
        for (int i = 1; i < d; i++)
            zeta += w[i] * v[o + i];
        out[o] = (w[0] * v[o] - zeta) / eta[k];
        const double f = -v[o] + zeta / (1. + w[0]);
        for (int i = 1; i < d; i++)
            out[o + i] = (v[o + i] + f * w[i]) / eta[k];
    }
}
// W^2 = eta^2 (2 w w' - J)
inline void SocpSolver::applyW2(const std::vector<double> &v, std::vector<double> &out) const
{
    out.resize(m);
    for (int i = 0; i < l; i++)
        out[i] = wlp[i] * wlp[i] * v[i];
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        const std::vector<double> &w = wbar[k];
        double wv = 0.;
        for (int i = 0; i < d; i++)
            wv += w[i] * v[o + i];
        const double e2 = eta[k] * eta[k];
        out[o] = e2 * (2. * w[0] * wv - v[o]);
        for (int i = 1; i < d; i++)
            out[o + i] = e2 * (2. * w[i] * wv + v[o + i]);
    }
}

inline void SocpSolver::conicProduct(const std::vector<double> &u, const std::vector<double> &v,
                                     std::vector<double> &out) const
{
    out.resize(m);
    for (int i = 0; i < l; i++)
        out[i] = u[i] * v[i];
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        double s0 = 0.;
        for (int i = 0; i < d; i++)
            s0 += u[o + i] * v[o + i];
        out[o] = s0;
        for (int i = 1; i < d; i++)
            out[o + i] = u[o] * v[o + i] + v[o] * u[o + i];
    }
}

// solve lam o out = d
inline void SocpSolver::conicDivision(const std::vector<double> &lam, const std::vector<double> &d,
                                      std::vector<double> &out) const
{
    out.resize(m);
    for (int i = 0; i < l; i++)
        out[i] = d[i] / lam[i];
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], dim = qdim[k];
        double l1d1 = 0., l1l1 = 0.;
        for (int i = 1; i < dim; i++)
        {
            l1d1 += lam[o + i] * d[o + i];
            l1l1 += lam[o + i] * lam[o + i];
        }
        const double rho = lam[o] * lam[o] - l1l1;
        const double u0 = (lam[o] * d[o] - l1d1) / rho;
        out[o] = u0;
        for (int i = 1; i < dim; i++)
            out[o + i] = (d[o + i] - u0 * lam[o + i]) / lam[o];
    }
}

inline void SocpSolver::bring2cone(const std::vector<double> &r, std::vector<double> &s) const
{
    double alpha = -opt.gamma;
    for (int i = 0; i < l; i++)
        if (r[i] <= 0. && -r[i] > alpha)
            alpha = -r[i];
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        double nrm = 0.;
        for (int i = 1; i < d; i++)
            nrm += r[o + i] * r[o + i];
        const double cres = r[o] - std::sqrt(nrm);
        if (cres <= 0. && -cres > alpha)
            alpha = -cres;
    }
    alpha += 1.;
    s = r;
    for (int i = 0; i < l; i++)
        s[i] += alpha;
    for (size_t k = 0; k < qdim.size(); k++)
        s[qoff[k]] += alpha;
}

inline void SocpSolver::updateScalings(const std::vector<double> &s, const std::vector<double> &z)
{
    for (int i = 0; i < l; i++)
    {
        if (!(s[i] > 0.) || !(z[i] > 0.))
            throw std::runtime_error("updateScalings: LP iterate left the cone");
        wlp[i] = std::sqrt(s[i] / z[i]);
    }
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        double s1 = 0., z1 = 0.;
        for (int i = 1; i < d; i++)
        {
            s1 += s[o + i] * s[o + i];
            z1 += z[o + i] * z[o + i];
        }
        const double sres = s[o] * s[o] - s1, zres = z[o] * z[o] - z1;
        if (!(sres > 0.) || !(zres > 0.))
            throw std::runtime_error("updateScalings: iterate left the cone");
        const double snorm = std::sqrt(sres), znorm = std::sqrt(zres);
        double sz = 0.;
        for (int i = 0; i < d; i++)
            sz += (s[o + i] / snorm) * (z[o + i] / znorm);
        const double gamma = std::sqrt(0.5 * (1. + sz));
        const double a = 0.5 / gamma;
        wbar[k][0] = a * (s[o] / snorm + z[o] / znorm);
        for (int i = 1; i < d; i++)
            wbar[k][i] = a * (s[o + i] / snorm - z[o + i] / znorm);
        eta[k] = std::sqrt(snorm / znorm);
        if (std::getenv("ORACLE_DEBUG_IR") && (!std::isfinite(eta[k]) || !std::isfinite(wbar[k][0]) || wbar[k][0] > 1e7))
            std::fprintf(stderr, "cone %zu dim %d: sres %.3e zres %.3e s0 %.3e z0 %.3e w0 %.3e eta %.3e\n", k, d, sres, zres, s[o], z[o], wbar[k][0], eta[k]);
    }
    applyW(z, lambda);
}

inline bool SocpSolver::factor()
{
    // Regularisation: cone rows are eliminated FIRST (pivots -W^2, definite by construction) and the
    // variables next (pivots G'W^-2G > 0), so neither block needs ECOS' static delta; only the equality
    // block (rank-deficient in the reference problems: duplicated rows) is shifted by -delta_eq.
    const double dx_ = opt.delta_x * reg_scale, dy_ = opt.delta_eq * reg_scale, dz_ = opt.delta_cone * reg_scale;
    std::fill(Kx.begin(), Kx.end(), 0.);
    for (int j = 0; j < n; j++)
        Kx[posDiag[j]] += dx_;
    for (int r = 0; r < p; r++)
        Kx[posDiag[n + r]] += -dy_;
    for (size_t e = 0; e < Aent.size(); e++)
        Kx[posA[e]] += Aent[e].v;
    for (size_t e = 0; e < Gent.size(); e++)
        Kx[posG[e]] += Gent[e].v;
    for (int i = 0; i < l; i++)
        Kx[posDiag[n + p + i]] += -wlp[i] * wlp[i] - dz_;
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        const double e2 = eta[k] * eta[k];
        const std::vector<double> &w = wbar[k];
        int e = 0;
        for (int i = 0; i < d; i++)
        {
            const double J = (i == 0) ? 1. : -1.;
            Kx[posDiag[n + p + o + i]] += -e2 * (2. * w[i] * w[i] - J) - dz_;
            for (int j = i + 1; j < d; j++)
                Kx[posW[k][e++]] += -e2 * 2. * w[i] * w[j];
        }
    }
    return ldl.numeric(Kp, Ki, Kx, psign, eps_dyn, delta_dyn);
}

inline void SocpSolver::kktMul(const std::vector<double> &x, const std::vector<double> &y,
                               const std::vector<double> &z, std::vector<double> &ox,
                               std::vector<double> &oy, std::vector<double> &oz) const
{
    std::vector<double> t;
    mulAT(y, ox);
    mulGT(z, t);
    for (int j = 0; j < n; j++)
        ox[j] += t[j];
    mulA(x, oy);
    mulG(x, oz);
    applyW2(z, t);
    for (int i = 0; i < m; i++)
        oz[i] -= t[i];
}

inline void SocpSolver::kktSolve(const std::vector<double> &rx, const std::vector<double> &ry,
                                 const std::vector<double> &rz, std::vector<double> &dx,
                                 std::vector<double> &dy, std::vector<double> &dz)
{
    std::vector<double> rhs(N), sol(N), ex, ey, ez, bx = rx, by = ry, bz = rz;
    dx.assign(n, 0.);
    dy.assign(p, 0.);
    dz.assign(m, 0.);
    double bnorm = 0.;
    for (double v : rx)
        bnorm = std::max(bnorm, std::fabs(v));
    for (double v : ry)
        bnorm = std::max(bnorm, std::fabs(v));
    for (double v : rz)
        bnorm = std::max(bnorm, std::fabs(v));
    double prev_err = 1e300;
    double accepted_err = 1e300;
    for (int it = 0; it <= opt.nitref; it++)
    {
        for (int j = 0; j < n; j++)
            rhs[iperm[j]] = bx[j];
        for (int j = 0; j < p; j++)
            rhs[iperm[n + j]] = by[j];
        for (int j = 0; j < m; j++)
            rhs[iperm[n + p + j]] = bz[j];
        sol = rhs;
        ldl.solve(sol);
        // candidate update
        std::vector<double> nx(dx), ny(dy), nz(dz);
        for (int j = 0; j < n; j++)
            nx[j] += sol[iperm[j]];
        for (int j = 0; j < p; j++)
            ny[j] += sol[iperm[n + j]];
        for (int j = 0; j < m; j++)
            nz[j] += sol[iperm[n + p + j]];
        kktMul(nx, ny, nz, ex, ey, ez);
        double err = 0.;
        for (int j = 0; j < n; j++)
        {
            ex[j] = rx[j] - ex[j];
            err = std::max(err, std::fabs(ex[j]));
        }
        for (int j = 0; j < p; j++)
        {
            ey[j] = ry[j] - ey[j];
            err = std::max(err, std::fabs(ey[j]));
        }
        for (int j = 0; j < m; j++)
        {
            ez[j] = rz[j] - ez[j];
            err = std::max(err, std::fabs(ez[j]));
        }
        if (std::getenv("ORACLE_DEBUG_IR"))
            std::fprintf(stderr, "   ir %d err %.3e bnorm %.3e\n", it, err, bnorm);
        if (it > 0 && !(err < prev_err))
            break; // refinement stopped improving: keep previous iterate
        dx = nx;
        dy = ny;
        dz = nz;
        accepted_err = err;
        if (err < 1e-14 * (1. + bnorm))
            break;
        if (it > 0 && err > prev_err / 6.)
            break; // ECOS IRERRFACT: not enough progress
        prev_err = err;
        bx = ex;
        by = ey;
        bz = ez;
    }
    // relative residual of the accepted solution against the UN-regularised system: what the caller's safeguard looks at
    const double rel = std::isfinite(accepted_err) ? accepted_err / (1. + bnorm) : 1e300;
    worst_kkt_err = std::max(worst_kkt_err, rel);
}

inline double SocpSolver::maxStep(const std::vector<double> &ds, const std::vector<double> &dz) const
{
    // ds, dz are the SCALED directions W^-1 ds and W dz; boundary of lambda + a*d (ECOS lineSearch)
    double amax_inv = 0.; // 1/alpha
    for (int i = 0; i < l; i++)
    {
        amax_inv = std::max(amax_inv, -ds[i] / lambda[i]);
        amax_inv = std::max(amax_inv, -dz[i] / lambda[i]);
    }
    for (size_t k = 0; k < qdim.size(); k++)
    {
        const int o = qoff[k], d = qdim[k];
        double l1 = 0.;
        for (int i = 1; i < d; i++)
            l1 += lambda[o + i] * lambda[o + i];
        const double lnorm = std::sqrt(lambda[o] * lambda[o] - l1);
        std::vector<double> lb(d);
        for (int i = 0; i < d; i++)
            lb[i] = lambda[o + i] / lnorm;
        for (int which = 0; which < 2; which++)
        {
            const std::vector<double> &v = which ? dz : ds;
            double lbJv = lb[0] * v[o];
            for (int i = 1; i < d; i++)
                lbJv -= lb[i] * v[o + i];
            const double rho0 = lbJv / lnorm;
            const double f = (lbJv + v[o]) / (lb[0] + 1.);
            double r1 = 0.;
            for (int i = 1; i < d; i++)
            {
                const double ri = (v[o + i] - f * lb[i]) / lnorm;
                r1 += ri * ri;
            }
            amax_inv = std::max(amax_inv, std::sqrt(r1) - rho0);
        }
    }
    return amax_inv;
}

inline SocpResult SocpSolver::solve()
{
    SocpResult R;
    std::vector<double> x(n, 0.), y(p, 0.), z(m, 0.), s(m, 0.);
    double tau = 1., kap = 1.;
    const int D = l + int(qdim.size());
    const double resx0 = std::max(1., norm2(c)), resy0 = std::max(1., norm2(b)), resz0 = std::max(1., norm2(h));

    std::vector<double> zero_n(n, 0.), zero_p(p, 0.), zero_m(m, 0.);
    std::vector<double> t1, t2, t3;

    // ---- initialisation (ECOS init(): W = I) ----
    if (!factor())
    {
        R.exitflag = -2;
        return R;
    }
    {
        std::vector<double> dx, dy, dz;
        kktSolve(zero_n, b, h, dx, dy, dz); // K [x;y;-r] = [0;b;h]
        x = dx;
        std::vector<double> r(m);
        for (int i = 0; i < m; i++)
            r[i] = -dz[i];
        bring2cone(r, s);
        std::vector<double> mc(n);
        for (int j = 0; j < n; j++)
            mc[j] = -c[j];
        kktSolve(mc, zero_p, zero_m, dx, dy, dz); // K [x;y;z] = [-c;0;0]
        y = dy;
        bring2cone(dz, z);
    }

    std::vector<double> rx(n), ry(p), rz(m), x1, y1, z1, x2, y2, z2;
    std::vector<double> dx(n), dy(p), dz(m), ds(m), dsa_s, dza_s, tmp, tmp2;
    // ECOS keeps the best iterate seen so far and returns it when the path breaks down later ("close to optimal",
    // ECOS_OPTIMAL + ECOS_INACC_OFFSET): here the last iterate that met the reduced tolerances
    struct Best
    {
        std::vector<double> x, y, z, s;
        double tau = 1., kap = 1., pcost = 0., dcost = 0., pres = 0., dres = 0., gap = 0., relgap = 0.;
        bool valid = false;
    } best;
    double pres_prev = 0.;
    auto restoreBest = [&]() {
        x = best.x;
        y = best.y;
        z = best.z;
        s = best.s;
        tau = best.tau;
        kap = best.kap;
        R.pcost = best.pcost;
        R.dcost = best.dcost;
        R.pres = best.pres;
        R.dres = best.dres;
        R.gap = best.gap;
        R.relgap = best.relgap;
    };

    for (int iter = 0;; iter++)
    {
        // residuals
        mulAT(y, t1);
        mulGT(z, t2);
        for (int j = 0; j < n; j++)
            rx[j] = t1[j] + t2[j] + c[j] * tau;
        mulA(x, t1);
        for (int j = 0; j < p; j++)
            ry[j] = -t1[j] + b[j] * tau;
        mulG(x, t1);
        for (int j = 0; j < m; j++)
            rz[j] = -t1[j] + h[j] * tau - s[j];
        const double cx = dot(c, x), by = dot(b, y), hz = dot(h, z);
        const double rt = -cx - by - hz - kap;
        const double gap = dot(s, z);
        const double mu = (gap + kap * tau) / (D + 1);
        const double pcost = cx / tau, dcost = -(by + hz) / tau;
        const double nx = norm2(x), ny = norm2(y), nz = norm2(z), ns = norm2(s);
        const double nry = p > 0 ? norm2(ry) / std::max(resy0 + nx, 1.) : 0.;
        const double nrz = norm2(rz) / std::max(resz0 + nx + ns, 1.);
        const double pres = std::max(nry, nrz) / tau;
        const double dres = norm2(rx) / std::max(resx0 + ny + nz, 1.) / tau;
        double relgap = 1e300;
        if (pcost < 0.)
            relgap = gap / (-pcost);
        else if (dcost > 0.)
            relgap = gap / dcost;
        R.iter = iter;
        R.pcost = pcost;
        R.dcost = dcost;
        R.pres = pres;
        R.dres = dres;
        R.gap = gap;
        R.relgap = relgap;
        R.mu = mu;
        if (opt.verbose)
            std::printf("%3d  pcost %+.6e dcost %+.6e gap %.2e pres %.2e dres %.2e k/t %.2e mu %.2e\n", iter, pcost,
                        dcost, gap, pres, dres, kap / tau, mu);
        if (!std::isfinite(pres) || !std::isfinite(dres) || !std::isfinite(gap) ||
            (best.valid && iter > 0 && (pres > 500. * pres_prev || gap < 0.)))
        {
            // residual explosion after an acceptable iterate: return that iterate
            R.exitflag = best.valid ? 10 : -2;
            if (best.valid)
                restoreBest();
            break;
        }
        pres_prev = pres;
        if (tau > 0. && pres >= 0. && dres >= 0. && (-cx > 0. || -by - hz >= -opt.abstol) && pres < opt.feastol && dres < opt.feastol &&
            (gap < opt.abstol || relgap < opt.reltol))
        {
            R.exitflag = 0;
            break;
        }
        // infeasibility certificates (ECOS checkExitConditions)
        {
            double pinfres = 1e300;
            if (hz + by < 0.)
            {
                std::vector<double> a1, g1v;
                mulAT(y, a1);
                mulGT(z, g1v);
                for (int j = 0; j < n; j++)
                    a1[j] += g1v[j];
                pinfres = norm2(a1) / std::max(ny + nz, 1.);
            }
            if ((hz + by) < -opt.abstol && pinfres / (-(hz + by)) < opt.feastol && kap > tau)
            {
                R.exitflag = 1;
                break;
            }
            if (cx < -opt.abstol && kap > tau)
            {
                std::vector<double> ax, gx;
                mulA(x, ax);
                mulG(x, gx);
                for (int j = 0; j < m; j++)
                    gx[j] += s[j];
                const double dinfres = std::max(p > 0 ? norm2(ax) / std::max(nx, 1.) : 0., norm2(gx) / std::max(nx + ns, 1.));
                if (dinfres / (-cx) < opt.feastol)
                {
                    R.exitflag = 2;
                    break;
                }
            }
        }
        // ECOS's reduced-accuracy exit (feastol_inacc 1e-4, abstol_inacc = reltol_inacc = 5e-5): when the iteration limit or a
        // numerical breakdown is hit at an iterate that already satisfies the relaxed tolerances, ECOS returns it as
        // "close to optimal" (exitflag ECOS_OPTIMAL + ECOS_INACC_OFFSET = 10) instead of failing
        const bool inacc_ok = tau > 0. && (-cx > 0. || -by - hz >= -5e-5) && pres < 1e-4 && dres < 1e-4 && (gap < 5e-5 || relgap < 5e-5);
        if (inacc_ok)
        {
            best.x = x;
            best.y = y;
            best.z = z;
            best.s = s;
            best.tau = tau;
            best.kap = kap;
            best.pcost = pcost;
            best.dcost = dcost;
            best.pres = pres;
            best.dres = dres;
            best.gap = gap;
            best.relgap = relgap;
            best.valid = true;
        }
        if (iter >= opt.maxit)
        {
            R.exitflag = inacc_ok ? 10 : -1;
            break;
        }

        try
        {
            updateScalings(s, z);
        }
        catch (const std::exception &e)
        {
            if (opt.verbose)
                std::printf("numerics: %s\n", e.what());
            R.exitflag = inacc_ok ? 10 : -2;
            break;
        }
        // ---- factorisation + the three solves + step length, SAFEGUARDED (round 5: the literal checker failed on ~5 % of the SCvx runs) ----
        // Near the optimum of a degenerate sub-problem the scalings span 1e+-9, a pivot of the quasi-definite factor falls to the
        // dynamic-regularisation floor and the directions come back with entries of 1e50: one such step took tau NEGATIVE, after which every
        // residual quotient was negative and passed the `< tolerance` tests -- "optimal" at a cost of 3.7e13.  Now: the directions must be
        // finite, their KKT residual against the un-regularised system small, and the step must keep tau and kappa positive; otherwise the
        // regularisations are escalated (x 30 per attempt, five attempts) and the iteration is recomputed; only then is it a numerical
        // failure (exit -2, or 10 from a reduced-accuracy iterate).
        double alpha = 0., dtau = 0., dkap = 0.;
        bool step_ok = false;
        reg_scale = 1.; // (every iteration starts from the nominal regularisation: an escalation is a repair of ONE factorisation)
        for (int attempt = 0; attempt < 6 && !step_ok; attempt++)
        {
        if (attempt > 0)
            reg_scale *= 30.;
        worst_kkt_err = 0.;
        delta_dyn = 2 * 1e-9 * reg_scale;
        eps_dyn = 1e-3 * 1e-9 * reg_scale;
        if (!factor())
        {
            if (opt.verbose)
                std::printf("numerics: factorisation failed (regularisation x %.0e)\n", reg_scale);
            continue;
        }

        // v1 = K^-1 [-c; b; h]
        {
            std::vector<double> mc(n);
            for (int j = 0; j < n; j++)
                mc[j] = -c[j];
            kktSolve(mc, b, h, x1, y1, z1);
        }
        const double g1 = dot(c, x1) + dot(b, y1) + dot(h, z1);
        const double denom = kap - tau * g1;

        // ---- affine direction: ds = -lambda o lambda  ->  -W(lambda\ds) = +s ----
        {
            std::vector<double> bx(n), bz(m);
            for (int j = 0; j < n; j++)
                bx[j] = -rx[j];
            for (int j = 0; j < m; j++)
                bz[j] = rz[j] + s[j];
            kktSolve(bx, ry, bz, x2, y2, z2);
        }
        double g2 = dot(c, x2) + dot(b, y2) + dot(h, z2);
        double dtau_a = (-tau * kap - tau * rt + tau * g2) / denom;
        for (int j = 0; j < m; j++)
            dz[j] = z2[j] + dtau_a * z1[j];
        for (int j = 0; j < n; j++)
            dx[j] = x2[j] + dtau_a * x1[j];
        for (int j = 0; j < p; j++)
            dy[j] = y2[j] + dtau_a * y1[j];
        double dkap_a = -(dot(c, dx) + dot(b, dy) + dot(h, dz)) + rt;
        // scaled directions: W dz  and  W^-1 ds = -lambda - W dz
        applyW(dz, dza_s);
        dsa_s.resize(m);
        for (int j = 0; j < m; j++)
            dsa_s[j] = -lambda[j] - dza_s[j];
        double ainv = maxStep(dsa_s, dza_s);
        if (dtau_a < 0.)
            ainv = std::max(ainv, -dtau_a / tau);
        if (dkap_a < 0.)
            ainv = std::max(ainv, -dkap_a / kap);
        double alpha_a = ainv > 0. ? std::min(1. / ainv, 1.) : 1.;
        double sigma = (1. - alpha_a) * (1. - alpha_a) * (1. - alpha_a);
        sigma = std::min(1., std::max(1e-4, sigma));

        // ---- combined direction ----
        conicProduct(dsa_s, dza_s, tmp); // (W^-1 ds_a) o (W dz_a)
        conicProduct(lambda, lambda, tmp2);
        std::vector<double> dsv(m);
        for (int j = 0; j < m; j++)
            dsv[j] = -tmp2[j] - tmp[j];
        for (int i = 0; i < l; i++)
            dsv[i] += sigma * mu;
        for (size_t k = 0; k < qdim.size(); k++)
            dsv[qoff[k]] += sigma * mu;
        std::vector<double> lds, Wlds;
        conicDivision(lambda, dsv, lds);
        applyW(lds, Wlds);
        {
            std::vector<double> bx(n), by_(p), bz(m);
            for (int j = 0; j < n; j++)
                bx[j] = -(1. - sigma) * rx[j];
            for (int j = 0; j < p; j++)
                by_[j] = (1. - sigma) * ry[j];
            for (int j = 0; j < m; j++)
                bz[j] = (1. - sigma) * rz[j] - Wlds[j];
            kktSolve(bx, by_, bz, x2, y2, z2);
        }
        g2 = dot(c, x2) + dot(b, y2) + dot(h, z2);
        const double dkap_rhs = -tau * kap - dtau_a * dkap_a + sigma * mu;
        dtau = (dkap_rhs - tau * (1. - sigma) * rt + tau * g2) / denom;
        for (int j = 0; j < m; j++)
            dz[j] = z2[j] + dtau * z1[j];
        for (int j = 0; j < n; j++)
            dx[j] = x2[j] + dtau * x1[j];
        for (int j = 0; j < p; j++)
            dy[j] = y2[j] + dtau * y1[j];
        dkap = -(dot(c, dx) + dot(b, dy) + dot(h, dz)) + (1. - sigma) * rt;
        // ds = W(lambda\ds) - W^2 dz ; scaled: W^-1 ds = lds - W dz
        applyW(dz, dza_s);
        for (int j = 0; j < m; j++)
            dsa_s[j] = lds[j] - dza_s[j];
        ainv = maxStep(dsa_s, dza_s);
        if (dtau < 0.)
            ainv = std::max(ainv, -dtau / tau);
        if (dkap < 0.)
            ainv = std::max(ainv, -dkap / kap);
        alpha = ainv > 0. ? opt.gamma / ainv : 1.;
        alpha = std::min(alpha, 0.999);
        alpha = std::max(alpha, 1e-6);
        {
            bool finite = std::isfinite(dtau) && std::isfinite(dkap) && std::isfinite(alpha) && std::isfinite(ainv);
            double dmax = 0.;
            for (int j = 0; j < n && finite; j++)
            {
                finite = std::isfinite(dx[j]);
                dmax = std::max(dmax, std::fabs(dx[j]));
            }
            for (int j = 0; j < m && finite; j++)
                finite = std::isfinite(dz[j]) && std::isfinite(dsa_s[j]);
            // (the residual bound is a sanity bound -- a healthy solve sits at 1e-9 .. 1e-6 late in the path, a broken one at 1e0 .. 1e20)
            step_ok = finite && worst_kkt_err < 1e-3 && tau + alpha * dtau > 0. && kap + alpha * dkap > 0. && dmax < 1e30;
            if (!step_ok && opt.verbose)
                std::printf("numerics: step refused (finite %d, kkt residual %.1e, tau' %.2e, kappa' %.2e, |dx| %.1e), regularisation x %.0e\n", int(finite),
                            worst_kkt_err, tau + alpha * dtau, kap + alpha * dkap, dmax, reg_scale);
        }
        } // attempts
        if (!step_ok)
        {
            R.exitflag = inacc_ok ? 10 : -2;
            break;
        }
        applyW(dsa_s, ds);
        for (int j = 0; j < n; j++)
            x[j] += alpha * dx[j];
        for (int j = 0; j < p; j++)
            y[j] += alpha * dy[j];
        for (int j = 0; j < m; j++)
        {
            z[j] += alpha * dz[j];
            s[j] += alpha * ds[j];
        }
        tau += alpha * dtau;
        kap += alpha * dkap;
    }

    R.x.resize(n);
    R.y.resize(p);
    R.z.resize(m);
    R.s.resize(m);
    for (int j = 0; j < n; j++)
        R.x[j] = x[j] / tau;
    for (int j = 0; j < p; j++)
        R.y[j] = y[j] / tau;
    for (int j = 0; j < m; j++)
    {
        R.z[j] = z[j] / tau;
        R.s[j] = s[j] / tau;
    }
    return R;
}

} // namespace oracle
