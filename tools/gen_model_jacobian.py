"""Generates scpp_amd/csrc/model_jacobian_rows.h: analytic rows of [df/dx | df/du] of the model plugins' flow maps.

The reference obtains its Jacobians by taping systemFlowMap with CppAD and letting CppADCodeGen emit optimised (sparse) C
source that is compiled and dlopen'ed at start-up (scpp_core/include/systemDynamics.hpp:109-168).  The build-time analogue
here: the C++ flow maps themselves, EXECUTED with a scalar type that records expressions (tools/flowmap_symbolic.cpp; round 6 --
the model is written once), parsed with sympy, differentiated, simplified by common-subexpression elimination and
printed as one `case` per state row -- discretize_kernel keeps one Jacobian row per lane, so a lane evaluates only the
handful of non-zeros of ITS row instead of taking part in an 18-direction forward-mode sweep of the whole map.
The generated header is committed (hipcc needs no sympy); tests/test_oracle_model.py checks it against the sympy goldens
and against forward-mode AD of systemFlowMap<Dual1>.

    python tools/gen_model_jacobian.py            # rewrites scpp_amd/csrc/model_jacobian_rows.h
"""
import os

import sympy as sp
from sympy.printing.c import C99CodePrinter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "scpp_amd", "csrc", "model_jacobian_rows.h")


class Printer(C99CodePrinter):
    def _print_Pow(self, e):
        b, ex = e.as_base_exp()
        if ex.is_Integer and 2 <= int(ex) <= 3:
            s = self.parenthesize(b, 100)
            return "(" + "*".join([s] * int(ex)) + ")"
        if ex.is_Integer and -3 <= int(ex) <= -1:
            s = self.parenthesize(b, 100)
            return "(1.0/(" + "*".join([s] * (-int(ex))) + "))"
        return super()._print_Pow(e)

    def _print_Rational(self, e):
        return "(%d.0/%d.0)" % (e.p, e.q)


_FLOWMAPS = None


def flowmaps_from_cpp():
    """The flow maps as sympy expressions, obtained by EXECUTING the C++ plugins with a recording scalar type (tools/flowmap_symbolic.cpp): the model is
    written once, in csrc/model_rocketquat.h.  (Until round 6 this file carried a hand transcription of both maps, kept in step with the C++ by tests.)"""
    global _FLOWMAPS
    if _FLOWMAPS is not None:
        return _FLOWMAPS
    import subprocess

    exe = os.path.join(ROOT, "tools", "bin", "flowmap_symbolic")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-DSCPP_HIP_EMU", "-DSCPP_FLOWMAP_ONLY", "-I" + os.path.join(ROOT, "tests", "emu"),
                           "-I" + os.path.join(ROOT, "scpp_amd", "csrc"), os.path.join(ROOT, "tools", "flowmap_symbolic.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True)
    models, cur = {}, None
    for line in out.splitlines():
        if line.startswith("model "):
            _, name, nx, nu, npar = line.split()
            cur = dict(name=name, x=sp.symbols("x0:%d" % int(nx), real=True), u=sp.symbols("u0:%d" % int(nu), real=True),
                       p=sp.symbols("p0:%d" % int(npar), real=True), f=[])
            models[name] = cur
        elif line.strip():
            lhs, rhs = line.split(" = ", 1)
            assert lhs == "f%d" % len(cur["f"])
            loc = {str(s_): s_ for s_ in list(cur["x"]) + list(cur["u"]) + list(cur["p"])}
            # numeric literals of the C++ source (2., 0.5, 1.) are exact rationals
            cur["f"].append(sp.nsimplify(sp.sympify(rhs, locals=loc), rational=True))
    _FLOWMAPS = models
    return models


def rocketquat():
    """csrc/model_rocketquat.h: RocketQuatModel::systemFlowMap (rocketQuat.cpp:7-37, incl. the un-normalised rotation matrix)"""
    m_ = flowmaps_from_cpp()["RocketQuat"]
    x, u, p, f = m_["x"], m_["u"], m_["p"], m_["f"]
    # OPTIMISATION HINTS, not model definition: wave-uniform, slow sub-expressions (a division or a square root each) evaluated ONCE before the row
    # switch instead of inside its divergent cases: (C name, sympy expression); applied in this order, a later entry may use an earlier symbol
    m = x[0]
    TT = u[0]**2 + u[1]**2 + u[2]**2
    tn = sp.Symbol("t_norm", real=True)
    hoist = [("inv_m", 1 / m), ("t_norm", sp.sqrt(TT)), ("inv_t_norm", 1 / tn),
             ("inv_j0", 1 / p[4]), ("inv_j1", 1 / p[5]), ("inv_j2", 1 / p[6])]  # the last three depend on par only -> aux[]
    return "RocketQuat", x, u, p, f, hoist


def rocket2d():
    """csrc/model_rocketquat.h: Rocket2dModel::systemFlowMap (rocket2d.cpp:7-38)"""
    m_ = flowmaps_from_cpp()["Rocket2d"]
    x, u, p, f = m_["x"], m_["u"], m_["p"], m_["f"]
    hoist = [("inv_m", 1 / p[0]), ("inv_j", 1 / p[1]), ("sin_g", sp.sin(u[0])), ("cos_g", sp.cos(u[0])),
             ("sin_e", sp.sin(x[4])), ("cos_e", sp.cos(x[4]))]
    return "Rocket2d", x, u, p, f, hoist


def lander3dof():
    """csrc/model_lander3dof.h: Lander3dofModel::systemFlowMap (this repository's third model)"""
    m_ = flowmaps_from_cpp()["Lander3dof"]
    x, u, p, f = m_["x"], m_["u"], m_["p"], m_["f"]
    tn = sp.Symbol("t_norm", real=True)
    hoist = [("inv_m", 1 / x[0]), ("t_norm", sp.sqrt(u[0]**2 + u[1]**2 + u[2]**2)), ("inv_t_norm", 1 / tn)]
    return "Lander3dof", x, u, p, f, hoist


def emit(model):
    name, x, u, p, f, hoist = model
    nx, nu = len(x), len(u)
    pr = Printer()
    subs = {**{x[i]: sp.Symbol("x[%d]" % i) for i in range(nx)}, **{u[i]: sp.Symbol("u[%d]" % i) for i in range(nu)},
            **{p[i]: sp.Symbol("par[%d]" % i) for i in range(len(p))}}
    v = list(x) + list(u)
    lines = []
    nnz = 0
    lines.append("struct %sJacobianRows\n{" % name)
    lines.append("    static constexpr int NX = %d, NU = %d;" % (nx, nu))
    lines.append("    // jr[0 .. NX+NU): row `row` of [df/dx | df/du] at (x, u; par); returns f[row]")
    lines.append("    // aux[NAUX]: sub-expressions of the parameters alone, computed once per kernel by prepare()")
    lines.append("    // uaux[NUAUX]: sub-expressions of the input (and parameters) alone, tabulated per stage time by prepareInput()")
    lines.append("    __host__ __device__ static inline double row(int row, const double *x, const double *u, const double *par, const double *aux, const double *uaux, double *jr)\n    {")
    lines.append("        double fr = 0.;")
    lines.append("#pragma unroll\n        for (int j = 0; j < NX + NU; j++)\n            jr[j] = 0.;")
    hsym = [(sp.Symbol(n, real=True), e) for n, e in hoist]
    psyms, usyms = set(p), set(u)
    aux = [(n, e) for n, e in hoist if e.free_symbols and e.free_symbols <= psyms]  # functions of the parameters only
    # functions of the input (and parameters / earlier input-only hoists) alone: the input is a known function of time, so
    # the kernel tabulates them once per (step, stage) instead of once per evaluation
    uaux, unames = [], set()
    for n, e in hoist:
        fs = e.free_symbols
        if (n, e) not in aux and fs and fs <= (psyms | usyms | {sp.Symbol(m, real=True) for m in unames}) and (fs - psyms):
            uaux.append((n, e))
            unames.add(n)
    for n, e in hoist:
        if (n, e) in aux:
            lines.append("        const double %s = aux[%d];" % (n, aux.index((n, e))))
        elif (n, e) in uaux:
            lines.append("        const double %s = uaux[%d];" % (n, uaux.index((n, e))))
        else:
            lines.append("        const double %s = %s;" % (n, pr.doprint(e.subs(subs))))

    def hoisted(e):
        for sy, he in hsym:
            if he.is_Pow and he.exp == -1:
                # negative integer powers of the base become powers of the hoisted reciprocal (positive powers stay)
                base = he.base
                e = e.replace(lambda ex: ex.is_Pow and ex.base == base and ex.exp.is_Integer and ex.exp.is_negative,
                              lambda ex: sy ** (-ex.exp))
            else:
                e = e.subs(he, sy)
        return e

    lines.append("        switch (row)\n        {")
    for i in range(nx):
        entries = [(j, sp.expand_trig(sp.diff(f[i], v[j])) if name == "Rocket2d" else sp.simplify(sp.diff(f[i], v[j]))) for j in range(nx + nu)]
        entries = [(j, e) for j, e in entries if e != 0]
        nnz += len(entries)
        exprs = [hoisted(e) for e in [f[i]] + [e for _, e in entries]]
        rep, red = sp.cse(exprs, symbols=sp.numbered_symbols("t%d_" % i), optimizations="basic")
        lines.append("        case %d:\n        {" % i)
        for s, e in rep:
            lines.append("            const double %s = %s;" % (s, pr.doprint(e.subs(subs))))
        lines.append("            fr = %s;" % pr.doprint(red[0].subs(subs)))
        for (j, _), e in zip(entries, red[1:]):
            lines.append("            jr[%d] = %s;" % (j, pr.doprint(e.subs(subs))))
        lines.append("            break;\n        }")
    lines.append("        default:\n            break;\n        }\n        return fr;\n    }")
    lines.append("    static constexpr int NNZ = %d; // structural non-zeros of [df/dx | df/du]" % nnz)
    lines.append("    static constexpr int NAUX = %d;" % max(1, len(aux)))
    lines.append("    __host__ __device__ static inline void prepare(const double *par, double *aux)\n    {")
    if not aux:
        lines.append("        aux[0] = 0.;")
    for k, (n, e) in enumerate(aux):
        lines.append("        aux[%d] = %s; // %s" % (k, pr.doprint(e.subs(subs)), n))
    lines.append("        (void)par;\n    }")
    lines.append("    static constexpr int NUAUX = %d;" % max(1, len(uaux)))
    lines.append("    __host__ __device__ static inline void prepareInput(const double *u, const double *par, double *uaux)\n    {")
    if not uaux:
        lines.append("        uaux[0] = 0.;")
    for k, (n, e) in enumerate(uaux):
        lines.append("        const double %s = %s;" % (n, pr.doprint(e.subs(subs))))
        lines.append("        uaux[%d] = %s;" % (k, n))
    lines.append("        (void)u;\n        (void)par;\n    }")
    lines.append("};\n")
    return "\n".join(lines)


MAXMON, MAXFAC = 4, 4  # monomials per table slot, factors per monomial


def emit_table(model):
    """Lane-parallel form of the same Jacobian: every structural non-zero (and f) is a short polynomial in the vector
    W = [x | u | par | aux | uaux | h | 1 | partial sums | J | f]; one table slot = one output = at most MAXMON monomials of
    at most MAXFAC factors.  Outputs with more monomials are split into partial sums (level 1) that a level-2 slot adds."""
    name, x, u, p, f, hoist = model
    nx, nu, npar = len(x), len(u), len(p)
    nj = nx + nu
    pr = Printer()
    psyms, usyms, xsyms = set(p), set(u), set(x)
    hsym = [(sp.Symbol(n, real=True), e) for n, e in hoist]
    names = {}
    aux, uaux, hev = [], [], []
    for (n, e), (sy, _) in zip(hoist, hsym):
        fs = e.free_symbols - {s_ for s_, _ in hsym}
        dep = set()
        for s_, _ in hsym:
            if s_ in e.free_symbols:
                dep |= names[s_]
        kinds = set(dep)
        if fs & xsyms:
            kinds.add("x")
        if fs & usyms:
            kinds.add("u")
        names[sy] = kinds
        (hev if "x" in kinds else uaux if "u" in kinds else aux).append((n, e, sy))
    # extra per-evaluation hoists: squares of per-evaluation reciprocals keep the degree within MAXFAC
    sq = [(n + "2", sy * sy, sp.Symbol(n + "2", real=True)) for n, e, sy in hev if e.is_Pow and e.exp == -1]
    hev_all = hev + sq
    W = list(x) + list(u) + list(p) + [sy for _, _, sy in aux] + [sy for _, _, sy in uaux] + [sy for _, _, sy in hev_all]
    W_ONE = len(W)
    base = dict(W_X=0, W_U=nx, W_PAR=nx + nu, W_AUX=nx + nu + npar, W_UAUX=nx + nu + npar + len(aux),
                W_H=nx + nu + npar + len(aux) + len(uaux), W_ONE=W_ONE)
    widx = {sy: i for i, sy in enumerate(W)}

    def hoisted(e):
        for sy, he in hsym:
            if he.is_Pow and he.exp == -1:
                b = he.base
                e = e.replace(lambda ex: ex.is_Pow and ex.base == b and ex.exp.is_Integer and ex.exp.is_negative,
                              lambda ex: sy ** (-ex.exp))
            else:
                e = e.subs(he, sy)
        for n, e2, sy2 in sq:
            root = sp.Symbol(n[:-1], real=True)
            e = e.replace(lambda ex: ex.is_Pow and ex.base == root and ex.exp == 2, lambda ex: sy2)
        return sp.expand(sp.expand_trig(e) if name == "Rocket2d" else e)

    def monomials(e):
        out = []
        for t in sp.Add.make_args(sp.expand(e)):
            c, rest = t.as_coeff_Mul()
            fac = []
            for b_, ex in rest.as_powers_dict().items():
                if b_ == 1:
                    continue
                assert ex.is_Integer and ex > 0 and b_ in widx, (name, t, b_)
                fac += [widx[b_]] * int(ex)
            assert len(fac) <= MAXFAC, (name, t)
            out.append((float(c), fac))
        return out

    v = list(x) + list(u)
    outputs = []  # (target kind, index, monomials)
    for i in range(nx):
        outputs.append(("f", i, monomials(hoisted(f[i]))))
        for j in range(nj):
            d = sp.diff(f[i], v[j])
            if d != 0:
                outputs.append(("J", i * nj + j, monomials(hoisted(sp.simplify(d) if name != "Rocket2d" else d))))
    parts, level1, level2 = [], [], []
    for kind, idx, mons in outputs:
        if len(mons) <= MAXMON:
            level1.append((kind, idx, mons))
            continue
        chunks = [mons[k:k + MAXMON] for k in range(0, len(mons), MAXMON)]
        assert len(chunks) <= MAXMON
        ids = []
        for ch in chunks:
            ids.append(len(parts))
            parts.append(ch)
        level2.append((kind, idx, ids))
    npart = len(parts)
    W_PART = W_ONE + 1
    W_J = W_PART + npart
    W_F = W_J + nx * nj
    NW = W_F + nx

    def tgt(kind, idx):
        return W_F + idx if kind == "f" else W_J + idx

    # largest level-1 outputs first: what spills into the second pass is then the single-monomial entries, and that pass
    # (shared with the level-2 sums) needs fewer monomials per slot
    level1.sort(key=lambda o: -len(o[2]))
    slots = [(W_PART + k, ch) for k, ch in enumerate(parts)] + [(tgt(k_, i_), m_) for k_, i_, m_ in level1]
    n1 = len(slots)
    assert npart <= 64 and n1 <= 64 + 40
    passA, rest = slots[:64], slots[64:]
    passB = rest + [(tgt(k_, i_), [(1.0, [W_PART + q]) for q in ids]) for k_, i_, ids in level2]
    assert len(passB) <= 64
    allslots = passA + [(-1, [])] * (64 - len(passA)) + passB + [(-1, [])] * (64 - len(passB))
    maxmon_b = max([len(m_) for _, m_ in passB] + [1])
    coef, offs, targ = [], [], []
    for t_, mons in allslots:
        targ.append(t_)
        for q in range(MAXMON):
            if q < len(mons):
                c, fac = mons[q]
                fac = fac + [W_ONE] * (MAXFAC - len(fac))
            else:
                c, fac = 0.0, [W_ONE] * MAXFAC
            coef.append(c)
            offs += fac
    L = []
    L.append("// Lane-parallel table form of the same Jacobian (see tools/gen_model_jacobian.py: emit_table)")
    L.append("struct %sJacobianTable\n{" % name)
    L.append("    static constexpr int NX = %d, NU = %d, NP = %d, NJ = %d, NAUX = %d, NUAUX = %d, NH = %d, NPART = %d;" %
             (nx, nu, npar, nj, max(1, len(aux)), max(1, len(uaux)), max(1, len(hev_all)), npart))
    L.append("    // W = [x | u | par | aux | uaux | h | 1 | partial sums | J (row-major, pitch NJ) | f]")
    L.append("    static constexpr int %s, W_PART = %d, W_J = %d, W_F = %d, NW = %d;" %
             (", ".join("%s = %d" % kv for kv in base.items()), W_PART, W_J, W_F, NW))
    L.append("    static constexpr int MAXMON = %d, MAXFAC = %d, NSLOT = 128, NOUT = %d, NLEVEL1 = %d;" % (MAXMON, MAXFAC, len(outputs), n1))
    L.append("    static constexpr int MAXMON_B = %d; // monomials per slot actually used in the second pass" % maxmon_b)
    L.append("    // slot s < 64: first pass (lane s), s >= 64: second pass (lane s - 64); the partial sums are all produced in the first")
    L.append("    __host__ __device__ static inline double coef(int slot, int q)\n    {")
    L.append("        static const double T[NSLOT * MAXMON] = {%s};" % ", ".join(repr(c) for c in coef))
    L.append("        return T[slot * MAXMON + q];\n    }")
    L.append("    __host__ __device__ static inline int factor(int slot, int q, int k) // index into W\n    {")
    L.append("        static const unsigned short T[NSLOT * MAXMON * MAXFAC] = {%s};" % ", ".join(str(o) for o in offs))
    L.append("        return T[(slot * MAXMON + q) * MAXFAC + k];\n    }")
    L.append("    __host__ __device__ static inline int target(int slot) // index into W, -1: idle slot\n    {")
    L.append("        static const short T[NSLOT] = {%s};" % ", ".join(str(t_) for t_ in targ))
    L.append("        return T[slot];\n    }")
    xs_ = {**{x[i]: sp.Symbol("x[%d]" % i) for i in range(nx)}, **{u[i]: sp.Symbol("u[%d]" % i) for i in range(nu)},
           **{p[i]: sp.Symbol("par[%d]" % i) for i in range(npar)}}
    L.append("    // per-evaluation sub-expressions h[NH] (they depend on the state)")
    L.append("    __host__ __device__ static inline void evalHoists(const double *x, const double *u, const double *par, const double *aux, const double *uaux, double *h)\n    {")
    for k, (n, e, sy) in enumerate(aux):
        L.append("        const double %s = aux[%d];" % (n, k))
    for k, (n, e, sy) in enumerate(uaux):
        L.append("        const double %s = uaux[%d];" % (n, k))
    for k, (n, e, sy) in enumerate(hev_all):
        L.append("        const double %s = %s;" % (n, pr.doprint(e.subs(xs_))))
        L.append("        h[%d] = %s;" % (k, n))
    if not hev_all:
        L.append("        h[0] = 0.;")
    L.append("        (void)x; (void)u; (void)par; (void)aux; (void)uaux;\n    }")
    L.append("};\n")
    return "\n".join(L)


def main(out=None):
    out = out or OUT
    hdr = ["// GENERATED by tools/gen_model_jacobian.py (sympy %s) -- do not edit; regenerate after changing a flow map." % sp.__version__,
           "// Analytic rows of [df/dx | df/du] of the model plugins (the build-time analogue of the reference's CppADCodeGen step,",
           "// scpp_core/include/systemDynamics.hpp:109-168).  One `case` per state row: discretize_kernel keeps one Jacobian row per lane.",
           "#pragma once", "#include \"common.h\"", "", "namespace scpp", "{", ""]
    body = [emit(rocketquat()), emit_table(rocketquat()), emit(rocket2d()), emit_table(rocket2d()), emit(lander3dof()), emit_table(lander3dof())]
    with open(out, "w") as fh:
        fh.write("\n".join(hdr) + "\n".join(body) + "} // namespace scpp\n")
    print("wrote", out)


if __name__ == "__main__":
    import sys

    main(sys.argv[1] if len(sys.argv) > 1 else OUT)
