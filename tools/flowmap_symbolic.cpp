// Host tool of tools/gen_model_jacobian.py: prints the flow map of every model plugin (csrc/model_rocketquat.h) as expressions in x0.., u0.., p0..,
// obtained by RUNNING the plugin's systemFlowMap<T, PT> with a scalar type that records what is done to it -- the counterpart of the reference's CppAD
// tape (scpp_core/include/systemDynamics.hpp:109-131: the model is written once, as code, and everything else is derived from executing it).
//   g++ -std=c++17 -DSCPP_HIP_EMU -DSCPP_FLOWMAP_ONLY -Itests/emu -Iscpp_amd/csrc tools/flowmap_symbolic.cpp -o tools/bin/flowmap_symbolic
// Output, per model:   model <Name> <NX> <NU> <NP>   followed by NX lines   f<i> = <expression>
#include <cstdio>
#include <sstream>
#include <string>

namespace scpp
{
struct Sym
{
    std::string e;
    Sym() : e("0") {}
    Sym(double v)
    {
        std::ostringstream o;
        o.precision(17);
        o << v;
        e = v < 0 ? "(" + o.str() + ")" : o.str();
    }
    explicit Sym(const std::string &s) : e(s) {}
};
inline Sym bin(const Sym &a, const char *op, const Sym &b) { return Sym("(" + a.e + " " + op + " " + b.e + ")"); }
inline Sym operator+(const Sym &a, const Sym &b) { return bin(a, "+", b); }
inline Sym operator-(const Sym &a, const Sym &b) { return bin(a, "-", b); }
inline Sym operator*(const Sym &a, const Sym &b) { return bin(a, "*", b); }
inline Sym operator/(const Sym &a, const Sym &b) { return bin(a, "/", b); }
inline Sym operator+(double a, const Sym &b) { return Sym(a) + b; }
inline Sym operator-(double a, const Sym &b) { return Sym(a) - b; }
inline Sym operator*(double a, const Sym &b) { return Sym(a) * b; }
inline Sym operator/(double a, const Sym &b) { return Sym(a) / b; }
inline Sym operator+(const Sym &a, double b) { return a + Sym(b); }
inline Sym operator-(const Sym &a, double b) { return a - Sym(b); }
inline Sym operator*(const Sym &a, double b) { return a * Sym(b); }
inline Sym operator/(const Sym &a, double b) { return a / Sym(b); }
inline Sym operator-(const Sym &a) { return Sym("(-" + a.e + ")"); }
inline Sym dsqrt(const Sym &a) { return Sym("sqrt(" + a.e + ")"); }
inline Sym dsin(const Sym &a) { return Sym("sin(" + a.e + ")"); }
inline Sym dcos(const Sym &a) { return Sym("cos(" + a.e + ")"); }
} // namespace scpp

#include "model_rocketquat.h"
#include "model_lander3dof.h"

template <class Model>
static void dump(const char *name)
{
    using scpp::Sym;
    Sym x[Model::NX], u[Model::NU], p[Model::NP], f[Model::NX];
    for (int i = 0; i < Model::NX; i++)
        x[i] = Sym("x" + std::to_string(i));
    for (int i = 0; i < Model::NU; i++)
        u[i] = Sym("u" + std::to_string(i));
    for (int i = 0; i < Model::NP; i++)
        p[i] = Sym("p" + std::to_string(i));
    Model::template systemFlowMap<Sym, Sym>(x, u, p, f);
    std::printf("model %s %d %d %d\n", name, Model::NX, Model::NU, Model::NP);
    for (int i = 0; i < Model::NX; i++)
        std::printf("f%d = %s\n", i, f[i].e.c_str());
}

int main()
{
    dump<scpp::RocketQuatModel>("RocketQuat");
    dump<scpp::Rocket2dModel>("Rocket2d");
    dump<scpp::Lander3dofModel>("Lander3dof");
    return 0;
}
