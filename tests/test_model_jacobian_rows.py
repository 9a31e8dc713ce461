"""Generated analytic Jacobian rows (scpp_amd/csrc/model_jacobian_rows.h, tools/gen_model_jacobian.py) against the sympy
golden vectors G1 and against forward-mode AD of the same plugin's systemFlowMap<Dual1> -- compiled for the host through the
emulation header (the functions are __host__ __device__)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

SRC = r'''
#include "hip_emu.h"
#include "model_rocketquat.h"
#include "model_lander3dof.h"
#include <cstdio>
using namespace scpp;
template <class M> int run()
{
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NJ = NX + NU;
    int n; if (scanf("%d", &n) != 1) return 1;
    for (int q = 0; q < n; q++)
    {
        double x[NX], u[NU], p[NP];
        for (auto &v : x) if (scanf("%lf", &v) != 1) return 1;
        for (auto &v : u) if (scanf("%lf", &v) != 1) return 1;
        for (auto &v : p) if (scanf("%lf", &v) != 1) return 1;
        double aux[M::JacobianRows::NAUX], uaux[M::JacobianRows::NUAUX];
        M::JacobianRows::prepare(p, aux);
        M::JacobianRows::prepareInput(u, p, uaux);
        for (int r = 0; r < NX; r++)
        {
            double jr[NJ];
            const double f = M::JacobianRows::row(r, x, u, p, aux, uaux, jr);
            printf("%.17g", f);
            for (int j = 0; j < NJ; j++) printf(" %.17g", jr[j]);
            printf("\n");
        }
        // the lane-parallel table form, evaluated the way discretize_kernel does (two passes over 64 slots)
        {
            using T = typename M::JacobianTable;
            double W[T::NW] = {};
            for (int i = 0; i < NX; i++) W[T::W_X + i] = x[i];
            for (int i = 0; i < NU; i++) W[T::W_U + i] = u[i];
            for (int i = 0; i < NP; i++) W[T::W_PAR + i] = p[i];
            for (int i = 0; i < M::JacobianRows::NAUX; i++) W[T::W_AUX + i] = aux[i];
            for (int i = 0; i < M::JacobianRows::NUAUX; i++) W[T::W_UAUX + i] = uaux[i];
            double h[T::NH];
            T::evalHoists(x, u, p, aux, uaux, h);
            for (int i = 0; i < T::NH; i++) W[T::W_H + i] = h[i];
            W[T::W_ONE] = 1.;
            for (int pass = 0; pass < 2; pass++)
            {
                double val[64];
                for (int l = 0; l < 64; l++)
                {
                    const int slot = pass * 64 + l;
                    double a = 0.;
                    for (int q = 0; q < T::MAXMON; q++)
                    {
                        double m = T::coef(slot, q);
                        for (int k = 0; k < T::MAXFAC; k++) m *= W[T::factor(slot, q, k)];
                        a += m;
                    }
                    val[l] = a;
                }
                for (int l = 0; l < 64; l++)
                    if (T::target(pass * 64 + l) >= 0) W[T::target(pass * 64 + l)] = val[l];
            }
            for (int r = 0; r < NX; r++)
            {
                printf("%.17g", W[T::W_F + r]);
                for (int j = 0; j < NJ; j++) printf(" %.17g", W[T::W_J + r * NJ + j]);
                printf("\n");
            }
        }
        // forward-mode AD of the plugin's flow map, one direction at a time
        for (int d = 0; d < NJ; d++)
        {
            Dual1 xd[NX], ud[NU], fd[NX];
            for (int i = 0; i < NX; i++) xd[i] = Dual1(x[i], d == i ? 1. : 0.);
            for (int i = 0; i < NU; i++) ud[i] = Dual1(u[i], d == NX + i ? 1. : 0.);
            M::template systemFlowMap<Dual1>(xd, ud, p, fd);
            for (int i = 0; i < NX; i++) printf("%s%.17g", i ? " " : "", fd[i].d);
            printf("\n");
        }
    }
    return 0;
}
int main(int argc, char **argv) { return argv[1][0] == 'q' ? run<RocketQuatModel>() : argv[1][0] == 'l' ? run<Lander3dofModel>() : run<Rocket2dModel>(); }
'''


@pytest.fixture(scope="module")
def rows_bin(tmp_path_factory):
    d = tmp_path_factory.mktemp("rows")
    (d / "rows.cpp").write_text(SRC)
    exe = str(d / "rows")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DSCPP_HIP_EMU", "-I" + os.path.join(ROOT, "tests", "emu"),
                           "-I" + os.path.join(ROOT, "scpp_amd", "csrc"), str(d / "rows.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("name,flag,nx,nu", [("rocketquat", "q", 14, 4), ("rocket2d", "2", 6, 2), ("lander3dof", "l", 7, 3)])
def test_generated_rows_match_sympy_golden_and_forward_ad(rows_bin, name, flag, nx, nu):
    g = np.load(os.path.join(GOLDEN, f"{name}_jacobians.npz"))
    n = g["x"].shape[0]
    inp = str(n) + "\n" + "\n".join(" ".join(repr(float(v)) for v in np.concatenate([g["x"][i], g["u"][i], g["par"][i]])) for i in range(n))
    out = subprocess.run([rows_bin, flag], input=inp, capture_output=True, text=True, check=True).stdout
    vals = np.array([[float(t) for t in line.split()] for line in out.strip().split("\n")], dtype=object)
    per = nx + nx + (nx + nu)
    for i in range(n):
        blk = vals[i * per:(i + 1) * per]
        rows = np.array([r for r in blk[:nx]], dtype=float)          # [f | A row | B row]
        tab = np.array([r for r in blk[nx:2 * nx]], dtype=float)      # the same from the lane-parallel table
        ad = np.array([r for r in blk[2 * nx:]], dtype=float).T        # [NX][NJ]
        f, A, B = rows[:, 0], rows[:, 1:1 + nx], rows[:, 1 + nx:]
        scale = max(1.0, np.abs(g["A"][i]).max(), np.abs(g["B"][i]).max())
        assert np.abs(f - g["f"][i]).max() <= 1e-13 * max(1.0, np.abs(g["f"][i]).max())
        assert np.abs(A - g["A"][i]).max() <= 1e-13 * scale and np.abs(B - g["B"][i]).max() <= 1e-13 * scale
        assert np.abs(np.hstack([A, B]) - ad).max() <= 1e-13 * scale
        assert np.abs(tab - rows).max() <= 1e-13 * max(scale, np.abs(rows[:, 0]).max())


def test_generated_header_is_up_to_date(tmp_path):
    """The committed header is what tools/gen_model_jacobian.py produces from the C++ flow maps (round 6: the generator EXECUTES
    csrc/model_rocketquat.h's systemFlowMap with an expression-recording scalar, tools/flowmap_symbolic.cpp -- a model's dynamics are written in one
    place; a change of a flow map without regenerating fails here)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "tools", "gen_model_jacobian.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    gen.OUT = str(tmp_path / "rows.h")
    gen.main()
    assert open(gen.OUT).read() == committed
