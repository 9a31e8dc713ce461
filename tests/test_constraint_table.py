"""The model constraint tables (scpp_amd/csrc/constraint_table.h) and what is derived from them at compile time, checked
against the reference-shaped (literal) problem the oracle assembles from the same model constraints: every cone / LP row of
rocketQuat.cpp:70-144 and rocket2d.cpp:46-84 is either in the table's active set at a node or a constant there."""
import os
import subprocess

import numpy as np

from conftest import ROOT

PROBE = r'''
#include <cstdio>
#include "constraint_table.h"
using namespace scpp::ipm;
template <class P> void dump(const char *name, int K)
{
    using D = Derived<P>;
    int cones = 0, lps = 0;
    for (int k = 0; k < K; k++)
    {
        const unsigned a = D::activeMask(k, K);
        for (int c = 0; c < D::NCONES; c++) cones += (a >> c) & 1u;
        for (int l = 0; l < P::NLP; l++) lps += (a >> (D::NCONES + l)) & 1u;
    }
    std::printf("%s NS %d HS_N %d NVU %d LP0 %d first %u last %u act_first %u act_mid %u act_last %u cones %d lps %d\n", name, D::NS, D::HS_N,
                D::NVU, D::LP0, D::fixedMask(0, K), D::fixedMask(K - 1, K), D::activeMask(0, K), D::activeMask(1, K), D::activeMask(K - 1, K), cones, lps);
}
int main()
{
    dump<RocketQuatSC>("rocketquat", 50);
    dump<Rocket2dSC>("rocket2d", 30);
    return 0;
}
'''


def _probe(tmp_path):
    src = tmp_path / "probe.cpp"
    src.write_text(PROBE)
    exe = tmp_path / "probe"
    subprocess.check_call(["g++", "-std=c++17", "-DSCPP_HIP_EMU", "-I" + os.path.join(ROOT, "tests", "emu"),
                           "-I" + os.path.join(ROOT, "scpp_amd", "csrc"), "-o", str(exe), str(src)], stderr=subprocess.DEVNULL)
    out = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        t = line.split()
        out[t[0]] = {t[i]: int(t[i + 1]) for i in range(1, len(t), 2)}
    return out


def test_tables_against_the_literal_problem(oracle, tmp_path):
    d = _probe(tmp_path)
    rq, r2 = d["rocketquat"], d["rocket2d"]
    # RocketQuat: slack layout 17 (trust) + 3 + 3 + 3 + 4 + 3 cone rows + 2 LP rows; 27 Hessian entries in 5 small blocks
    assert (rq["NS"], rq["HS_N"], rq["NVU"], rq["LP0"]) == (35, 27, 16, 33)
    assert rq["first"] == 0x1FFF and rq["last"] == sum(1 << j for j in (1, 2, 3, 4, 5, 6, 8, 9, 11, 12, 13, 14))
    # the literal problem: 301 cones = sigma cone + 50 trust regions + 50 x 5 model cones; 1474 LP rows = sigma >= 0.001
    # + 1372 nu box rows + sum(nu_bound) row + 50 mass rows + 50 minimum-thrust rows
    s = oracle.SC(oracle.ROCKETQUAT, K=50); s.set_solver(0); s.set_tolerances(1e-8, 1e-6, 1e-6, 2)
    s.solve()
    m = s.meta()
    assert (m["ncones"], m["l"]) == (301, 1474)
    # table: cones that are constants at a node (all variables presolved there) are dropped -- glide slope, tilt and rate at
    # node 0 and node K-1 (3 + 3), the mass row at node 0
    assert rq["cones"] == 50 + 250 - 6 and rq["cones"] + 1 == m["ncones"] - 6
    assert rq["lps"] == 100 - 1 and rq["lps"] + 1 + 1372 + 1 == m["l"] - 1
    # Rocket2d: 17 + 2 cone rows + 8 LP rows; box rows give four 1x1 Hessian entries, the glide-slope cone a 2x2 block
    assert (r2["NS"], r2["HS_N"], r2["NVU"], r2["LP0"]) == (27, 8, 8, 19)
    assert r2["first"] == 0xFF00 | 0x3F and r2["last"] == 0xFF00 | 0x7F  # padding variables 8..15 are always presolved
    s2 = oracle.SC(oracle.ROCKET2D, K=30); s2.solve()
    m2 = s2.meta()
    # literal (tests/golden/sc_regression.json run): 61 cones = sigma + 30 trust + 30 glide slope; 590 LP rows = 1 + 2*6*29 + 1 + 8*30
    assert (m2["ncones"], m2["l"]) == (61, 590)
    assert r2["cones"] == 30 + 30 - 2 and r2["cones"] + 1 == m2["ncones"] - 2   # glide slope is a constant at nodes 0 and K-1
    # box rows on eta, omega (nodes 0, K-1) and on the gimbal angle (node K-1: U(0, K-1) = 0) are constants: 4 + 4 + 2 dropped
    assert r2["lps"] == 8 * 30 - 10 and r2["lps"] + 1 + 2 * 6 * 29 + 1 == m2["l"] - 10
