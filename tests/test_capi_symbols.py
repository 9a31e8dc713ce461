"""The C-ABI library builds for gfx950 (hipcc cross-compiles without a GPU), loads, and exports every symbol
include/scpp_hip.h declares.  No compute calls here (no GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "scpp_hip.h")).read()
    return sorted(set(re.findall(r"\b(scpp_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), s
    lib.scpp_hip_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.scpp_hip_version()


def test_python_binding_covers_header():
    from scpp_amd import _lib

    assert sorted(_lib.SYMBOLS) == declared_symbols()


def test_library_contains_gfx950_code_object(hip_lib):
    blob = open(hip_lib, "rb").read()
    assert b"gfx950" in blob and b"ipm_kernel" in blob and b"discretize_kernel" in blob


def test_committed_profiles_are_of_these_sources():
    """The PMC and parity summaries bench.py imports (roofline.traffic, config.parity) carry the content hash of the kernel sources + floating-point
    build flags they were measured on (tools/csrc_hash.py).  The newest committed ones must be of THESE sources: a kernel change without a new
    measurement session would otherwise ship a bench line that says `"stale": true`."""
    import glob
    import json
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import csrc_hash

    here = csrc_hash.csrc_sha()
    sys.path.insert(0, ROOT)
    import bench  # the selection rule is bench.py's own: the newest summary of the newest round

    pmc, d = bench.measured_traffic()
    assert d["csrc_sha"] == here, (pmc, here)
    par = bench.parity_summary()
    assert par["csrc_sha"] == here and par["stale"] is False, (par["imported_from"], here)
