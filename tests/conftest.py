import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: the kernel sources on the wave EMULATOR, oracle runs, multi-process gloo jobs) is ~12 minutes of
    independent single-threaded tests: spread it over the host cores with pytest-xdist when that plugin is there (same image: it is).
    Only for that marker expression -- the GPU suite stays in THIS process (one GPU; the driver records the libraries the pytest process
    loads) -- and never on a worker, with an explicit -n, or with SCPP_TESTS_SERIAL=1.  Tests that share a build directory or a
    rendezvous port carry an xdist_group mark (--dist loadgroup keeps a group on one worker)."""
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("SCPP_TESTS_SERIAL"):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) is not None:
        return None
    if (getattr(config.option, "markexpr", "") or "").strip() != "not gpu" or getattr(config.option, "usepdb", False):
        return None
    n = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 8)
    if n >= 2:
        config.option.numprocesses = n
        config.option.dist = "loadgroup"
    return None


@pytest.fixture(scope="session")
def emu_lib():
    """CPU emulation build of the HIP kernel sources (tests/emu/hip_emu.h) -- test infrastructure only."""
    import __graft_entry__ as g

    alt = os.environ.get("SCPP_EMU_LIBRARY")  # e.g. an -fsanitize=address,undefined build of the same sources
    if alt:
        return alt
    return g.build_emu()


@pytest.fixture(scope="session")
def hip_lib():
    import __graft_entry__ as g

    alt = os.environ.get("SCPP_HIP_LIBRARY")  # an alternative build of the same sources (tools/r02_variant.sh)
    if alt:
        return alt
    if not os.path.exists(g.HIP_LIB):
        g.build_hip()
    return g.HIP_LIB


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def model():
    import scpp_amd

    return scpp_amd.RocketQuat().loadParameters()
