import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
