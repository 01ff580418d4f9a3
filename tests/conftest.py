import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = graft.load_package()
    if not os.path.exists(p.lib_path()):
        p.build_library()
    return p


@pytest.fixture(scope="session")
def O():
    return graft.load_oracle()


@pytest.fixture(scope="session")
def ctx(pkg):
    """GPU context through the C-ABI; fails loudly (no CPU fallback) when there is no GPU."""
    c = pkg.Context(0)
    yield c
    c.close()


def load_cloud(path):
    """Reference .gird/.wlop format: first line N, then N lines 'x y z' (Main_KSS_List.cpp:65-94)."""
    return np.loadtxt(path, skiprows=1, dtype=np.float64)


@pytest.fixture(scope="session")
def ref_pairs():
    d = os.path.join(GOLDEN, "ref_data")
    out = {}
    for sub, name in [("registration", "Bunny"), ("registration", "Horse"), ("registration", "ant"),
                      ("registration_scale", "Bunny")]:
        out[(sub, name)] = (load_cloud(os.path.join(d, sub, name + ".gird")), load_cloud(os.path.join(d, sub, name + ".wlop")))
    return out
