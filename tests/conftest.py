import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    import oracle
    return oracle.port()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden_ref.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session.  Fails loudly (no skip, no
    CPU fallback) if the HIP library or the device is missing."""
    import bitmagic_amd as bm
    c = bm.context(0)
    yield c
    c.close()


@pytest.fixture(params=["direct", "rows"])
def agg_path(request, ctx):
    """both implementations of a single aggregation call: the one-launch kernel straight from the descriptor tables
    (k_direct, the default for short operand lists and small collections) and the row-table pipeline kernels"""
    ctx.set_tuning("direct_cols", 384 if request.param == "direct" else 0)
    yield request.param
    ctx.set_tuning("direct_cols", 384)
