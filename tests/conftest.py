import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native libraries built in-tree (hipcc cross-compiles without a GPU)."""
    from vcf2prot_amd import build
    return build.build_all()


@pytest.fixture(scope="session")
def coracle():
    from sir_oracle import COracle
    return COracle()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kat_transcripts.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_ctx(built):
    """A live engine context; fails loudly (no fallback) when the extension or the GPU is missing."""
    from vcf2prot_amd.engine import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
