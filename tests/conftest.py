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


@pytest.fixture(scope="session")
def dev_ctx(built):
    """A context of the DEVELOPMENT library (libv2p_bench.so: the same engine compiled with V2P_BENCH_VARIANTS -- the A/B switches of the
    builders and launchers, slices, PATCH images, the grid builders of rounds 2-3; csrc/bench/v2p_bench.h).  The tests of those paths use it;
    every other GPU test runs on the product library (gpu_ctx)."""
    from vcf2prot_amd import build
    from vcf2prot_amd.engine import Context
    build.build_bench()
    ctx = Context(0, development=True)
    yield ctx
    ctx.close()
