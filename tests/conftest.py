import os
import sys

import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: tens of seconds of CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library(request):
    """A checkout without the (git-ignored) shared object -- e.g. a fresh clone on the GPU box -- builds it once before
    the first test; hipcc cross-compiles without a GPU, ~2.5 min.  This is the product build, not a fallback: the tests
    still fail loudly if the build fails."""
    from tce_rvos_amd import build as b
    if not os.path.exists(b.LIB):
        b.build(verbose=False)
    so = os.path.join(ROOT, "oracle", "libmsda_ref.so")
    mk = os.path.join(ROOT, "oracle", "build_oracle.py")
    if not os.path.exists(so) and os.path.exists(mk):
        import runpy
        runpy.run_path(mk, run_name="__main__")
