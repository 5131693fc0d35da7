import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_addoption(parser):
    parser.addoption("--x6", action="store_true", default=False,
                     help="run the gpu suite with the bf16 x 6 arithmetic as the package default (arith.set_default) and in every bare kernel table: "
                          "evidence that every parity test holds with that arithmetic at the same tolerances (the end-to-end parity tests are "
                          "parametrised over both arithmetics anyway)")


@pytest.fixture(scope="session", autouse=True)
def _x6_everywhere(request):
    if not request.config.getoption("--x6"):
        yield
        return
    from mfn_import import load_package
    load_package()
    from music_fader_nets_amd import arith, hipops
    init = hipops.HipOps.__init__
    prev = arith.set_default(arith.BF16X6)          # models follow the package default; bare HipOps tables (the `ops` fixtures) are patched below

    def patched(self, *a, **k):
        init(self, *a, **k)
        self.dw_x6 = True
    hipops.HipOps.__init__ = patched
    try:
        yield
    finally:
        hipops.HipOps.__init__ = init
        arith.set_default(prev)
