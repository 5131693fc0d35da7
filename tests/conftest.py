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
    parser.addoption("--f32", action="store_true", default=False,
                     help="run the suite with the fp32 MFMA as the package default arithmetic (arith.set_default) - models and bare kernel tables alike; "
                          "the package default is bf16 x 6 since round 5.  The end-to-end parity tests are parametrised over both arithmetics anyway")
    parser.addoption("--x6", action="store_true", default=False, help="the bf16 x 6 arithmetic as the package default (what it is anyway; kept from round 4)")


@pytest.fixture(scope="session", autouse=True)
def _default_arithmetic(request):
    f32, x6 = request.config.getoption("--f32"), request.config.getoption("--x6")
    if not (f32 or x6):
        yield
        return
    assert not (f32 and x6), "--f32 and --x6 exclude each other"
    from mfn_import import load_package
    load_package()
    from music_fader_nets_amd import arith
    prev = arith.set_default(arith.F32 if f32 else arith.BF16X6)      # models AND bare HipOps tables (the `ops` fixtures) start from the package default
    try:
        yield
    finally:
        arith.set_default(prev)
