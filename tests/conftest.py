import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """The CPU oracle is (re)built before any test runs, i.e. before anything has initialised the GPU: building
    it from inside a GPU test would start a child process (make, gcc) from a process that holds a device."""
    import oracle_lib
    try:
        oracle_lib.build()
    except Exception as e:   # the tests that need it will say so
        print(f"conftest: oracle build failed: {e}")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def rdsp():
    """The product package with its HIP library loaded (fails loudly if missing)."""
    import radiodsp_sdr_rx_amd as R
    R.load()
    return R


@pytest.fixture(params=["default", "direct", "fd"])
def front_form(request):
    """Stage A3 of the chains a test makes without choosing: the library's default (frequency domain with frames
    of one granule beside a tail stage, on 16-lane rows without one: the bits do not depend on the call split),
    the direct form (rdsp_chain_set_fir_variant 0,
    split-invariant too) and the frequency-domain decimator with 448-sample frames (variant 2, what bench.py
    runs).  The GPU test modules use it module-wide, so every chain test covers all three (and through the default both of its kernels: rows without a tail stage, wave-wide frames with one); a test that is about
    one form sets it explicitly, which wins."""
    from radiodsp_sdr_rx_amd.chain import Chain
    old = Chain.default_fir_variant
    Chain.default_fir_variant = {"default": None, "direct": 0, "fd": 2}[request.param]
    yield request.param
    Chain.default_fir_variant = old
