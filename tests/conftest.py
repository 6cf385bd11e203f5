import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liblocus_oracle.so if needed."""
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def hip():
    """The product's host-array front end; fails loudly without the .so / a GPU."""
    import locus_amd
    from locus_amd import _lib
    assert _lib.device_count() >= 1, "no HIP device visible"
    return locus_amd.core
