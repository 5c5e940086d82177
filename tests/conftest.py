import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the oracle is test infrastructure (oracle/fcm_oracle.c header)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def fcm():
    import flag_complex_mcmc_amd
    flag_complex_mcmc_amd.lib()
    return flag_complex_mcmc_amd


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
