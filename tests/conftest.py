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
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def keyset_small(oracle):
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS with the LWE dimension cut to 12 (same N, k, bases, levels):
    exercises exactly the same kernels, 60x fewer CMUX iterations."""
    return oracle.KeySet(oracle.param_message_2_carry_2(lwe_dim=12), seed=0x5EED0001)


@pytest.fixture(scope="session")
def keyset_full(oracle):
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS at full size (n = 742); ~3 s of key generation."""
    return oracle.KeySet(oracle.param_message_2_carry_2(), seed=0x5EED0001, fourier=False)
