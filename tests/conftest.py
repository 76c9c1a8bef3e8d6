import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; built with gcc on first use)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def engine():
    """The product engine on cuda:0.  Fails loudly (no fallback) when the HIP library or the
    GPU is missing."""
    from fakebob_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="session")
def small_system():
    """UBM + 3 speakers, C=256, D=72 (fast for the CPU oracle)."""
    from fakebob_amd.models import synthetic_gmm_system
    return synthetic_gmm_system(n_speakers=3, C=256, D=72)


@pytest.fixture(scope="session")
def full_system():
    """BASELINE config 2 shape: UBM + 5 speakers, C=2048, D=72."""
    from fakebob_amd.models import synthetic_gmm_system
    return synthetic_gmm_system(n_speakers=5, C=2048, D=72)
