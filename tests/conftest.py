import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build the host-only helpers (oracle, synth) if missing; the HIP library is prebuilt by build()."""
    from nextpolish2_amd import synth
    from oracle import np2_oracle
    synth.lib()
    np2_oracle.lib()


@pytest.fixture(scope="session")
def small_haploid():
    from nextpolish2_amd.synth import Synth
    s = Synth(40000, depth=30, seed=21, read_len_mean=8000.0, read_len_sd=1500.0)
    return s, [s.yak(21)]


@pytest.fixture(scope="session")
def small_diploid():
    from nextpolish2_amd.synth import Synth
    s = Synth(60000, depth=30, seed=22, diploid=True, read_len_mean=9000.0, read_len_sd=1500.0)
    return s, [s.yak(21), s.yak(31)]
