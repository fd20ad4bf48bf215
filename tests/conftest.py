import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _build_oracle_lib():
    """The oracle's C half (serial Lewiner MC) is built on demand; it is test infrastructure."""
    so = os.path.join(ROOT, 'oracle', 'libmc_oracle.so')
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])


@pytest.fixture(scope='session', autouse=True)
def _bounded_torch_threads():
    """The torch-CPU oracle runs small per-chunk GEMMs: with every hardware thread of a 256-thread GPU host it is several
    times slower than with 32 (bench.py measured the same), which made the parity suite take ten minutes."""
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
