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


_OBSERVED = {}


@pytest.fixture(scope='session')
def observed():
    """record(name, **numbers): the parity tests state what they OBSERVED, not only that it stayed under a threshold; the
    collection is written to gpurun_out/parity_observed.json at the end of the session (copied to profiles/ per round)."""
    def record(name, **vals):
        _OBSERVED[name] = {k: (float(v) if not isinstance(v, (str, bool, int)) else v) for k, v in vals.items()}
        print('observed', name, _OBSERVED[name])
    return record


def pytest_sessionfinish(session, exitstatus):
    if not _OBSERVED:
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_observed.json'), 'w') as fh:
            json.dump(_OBSERVED, fh, indent=1, sort_keys=True)
    except OSError:
        pass
