import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _build_oracle():
    """The C restatement of the sampler is test infrastructure: build it on demand."""
    import subprocess
    so = os.path.join(ROOT, 'oracle', '_build', 'libmsda_oracle.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
