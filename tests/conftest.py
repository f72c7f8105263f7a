import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


ORACLE_THREADS = 16   # a 1-GPU box is a 16-CPU share of a 256-thread host


@pytest.fixture(scope='session', autouse=True)
def _cpu_threads():
    """torch's intra-op pool defaults to half the HOST's hardware threads (128 on the GPU boxes, which are
    16-CPU shares of that host): the CPU oracle then runs 2.8 x slower than with 16 - 32 threads
    (tests/oracle_threads.py, profiles/r05_oracle_threads.txt: one T = 3 clip 31.8 s at 128 threads, 11.5 s at
    16 / 32) -- three full-size oracle runs were 4 of the GPU suite's 8.7 minutes."""
    import torch
    if torch.get_num_threads() > ORACLE_THREADS:
        torch.set_num_threads(ORACLE_THREADS)


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
