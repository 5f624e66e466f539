import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')


@pytest.fixture(scope='session', autouse=True)
def _cpu_threads():
    """The CPU oracle runs on the cores this process may really use (affinity + cgroup quota): on
    the GPU box os.cpu_count() is the whole machine and over-subscribing the share is many times
    slower."""
    import torch
    import bench
    torch.set_num_threads(bench.host_cores())


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    return load
