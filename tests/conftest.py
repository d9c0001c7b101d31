import os
import sys

import pytest

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')   # autolabel_amd/__init__.py: before HIP initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


_LAST = ('test_gpu_quality.py', 'test_gpu_rccl.py', 'test_parallel_gloo.py', 'test_gpu_parallel.py')


def pytest_collection_modifyitems(session, config, items):
    """With `-x`, a failure in the multi-process data-parallel tests must not hide the single-GPU rows (8a R1-R15):
    the long quality gate and the multi-process files are collected last, whatever their names sort as."""
    rank = lambda it: next((i + 1 for i, name in enumerate(_LAST) if it.nodeid.split('::')[0].endswith(name)), 0)
    items.sort(key=rank)      # (stable: the order inside a file and among the other files is kept)
