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
