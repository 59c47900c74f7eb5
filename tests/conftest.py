import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'lstm-unet_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


# small nets used across tests ---------------------------------------------------------
def tiny_net(k_lstm=3, widths=(8, 8, 12, 16), up=(12, 8, 8, 8)):
    return {
        'down_conv_kernels': [[(3, w), (3, w)] for w in widths],
        'lstm_kernels': [[(k_lstm, w)] for w in widths],
        'up_conv_kernels': [[(3, up[0]), (3, up[0])], [(3, up[1]), (3, up[1])], [(3, up[2]), (3, up[2])],
                            [(3, up[3]), (3, up[3]), (1, 3)]],
    }


def c1_net():
    """BASELINE config-1: 32-channel net, 3x3 everywhere."""
    return tiny_net(3, (32, 32, 32, 32), (32, 32, 32, 32))
