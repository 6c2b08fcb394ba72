import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
H = (100, 10000)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'))
        return cache[name]
    return load


@pytest.fixture(scope='session')
def ref_masks(golden):
    """Cut-off masks as evaluated by the NumPy that ran the reference for tests/golden."""
    g = golden('g1_ao_zone')
    rec = np.unpackbits(g['mask_rec'])[:6400].reshape(80, 80).astype(bool)
    res = np.unpackbits(g['mask_res'])[:6400].reshape(80, 80).astype(bool)
    return rec, res


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max())
