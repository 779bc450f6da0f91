import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    out = {k: d[k] for k in d.files}
    # compact inputs of the larger fixtures (tools/gen_golden_512.py): uint8 slice values, int8 cotangent signs
    if 'x_u8' in out:
        out['x'] = out.pop('x_u8').astype(np.float32) * np.float32(2.0 / 255.0) - np.float32(1.0)
    if 'r_i8' in out:
        out['r'] = out.pop('r_i8').astype(np.float32)
    return out


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith('.npz'))


@pytest.fixture(scope='session')
def golden():
    return load_golden
