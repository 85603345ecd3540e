import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """-> (cfg dict, state dict of torch tensors, dict of torch tensors)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    cfg = json.loads(str(z['cfg']))
    sd, arrays = {}, {}
    for k in z.files:
        if k == 'cfg':
            continue
        t = torch.from_numpy(z[k])
        if k.startswith('sd/'):
            sd[k[3:]] = t
        else:
            arrays[k] = t
    return cfg, sd, arrays


@pytest.fixture(scope='session')
def golden():
    return load_golden
