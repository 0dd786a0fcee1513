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
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


class Cfg:
    """plain object with the EcgVitConfig fields (the oracle only reads attributes)"""

    def __init__(self, **kw):
        d = dict(max_signal_length=2560, patch_size=64, num_channels=12, hidden_size=512, num_hidden_layers=8,
                 num_attention_heads=8, intermediate_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                 num_class=71)
        d.update(kw)
        self.__dict__.update(d)


def load_micro(tag):
    z = np.load(os.path.join(GOLDEN, f'micro_{tag}.npz'))
    spec = json.loads(bytes(z['cfg']).decode())
    return z, spec


@pytest.fixture(scope='session')
def host_contract():
    with open(os.path.join(GOLDEN, 'host_contract.json')) as f:
        return json.load(f)


@pytest.fixture(autouse=True)
def _release_device_temporaries():
    yield
    try:
        import hiputil
        hiputil.release()
    except Exception:
        pass
