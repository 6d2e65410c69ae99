import json
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
    """-> (meta dict, {array name: ndarray}).  Fixtures come from tools/make_golden.py (reference outputs)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta']))
    return meta, {k: z[k] for k in z.files if k != 'meta'}


def golden_clip(seed, n, sha):
    """Regenerate a fixture's input clip from its seed and verify it is the clip the reference saw."""
    from salsa_amd.synth import sha256_of, synth_clip
    y = synth_clip(seed, n)
    assert sha256_of(y) == sha, 'synthetic clip generator drifted from the golden fixtures (seed %d)' % seed
    return y


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as o
    o.build()
    o.set_threads(min(8, o.max_threads()))
    return o
