import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
    """-> dict of numpy arrays (flat keys, see tests/golden/make_golden.py)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(store, prefix):
    """All entries under `prefix.` with the prefix stripped."""
    p = prefix + "."
    return {k[len(p):]: v for k, v in store.items() if k.startswith(p)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
