import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    # materialise: NpzFile re-reads and re-inflates the member on every [] access
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_rules():
    return load_golden("rules.npz")


@pytest.fixture(scope="session")
def golden_mcts():
    return load_golden("mcts.npz")


@pytest.fixture(scope="session")
def golden_episodes():
    return load_golden("episodes.npz")


@pytest.fixture(scope="session")
def golden_arena():
    return load_golden("arena.npz")
