import json
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


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_small():
    arrays = dict(np.load(os.path.join(GOLDEN, "loss_small.npz")))
    return arrays, load_json("loss_small.json")


@pytest.fixture(scope="session")
def golden_large():
    return load_json("loss_large.json")


@pytest.fixture(scope="session")
def golden_metrics():
    return dict(np.load(os.path.join(GOLDEN, "metric_inputs.npz"))), load_json("metrics.json")
