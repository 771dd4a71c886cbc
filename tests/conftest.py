import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def formula_sd():
    from mind_amd.weights import formula_state_dict
    return formula_state_dict(as_torch=True)


@pytest.fixture(scope="session")
def golden_predictor():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "predictor.npz")))


@pytest.fixture(scope="session")
def hip_predictor(formula_sd):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mind_amd.predictor import HipPredictor
    hp = HipPredictor(0)
    hp.load_state_dict(formula_sd)
    yield hp
    hp.close()
