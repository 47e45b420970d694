import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(REPO, "tests", "golden", "reference_vectors.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def matcha_sd():
    from emojivoice_amd import weights as W

    return W.synthetic_matcha_state(178, 109)


@pytest.fixture(scope="session")
def voc_sd():
    from emojivoice_amd import weights as W

    return W.synthetic_hifigan_state()
