import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "distantspeechrecognition-mirror_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def headset():
    return np.load(os.path.join(GOLDEN, "Headset1_16k_s16.npy")).astype(np.float32)


def load_proto(name):
    hg = np.load(os.path.join(GOLDEN, "proto_%s.npy" % name))
    return hg[0], hg[1]


@pytest.fixture(scope="session")
def protos():
    return {"M256-m4-r1": (256, 4, 1) + load_proto("M256-m4-r1"),
            "M512-m2-r2": (512, 2, 2) + load_proto("M512-m2-r2"),
            "M512-m2-r3": (512, 2, 3) + load_proto("M512-m2-r3")}


@pytest.fixture(scope="session")
def dsr():
    """The product binding; fails loudly when the HIP library is missing."""
    import dsr._capi as capi
    capi.load()
    return capi


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")
