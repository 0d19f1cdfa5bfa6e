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
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def warm_s_state(g):
    """state_dict of the network_yolox_s_warm fixture: float tensors are stored as bf16 bit patterns (the reference's recorded step
    ran from exactly these values), integer buffers as they are."""
    import torch
    sd = {}
    for k, v in g.items():
        if k.startswith("state16/"):
            sd[k[8:]] = torch.from_numpy(v.copy()).view(torch.bfloat16).float()
        elif k.startswith("state/"):
            sd[k[6:]] = torch.from_numpy(v.copy())
    return sd


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _release_device_memory():
    """A detector keeps every activation / gradient / dz buffer of its traced shapes resident (YOLOX-x at 1280x1280
    batch 16: tens of GB) and model <-> runner reference each other, so a finished test's buffers only go away after a
    cycle collection: collect and hand the blocks back after every test, or the full-size tests of one pytest process
    pile up to the 288 GB of the GPU."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except Exception:
        pass
