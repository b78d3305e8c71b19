import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def scene_mod():
    import mirres_restir_nerf_mesh_amd as M
    return M.scene


@pytest.fixture(autouse=True)
def _fixed_global_seeds():
    """Every test starts from the same global generator states (torch's is drawn from by torch.nn.init in MLPTexture3D), so a test's inputs
    never depend on which tests ran before it."""
    import random
    import numpy as np
    random.seed(0); np.random.seed(0)
    try:
        import torch
        torch.manual_seed(0)
    except ImportError:
        pass
    yield
