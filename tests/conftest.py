import importlib
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

PKG_NAME = "multimodal-detection-consistency_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has hyphens, hence importlib)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def gpu_engine(pkg):
    """A consistency-only engine (no towers) on cuda:0; fails loudly without the HIP library."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    eng = pkg.TVCEngine(device="cuda:0")
    yield eng
    eng.close()
