import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bluenoise():
    return np.fromfile(os.path.join(ROOT, "assets", "bluenoise256.rgba"), np.uint8).reshape(256, 256, 4)


@pytest.fixture(scope="session")
def cornell_path():
    return os.path.join(ROOT, "assets", "cornell_box.obj")


@pytest.fixture(scope="session")
def native_lib():
    """The product library, built in-tree if needed (hipcc cross-compiles gfx950 without a GPU)."""
    from capsaicin_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build_native()
    return capi.lib()
