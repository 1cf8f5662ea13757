import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The test process is an application of the library: it asks (explicitly, before torch / HIP is loaded) for a hardware queue per stream,
# as bench.py does -- the package itself never writes the environment (adaptive_edge_aware_jpeg_amd/_lib.py).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def lena():
    import numpy as np
    from PIL import Image as PILImage
    return np.asarray(PILImage.open(os.path.join(GOLDEN, "lena.png")).convert("RGB")).astype(np.float32) / np.float32(255.0)


def golden_image(name):
    """float32 RGB in [0, 1] (uint8 levels / 255, image.py:80) of tests/golden/<name>.png -- "lena" or "natural/<x>" (the reference's
    own test images, converted by tests/golden/make_natural_fixtures.py)"""
    import numpy as np
    from PIL import Image as PILImage
    return np.asarray(PILImage.open(os.path.join(GOLDEN, name + ".png")).convert("RGB")).astype(np.float32) / np.float32(255.0)
