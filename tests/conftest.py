import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))     # orc.py: the checker's binding
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def recorded():
    with open(os.path.join(GOLDEN, "reference_recorded_outputs.json")) as f:
        return json.load(f)


def load_png_rgb(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)))[:, :, :3]


@pytest.fixture(scope="session")
def ctx():
    """One rt_ctx for the whole GPU session (the library is the thing under test; no fallback)."""
    # Some tests hand torch device buffers to the library.  PyTorch ships its own libamdhip64; whichever HIP runtime is
    # loaded first serves the whole process, and torch only finds the GPU through its own, so torch goes first.
    import torch  # noqa: F401
    from rtamd import lib
    c = lib.Context()
    yield c
    c.close()


def frac_within(g, o, rel, floor=1e-3):
    d = np.abs(np.asarray(g, np.float64) - np.asarray(o, np.float64))
    ok = (d <= rel * np.maximum(np.abs(o), floor)).all(axis=-1)
    return float(ok.mean())
