import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))     # orc.py: the checker's binding
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Before anything imports torch: the tests run the scene-specialised kernels as the benchmark runs them, compiled by the image's
# own compiler library rather than the older one PyTorch bundles (rtamd/lib.py prefer_system_compiler; the C++ example binaries,
# which several tests compare with, use the system runtime anyway).  RT_MI355_SYSTEM_COMGR=0 keeps PyTorch's.
from rtamd import lib as _lib_early  # noqa: E402
JIT_COMPILER = _lib_early.prefer_system_compiler()


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


def nest_scene():
    """Exact ties and the sphere tMax rule in one scene: coincident duplicates that differ in material (the LAST in scene order
    wins), glass spheres nested and intersecting so that rays inside one sphere find another surface before its far wall
    (Common/Ray.hpp:127-134 hands the hit to the enclosing sphere when it comes later), rotated boxes, a rectangle coplanar
    with a box face, an inverted shell around the nest."""
    from rtamd import scenes
    s = scenes.Scene()
    white = s.diffuse((0.8, 0.8, 0.8)); red = s.diffuse((0.8, 0.1, 0.1)); green = s.diffuse((0.1, 0.8, 0.1))
    glass = s.dielectric(0.66); glass2 = s.dielectric(0.8); metal = s.metal((0.9, 0.9, 0.9), 0.05); lm = s.light_material((9, 9, 9))
    s.plane(white, (0, -2, 0), (0, 1, 0))
    s.sphere(glass, (0, 0, -6), 1.0); s.sphere(red, (0, 0, -6), 0.5)
    s.sphere(glass2, (0.3, 0, -6), 1.6)
    s.sphere(glass, (2.2, 0, -6), 1.2); s.sphere(glass2, (3.2, 0.2, -6), 1.2)
    s.sphere(red, (-3, 0, -7), 1.0); s.sphere(green, (-3, 0, -7), 1.0)
    s.rect(red, (0, 3, -7), (2, 2), (90, 0, 0)); s.rect(green, (0, 3, -7), (2, 2), (90, 0, 0))
    s.triangle(red, (-6, -2, -9), (-2, -2, -11), (-4, 3, -10)); s.triangle(metal, (-6, -2, -9), (-2, -2, -11), (-4, 3, -10))
    s.box(metal, (3, -1, -9), (2, 2, 2), (10, 30, 5)); s.box(glass, (-1.5, -1.2, -4), (1.2, 1.6, 1.2), (0, 20, 0))
    s.rect(green, (3, -1, -8), (2, 2), (0, 0, 0))
    s.sphere(white, (0.3, 0, -6), 2.4, flip=True)
    s.rect(lm, (0, 6, -6), (6, 6), (90, 0, 0))
    return s, scenes.camera(70.0, (0, 0.5, 1.0), (-3, 0, 0))
