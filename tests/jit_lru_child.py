"""Child process of tests/test_gpu_bundled_compiler.py::test_jit_table_replaces_stale_kernels (RT_MI355_JIT_MAX_SCENES=2 is set by the parent): the table of
compiled scenes holds two; a third scene evicts the least recently used one only when that one is stale (not used for 4 x the capacity in lookups), else it
runs on the generic kernel.  Prints the kernel kind (2 specialised, 1 generic) of every render in order.  Not collected by pytest."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np  # noqa: E402
from rtamd import lib, scenes  # noqa: E402

ctx = lib.Context()
ctx.set_debug_flags(16)


def scene(k):
    s, cam, prm = scenes.scene04(4, seed=2)
    s.sphere(s.diffuse((0.5, 0.5, 0.5)), (0.0, 0.0, -2.0 + 0.1 * k), 0.3)      # a different scene per k
    return s, cam, prm


kinds, frames = [], {}
def render(k):
    s, cam, prm = scene(k)
    g = ctx.render_path(s, cam, prm, 96, 64)
    kinds.append(ctx.last_path_kernel()[0])
    if k in frames: assert np.array_equal(frames[k], g["rgb"]) or kinds[-1] == 1      # a recompiled kernel renders the same frame
    frames.setdefault(k, g["rgb"])


render(1); render(2)                 # the table is full
render(3)                            # scene 1 is not stale yet (age 2 < 8): scene 3 runs on the generic kernel
for _ in range(8): render(2)         # eight more lookups, all of scene 2
render(3)                            # scene 1 has not been used for >= 8 lookups: it makes room, scene 3 is compiled
render(1)                            # scene 1 is gone and scene 3 / 2 are fresh: generic kernel
render(3); render(2)                 # both still specialised
print(json.dumps({"kinds": kinds}), flush=True)
