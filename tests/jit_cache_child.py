"""Child process of tests/test_gpu_bundled_compiler.py::test_jit_cache_on_disk: renders 04 with the scene-specialised kernel (RT_MI355_JIT_CACHE is set by the
parent) and prints {kernel, seconds of the first render (compile or cache load included), checksum of the frame}.  Not collected by pytest."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes  # noqa: E402

ctx = lib.Context()
s, cam, prm = scenes.scene04(4, seed=2)
ctx.set_debug_flags(16)
t0 = time.perf_counter()
g = ctx.render_path(s, cam, prm, 128, 96)
dt = time.perf_counter() - t0
print(json.dumps({"kernel": ctx.last_path_kernel()[0], "first_render_s": dt, "rays": int(g["rays"]), "sha": hashlib.sha256(g["rgb"].tobytes()).hexdigest()}), flush=True)
