"""Child process of tests/test_gpu_bundled_compiler.py: imports torch FIRST, so that hiprtc compiles the scene-specialised kernels through the
compiler library PyTorch bundles (ROCm 7.0's libamd_comgr) and not the image's (7.2), then checks generic == specialised, bit for bit, ray for ray.
Prints one JSON line.  Not collected by pytest (no test_ prefix)."""
import json
import os
import sys

os.environ["RT_MI355_SYSTEM_COMGR"] = "0"
import torch  # noqa: F401  (first: its HIP runtime and compiler library serve the process)

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np  # noqa: E402
from rtamd import abi, lib, scenes  # noqa: E402

out = {"compiler": lib.prefer_system_compiler(), "cases": {}}
ctx = lib.Context()
for name, (s, cam, prm) in (("04", scenes.scene04(8, seed=5)), ("03", scenes.scene03(8, seed=5, readme_variant=True))):
    ctx.set_debug_flags(16); a = ctx.render_path(s, cam, prm, 128, 96); ka = ctx.last_path_kernel()[0]
    ctx.set_debug_flags(8); b = ctx.render_path(s, cam, prm, 128, 96); kb = ctx.last_path_kernel()[0]
    ctx.set_debug_flags(0)
    out["cases"][name] = {"kernels": [ka, kb], "rays": [int(a["rays"]), int(b["rays"])], "rgb_equal": bool(np.array_equal(a["rgb"], b["rgb"])), "mean": float(a["rgb"].mean())}
s, cam, prm = scenes.scene02(16, seed=3)
ctx.set_debug_flags(8 | 4096); a = ctx.render_whitted(s, cam, prm, 196, 140); ka = ctx.last_whitted_kernel()
ctx.set_debug_flags(16); b = ctx.render_whitted(s, cam, prm, 196, 140); kb = ctx.last_whitted_kernel()
ctx.set_debug_flags(0)
out["cases"]["02"] = {"kernels": [ka, kb], "rays": [int(a["rays"]) + int(a["shadow_rays"]), int(b["rays"]) + int(b["shadow_rays"])],
                      "rgb_equal": bool(np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["bgra"], b["bgra"])), "mean": float(a["rgb"].mean())}
print(json.dumps(out), flush=True)
