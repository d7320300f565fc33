"""The scheduled BVH kernel on the stress scene, three renders (for profiling passes)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import abi, lib, scenes
ctx = lib.Context()
s, cam, prm = scenes.scene_stress(spp=16)
ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
ctx.set_debug_flags(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for rep in range(3):
    g = ctx.render_path(s, cam, prm, 1920, 1080)
    print(f"{g['kernel_ms']:.1f} ms {g['rays']} rays", flush=True)
