import os, sys
sys.path.insert(0, "software-raytracing_amd")
os.environ["RT_MI355_JIT"] = "2"
from rtamd import abi, lib, scenes
ctx = lib.Context()
for spp, sl in ((256, 2), (512, 4), (1024, 8), (2048, 16), (256, 1), (1024, 1), (2048, 1)):
    os.environ["RT_MI355_SLICES"] = str(sl)
    s, cam, prm = scenes.scene04(spp, seed=1)
    best = min(ctx.render_path(s, cam, prm, 3840, 2160)["kernel_ms"] for _ in range(3))
    print(f"spp {spp} slices {sl} (block = {spp // sl} spp): {best:.2f} ms  {best / spp * 1024:.2f} ms per 1024 spp", flush=True)
