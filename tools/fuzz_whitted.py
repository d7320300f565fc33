"""Random Whitted scenes (01 hard shadows / 02 soft shadows) through every kernel form against the oracle.
python3 tools/fuzz_whitted.py [n_scenes] [first_seed]
Per scene and form: ray counts equal?  pixels whose floats differ, largest difference.  Exit status 1 on a ray-count
mismatch or a pixel off by more than 1e-3."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import lib, scenes
import orc as oracle

sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz_scenes import random_whitted_scene      # noqa: E402
K = float(os.environ.get("FUZZ_SCALE", "1"))      # the whole scene times K
def random_scene(seed): return random_whitted_scene(seed, K)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = lib.Context()
    W, H = 128, 96
    bad = 0; totals = {}
    for seed in range(first, first + n):
        s, cam, depth, samples = random_scene(seed)
        prm = scenes.whitted_params(depth, samples, seed=seed)
        o = oracle.render_whitted(s, cam, prm, W, H, threads=8)
        line = f"seed {seed:3d} depth {depth} samples {samples} prims {len(s.prims)} rays {o['rays']:7d}:"
        for name, flags in (("generic", 8 | 4096), ("generated", 16), ("default", 0)):
            ctx.set_debug_flags(flags)
            try: gpu = ctx.render_whitted(s, cam, prm, W, H)
            finally: ctx.set_debug_flags(0)
            same_rays = (gpu["rays"], gpu["shadow_rays"]) == (o["rays"], o["shadow_rays"])
            d = np.abs(gpu["rgb"].astype(np.float64) - o["rgb"])
            npx = int((d.max(axis=-1) > 0).sum()); dmax = float(d.max())
            t = totals.setdefault(name, [0, 0, 0.0]); t[0] += npx; t[1] += W * H; t[2] = max(t[2], dmax)
            form = "w" if samples > 0 and ctx.last_whitted_form() == 1 else "m"      # w: wavefront form (weighted sum, within 4e-7; the library compiled at most 64 scenes per process then (256 now), later ones run the generic kernels)
            line += f"  {name} k{ctx.last_whitted_kernel()}{form} {'rays=' if same_rays else 'RAYS DIFFER'} px {npx} max {dmax:.2e}"
            if not same_rays or dmax > 1e-3: bad += 1
        print(line, flush=True)
    for name, (npx, tot, dmax) in totals.items():
        print(f"{name}: {npx} of {tot} pixels differ from the oracle ({npx / tot:.2e}), largest difference {dmax:.2e}")
    print("FAILURES", bad)
    return 1 if bad else 0
sys.exit(main())
