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

K = float(os.environ.get("FUZZ_SCALE", "1"))      # the whole scene times K (the kernels' tMin and shadow offsets stay: other conditioning)
def random_scene(seed):
    g = np.random.default_rng(seed)
    s = scenes.Scene()
    mats = []
    for _ in range(int(g.integers(2, 5))):
        r2r = float(g.choice([1.0, 1.0, 0.0, 0.5])) if seed % 3 else float(g.choice([1.0, 0.0]))     # every third seed: chains only
        mats.append(s.phong(tuple(g.uniform(0.1, 1.0, 3)), float(g.uniform(0, 1)), float(g.choice([1, 2, 8, 16, 50, 64.5, 300, 1000])),
                            float(g.choice([1.0, 0.8, 0.4, 0.0])), r2r, float(g.choice([1.0, 0.7, 1.5]))))
    pick = lambda: int(g.choice(mats + ([-1] if g.random() < 0.1 else [])))
    s.plane(pick(), (0, K * float(g.uniform(-5, -2)), 0), (0, 1, 0))
    if g.random() < 0.6: s.plane(pick(), (0, 0, K * float(g.uniform(-40, -20))), (float(g.uniform(-.3, .3)), float(g.uniform(-.3, .3)), 1))
    for _ in range(int(g.integers(1, 6))):
        s.sphere(pick(), (K * float(g.uniform(-6, 6)), K * float(g.uniform(-2, 4)), K * float(g.uniform(-18, -6))), K * float(g.uniform(0.5, 3)))
    for _ in range(int(g.integers(0, 3))):
        v = g.uniform(-8, 8, (3, 3)); v[:, 2] = g.uniform(-22, -8, 3)
        s.triangle(pick(), *[tuple(map(float, r * K)) for r in v])
    for _ in range(int(g.integers(1, 4))):
        s.light((K * float(g.uniform(-10, 10)), K * float(g.uniform(3, 12)), K * float(g.uniform(-10, 4))), tuple(g.uniform(0.2, 0.9, 3)), K * float(g.uniform(0.3, 2.0)))
    return s, scenes.camera(float(g.uniform(40, 90))), int(g.integers(1, 7)), int(g.choice([0, 0, 1, 3, 8]))

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
            form = "w" if samples > 0 and ctx.last_whitted_form() == 1 else "m"      # w: wavefront form (weighted sum, within 4e-7; the library compiles at most 64 scenes per process, later ones run the generic kernels)
            line += f"  {name} k{ctx.last_whitted_kernel()}{form} {'rays=' if same_rays else 'RAYS DIFFER'} px {npx} max {dmax:.2e}"
            if not same_rays or dmax > 1e-3: bad += 1
        print(line, flush=True)
    for name, (npx, tot, dmax) in totals.items():
        print(f"{name}: {npx} of {tot} pixels differ from the oracle ({npx / tot:.2e}), largest difference {dmax:.2e}")
    print("FAILURES", bad)
    return 1 if bad else 0
sys.exit(main())
