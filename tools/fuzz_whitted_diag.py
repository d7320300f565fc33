"""Where a fuzz scene's pixels differ from the oracle, and which term does it: python3 tools/fuzz_whitted_diag.py seed [seed ...]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from rtamd import lib, scenes
import orc as oracle
src = open(os.path.join(ROOT, "tools", "fuzz_whitted.py")).read().replace("sys.exit(main())", "")
exec(compile(src, "fuzz_whitted", "exec"))
ctx = lib.Context()
W, H = 128, 96
def diff(s, cam, prm, tag):
    ctx.set_debug_flags(8 | 4096)
    try: g = ctx.render_whitted(s, cam, prm, W, H)
    finally: ctx.set_debug_flags(0)
    o = oracle.render_whitted(s, cam, prm, W, H, threads=8)
    d = np.abs(g["rgb"].astype(np.float64) - o["rgb"]).reshape(H, W, 3)
    ys, xs = np.nonzero(d.max(axis=-1) > 0)
    print(f"  {tag}: {len(ys)} px differ", flush=True)
    for y, x in list(zip(ys, xs))[:3]:
        print(f"    ({x},{y}) gpu {g['rgb'].reshape(H, W, 3)[y, x]} oracle {o['rgb'].reshape(H, W, 3)[y, x]}")
for seed in map(int, sys.argv[1:]):
    s, cam, depth, samples = random_scene(seed)
    print(f"seed {seed}: depth {depth} samples {samples} shininess {[m.shininess for m in s.materials]}")
    diff(s, cam, scenes.whitted_params(depth, samples, seed=seed), "as generated")
    diff(s, cam, scenes.whitted_params(depth, 0, seed=seed), "hard shadows")
    diff(s, cam, scenes.whitted_params(0, samples, seed=seed), "depth 0")
    keep = [m.specular_intensity for m in s.materials]
    for m in s.materials: m.specular_intensity = 0.0
    s._c = None
    diff(s, cam, scenes.whitted_params(depth, samples, seed=seed), "no specular term")
    for m, k in zip(s.materials, keep): m.specular_intensity = k
    for m in s.materials: m.shininess = 1.0
    s._c = None
    diff(s, cam, scenes.whitted_params(depth, samples, seed=seed), "shininess 1")
