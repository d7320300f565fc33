"""Stress scene at reduced counts: per-pixel agreement of the global-memory kernel with the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))     # orc.py: the checker's binding
from rtamd import lib, scenes, abi
import orc as oracle
ctx = lib.Context()
full, cam, _ = scenes.scene_stress(n_spheres=600, grid=12, spp=4, seed=3)
s = scenes.Scene()
for p in [p for p in full.prims if p.type == 0]:
    s.sphere(s.diffuse((0.5, 0.5, 0.5)), (p.f[0], p.f[1], p.f[2]), p.f[3])
w, h = 96, 54
prm = scenes.path_params(abi.RT_PATH_LIGHTS, 1, max_depth=8, seed=3)
acc = torch.zeros(h, w, 4, device="cuda")
ctx.set_debug_flags(4)
ctx.path_accumulate_device(s, cam, prm, w, h, acc.data_ptr()); ctx.sync(); torch.cuda.synchronize()
g = acc[:, :, 3].cpu().numpy().astype(np.int64)
o = oracle.render_path_px(s, cam, prm, w, h, threads=8)["px_rays"].astype(np.int64)
ys, xs = np.nonzero(g != o)
print("pixels differing", len(ys), "gpu total", g.sum(), "cpu total", o.sum())
import collections
print(sorted(collections.Counter((int(g[y, x]), int(o[y, x])) for y, x in zip(ys, xs)).items()))

print([(int(y), int(x)) for y, x in zip(ys, xs) if g[y, x] == 9 and o[y, x] == 3][:10])
print([(int(y), int(x)) for y, x in zip(ys, xs) if g[y, x] == 2 and o[y, x] == 9][:10])
