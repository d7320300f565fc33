"""02: per-pixel comparison of the HIP frame with the oracle at one seed (where the few differing pixels are)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))     # orc.py: the checker's binding
from rtamd import lib, scenes
import orc as oracle
ctx = lib.Context()
for S in (1, 4):
    s, cam, prm = scenes.scene02(S, seed=1)
    w, h = 196, 140
    g = ctx.render_whitted(s, cam, prm, w, h); o = oracle.render_whitted(s, cam, prm, w, h)
    d = g["rgb"] - o["rgb"]
    bad = np.argwhere(np.abs(d).max(axis=-1) > 1e-3)
    print("S", S, "bad", len(bad), "sum diff", d.sum(), "pos", (d.max(axis=-1) > 1e-3).sum(), "neg", (d.min(axis=-1) < -1e-3).sum())
    for y, x in bad[:40]:
        print(y, x, g["rgb"][y, x], o["rgb"][y, x])
