"""02 (soft shadows) HIP vs the CPU oracle at equal seeds: how many pixels agree to 1e-6 / 1e-4, and by how much the rest differ (a shadow sample that
flips moves a pixel by 1/samples of one light's contribution: the fast shadow tests may decide a grazing sample differently from the strict ones)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import lib, scenes
import orc as oracle
ctx = lib.Context()
for ss in (1, 16, 64):
    s, cam, prm = scenes.scene02(ss, seed=3)
    o = oracle.render_whitted(s, cam, prm, 392, 280, threads=16)
    for flags, name in ((8 | 4096, "generic monolithic"), (16, "specialised")):
        ctx.set_debug_flags(flags); g = ctx.render_whitted(s, cam, prm, 392, 280); ctx.set_debug_flags(0)
        rel = (np.abs(g["rgb"] - o["rgb"]) / np.maximum(np.abs(o["rgb"]), 1e-3)).max(axis=2)
        print(f"02 x{ss} {name}: rays equal {(g['rays'], g['shadow_rays']) == (o['rays'], o['shadow_rays'])} exact {float((rel == 0).mean()):.4f} within 1e-6 {float((rel <= 1e-6).mean()):.4f} within 1e-4 {float((rel <= 1e-4).mean()):.4f} max rel {float(rel.max()):.3e} pixels beyond 1e-4: {int((rel > 1e-4).sum())}", flush=True)
