"""A/B of the scene-specialised path kernel's code generations on the headline workload (04 scene, 3840x2160, 1024 spp) and on 03
(1920x1080, 256 spp): python tools/perf_jit.py [gens...]   e.g.  python tools/perf_jit.py 1 2
Prints kernel time, rays/s and the fraction of the FP32 vector peak (algorithmic FLOPs/ray); checks that the
generations render the same small frame."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np
from rtamd import lib, scenes
gens = sys.argv[1:] or ["1r", "2r", "2q"]        # <generation><r|q>: r = per-lane regeneration, q = queue form
ctx = lib.Context()
ctx.set_debug_flags(16)
small = {}
for which, (w, h, spp), fpr in (("04", (3840, 2160, 1024), 249), ("03", (1920, 1080, 256), 168)):
    for rnd in range(2):
        for g in gens:
            os.environ["RT_MI355_JIT_GEN"] = g[0]; os.environ["RT_MI355_PATH"] = g[1:] or "q"
            s, cam, prm = scenes.scene04(spp) if which == "04" else scenes.scene03(spp)
            best = None
            for rep in range(3):
                r = ctx.render_path(s, cam, prm, w, h)
                assert ctx.last_path_kernel()[0] == 2, ctx.last_path_kernel()
                best = r if best is None or r["kernel_ms"] < best["kernel_ms"] else best
            frac = best["rays"] * fpr / (best["kernel_ms"] * 1e-3) / 157.3e12
            print(f"{which} gen {g} round {rnd}: {best['kernel_ms']:.2f} ms  {best['rays'] / best['kernel_ms'] / 1e6:.1f} Grays/s  frac {frac:.3f}  rays {best['rays']}", flush=True)
            if rnd == 0:
                ss, scam, sprm = scenes.scene04(8, seed=3) if which == "04" else scenes.scene03(8, seed=3)
                small[(which, g)] = ctx.render_path(ss, scam, sprm, 256, 160)
    a = small[(which, gens[0])]
    for g in gens[1:]:
        b = small[(which, g)]
        d = np.abs(a["rgb"] - b["rgb"]) / np.maximum(np.abs(a["rgb"]), 1e-3)
        print(f"{which}: gen {gens[0]} vs gen {g}: rays {a['rays']} vs {b['rays']}, identical {np.array_equal(a['rgb'], b['rgb'])}, pixels within 1e-5: {(d.max(axis=-1) <= 1e-5).mean():.5f}, max rel {d.max():.2e}")
