"""Sweep of A/B switches of the scene-specialised path kernel: python3 tools/perf_sweep.py 04|03 spp cfg [cfg ...]
cfg = gen:path:defs:threshold   e.g. 1:r:RT_NO_SPARE:16   2:r::24   (defs: comma-separated RT_MI355_JIT_DEFS names; threshold -1 = default)"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np
from rtamd import lib, scenes
which, spp = sys.argv[1], int(sys.argv[2])
ctx = lib.Context()
os.environ["RT_MI355_JIT"] = "2"
(w, h), fpr = ((1920, 1080), 168) if which == "03" else ((3840, 2160), 249)
ref = None
for rnd in range(2):
    for cfg in sys.argv[3:]:
        gen, path, defs, thr = (cfg.split(":") + ["", "", "-1"])[:4]
        os.environ["RT_MI355_JIT_GEN"] = gen; os.environ["RT_MI355_PATH"] = path or "r"; os.environ["RT_MI355_JIT_DEFS"] = defs
        ctx.set_regen_threshold(int(thr or -1))
        s, cam, prm = scenes.scene03(spp) if which == "03" else scenes.scene04(spp)
        best = None
        for rep in range(3):
            r = ctx.render_path(s, cam, prm, w, h)
            assert ctx.last_path_kernel()[0] == 2, ctx.last_path_kernel()
            best = r if best is None or r["kernel_ms"] < best["kernel_ms"] else best
        note = ""
        if rnd == 0:
            ss, scam, sprm = scenes.scene03(8, seed=3) if which == "03" else scenes.scene04(8, seed=3)
            g = ctx.render_path(ss, scam, sprm, 256, 160)
            if ref is None: ref = g
            d = np.abs(g["rgb"] - ref["rgb"]) / np.maximum(np.abs(ref["rgb"]), 1e-3)
            note = f"  vs first cfg: rays {g['rays']}/{ref['rays']} identical {np.array_equal(g['rgb'], ref['rgb'])} max rel {d.max():.1e}"
        print(f"{which} {cfg:28s} round {rnd}: {best['kernel_ms']:8.2f} ms {best['rays'] / best['kernel_ms'] / 1e6:7.1f} Grays/s frac {best['rays'] * fpr / (best['kernel_ms'] * 1e-3) / 157.3e12:.3f}{note}", flush=True)
