"""Outcome-sorted form (debug flag 2048) vs the item form: same frame bit for bit? how fast?  python3 tools/sorted_ab.py [check|time03|time04]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
import numpy as np
from rtamd import abi, lib, scenes
ctx = lib.Context()
what = sys.argv[1] if len(sys.argv) > 1 else "check"
if what == "check":
    for name, (s, cam, prm), (w, h) in (("03", scenes.scene03(16), (196, 140)), ("04", scenes.scene04(16), (196, 140)), ("03 readme", scenes.scene03(8, readme_variant=True), (98, 70)),
                                        ("04 odd", scenes.scene04(5), (61, 37)), ("03 depth 3", (lambda t: (t[0], t[1], scenes.path_params(abi.RT_PATH_SKY, 8, max_depth=3)))(scenes.scene03(8)), (98, 70))):
        for flags in (0, 8):
            ctx.set_debug_flags(flags | 1024 | 2048); a = ctx.render_path(s, cam, prm, w, h); fa = ctx.last_path_form()
            ctx.set_debug_flags(flags | 1024); b = ctx.render_path(s, cam, prm, w, h); fb = ctx.last_path_form()
            ctx.set_debug_flags(0)
            print(f"{name:12s} flags {flags}: forms {fa}/{fb} rays {a['rays']} vs {b['rays']} equal {a['rays'] == b['rays']} frames bit-identical {np.array_equal(a['rgb'], b['rgb'])} max abs diff {np.abs(a['rgb'] - b['rgb']).max():.2e}", flush=True)
else:
    (s, cam, prm), (w, h), fpr = {"time04": (scenes.scene04(256), (3840, 2160), 249), "time03": (scenes.scene03(256), (1920, 1080), 168)}[what]
    for flags in ((int(sys.argv[2]),) * 2 if len(sys.argv) > 2 else (0, 2048, 0, 2048)):      # a third argument: one form only (PMC passes)
        ctx.set_debug_flags(flags)
        r = ctx.render_path(s, cam, prm, w, h)
        print(f"{what} flags {flags} form {ctx.last_path_form()}: {r['kernel_ms']:.2f} ms rays {r['rays']} ns/kray {r['kernel_ms'] * 1e6 / r['rays'] * 1e3:.3f} frac {r['rays'] * fpr / (r['kernel_ms'] * 1e-3) / 157.3e12:.4f}", flush=True)
