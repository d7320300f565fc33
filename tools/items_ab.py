"""Item form (persistent waves, work counter) vs per-lane regeneration (debug flag 512): same frame? how fast?
python3 tools/items_ab.py [check|time04|time03|time04full]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
import numpy as np
from rtamd import abi, lib, scenes
ctx = lib.Context()
what = sys.argv[1] if len(sys.argv) > 1 else "check"

def both(s, cam, prm, w, h, flags=0):
    ctx.set_debug_flags(flags | 1024); a = ctx.render_path(s, cam, prm, w, h); ka = ctx.last_path_kernel()
    ctx.set_debug_flags(flags | 512); b = ctx.render_path(s, cam, prm, w, h); kb = ctx.last_path_kernel()
    ctx.set_debug_flags(0)
    return a, b, ka, kb

if what == "check":
    for name, (s, cam, prm), (w, h) in (("04", scenes.scene04(32), (784, 561)), ("03", scenes.scene03(16), (784, 561)), ("04 small", scenes.scene04(8), (98, 70)),
                                        ("04 odd", scenes.scene04(5), (61, 37)), ("03 1spp", scenes.scene03(1), (196, 140)), ("04 2000spp tiny", scenes.scene04(2000), (16, 16))):
        for flags in (0, 8, 2):
            a, b, ka, kb = both(s, cam, prm, w, h, flags)
            rel = np.abs(a["rgb"] - b["rgb"]) / np.maximum(np.abs(b["rgb"]), 1e-3)
            print(f"{name:16s} flags {flags}: kernels {ka[0]}/{kb[0]} rays {a['rays']} vs {b['rays']} equal {a['rays'] == b['rays']}  max rel diff {rel.max():.2e}  bytes differ {(a['bgra'] != b['bgra']).mean():.2e}  ms {a['kernel_ms']:.3f} vs {b['kernel_ms']:.3f}", flush=True)
    # per-pixel query counts (debug flag 4)
    s, cam, prm = scenes.scene04(4)
    ctx.set_debug_flags(4 | 1024); a = ctx.render_path(s, cam, prm, 98, 70); ctx.set_debug_flags(4 | 512); b = ctx.render_path(s, cam, prm, 98, 70); ctx.set_debug_flags(0)
    print("flag 4 (per-pixel queries) equal:", a["rays"] == b["rays"], "per-pixel counts equal:", bool((a["rgb"] == b["rgb"]).all()))
else:
    (s, cam, prm), (w, h), fpr = {"time04": (scenes.scene04(256), (3840, 2160), 249), "time03": (scenes.scene03(256), (1920, 1080), 168),
                                  "time04full": (scenes.scene04(1024), (3840, 2160), 249)}[what]
    for flags in (0, 512, 0, 512):
        ctx.set_debug_flags(flags)
        r = ctx.render_path(s, cam, prm, w, h)
        print(f"{what} flags {flags} kernel {ctx.last_path_kernel()[0]}: {r['kernel_ms']:.2f} ms rays {r['rays']} ns/kray {r['kernel_ms'] * 1e6 / r['rays'] * 1e3:.3f} frac {r['rays'] * fpr / (r['kernel_ms'] * 1e-3) / 157.3e12:.4f}", flush=True)
