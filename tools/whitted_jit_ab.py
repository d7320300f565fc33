"""Whitted kernels: generated strict queries (round 3) vs the generic run-table walk.  01: scene-specialised vs generic kernel; 02: specialised with /
without the generated strict tree-walk queries (RT_MI355_JIT_NO_STRICT=1: round 2's specialised kernel, shadow samples only).  Same image? how fast?
python3 tools/whitted_jit_ab.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np
from rtamd import lib, scenes
ctx = lib.Context()

def timed(fn, n=7):
    r = [fn() for _ in range(n)]
    ms = sorted(x["kernel_ms"] for x in r)
    return r[-1], ms[0], ms[len(ms) // 2]

def flops(g, a, b): return (g["rays"] * a + g["shadow_rays"] * b)

for w, h in ((784, 561), (3840, 2160)):
    s, cam, prm = scenes.scene01()
    ctx.set_debug_flags(8); g, g_lo, g_med = timed(lambda: ctx.render_whitted(s, cam, prm, w, h)); kg = ctx.last_whitted_kernel()
    ctx.set_debug_flags(16); j, j_lo, j_med = timed(lambda: ctx.render_whitted(s, cam, prm, w, h)); kj = ctx.last_whitted_kernel()
    ctx.set_debug_flags(0)
    print(f"01 {w}x{h}: kernels {kg}/{kj} rays equal {(g['rays'], g['shadow_rays']) == (j['rays'], j['shadow_rays'])} rgb bit-identical {bool(np.array_equal(g['rgb'], j['rgb']))} bgra identical {bool(np.array_equal(g['bgra'], j['bgra']))} | "
          f"generic {g_lo:.4f} ms (median {g_med:.4f}, {flops(g, 223, 123) / (g_lo * 1e-3) / 157.3e12:.4f} of peak)  specialised {j_lo:.4f} ms (median {j_med:.4f}, {flops(j, 223, 123) / (j_lo * 1e-3) / 157.3e12:.4f})", flush=True)
for w, h, ss in ((784, 561, 1), (784, 561, 16), (3840, 2160, 16), (3840, 2160, 256)):
    s, cam, prm = scenes.scene02(ss, seed=1)
    ctx.set_debug_flags(16)
    os.environ["RT_MI355_JIT_NO_STRICT"] = "1"; a, a_lo, a_med = timed(lambda: ctx.render_whitted(s, cam, prm, w, h)); ka = ctx.last_whitted_kernel()
    os.environ.pop("RT_MI355_JIT_NO_STRICT"); b, b_lo, b_med = timed(lambda: ctx.render_whitted(s, cam, prm, w, h)); kb = ctx.last_whitted_kernel()
    ctx.set_debug_flags(0)
    print(f"02 {w}x{h} x{ss}: kernels {ka}/{kb} rays equal {(a['rays'], a['shadow_rays']) == (b['rays'], b['shadow_rays'])} rgb bit-identical {bool(np.array_equal(a['rgb'], b['rgb']))} | "
          f"shadow query only {a_lo:.4f} ms (median {a_med:.4f}, {flops(a, 189, 134) / (a_lo * 1e-3) / 157.3e12:.4f} of peak)  + strict queries {b_lo:.4f} ms (median {b_med:.4f}, {flops(b, 189, 134) / (b_lo * 1e-3) / 157.3e12:.4f})", flush=True)
