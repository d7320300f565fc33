"""Whole-call time vs kernel time of the host-buffer entry points (what a drop-in Render pays beyond the kernel): one-shot
frames with a BGRA buffer only (what the C++ Render passes), and progressive passes with a preview per pass."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import ctypes as C
import numpy as np
from rtamd import abi, lib, scenes
ctx = lib.Context()
L = ctx.L
def bgra_only(fn, sc, cam, prm, w, h, n=5):
    buf = np.zeros((h, w, 4), np.uint8)
    img = abi.rt_image(buf.ctypes.data, None, w, h, 0)
    scc = sc.c()
    best = None
    for _ in range(n):
        t0 = time.perf_counter()
        ctx._chk(fn(ctx.h, C.byref(scc), C.byref(cam), C.byref(prm), C.byref(img)))
        dt = (time.perf_counter() - t0) * 1e3
        st = ctx.stats()
        if best is None or dt < best[0]: best = (dt, st["kernel_ms"])
    return best
for name, (sc, cam, prm), fn in (("01", scenes.scene01(), L.rt_render_whitted), ("02/16", scenes.scene02(16, seed=1), L.rt_render_whitted),
                                 ("03/16spp", scenes.scene03(16, seed=1), L.rt_render_path), ("04/32spp", scenes.scene04(32, seed=1), L.rt_render_path)):
    for w, h in ((784, 561), (1920, 1080), (3840, 2160)):
        call, k = bgra_only(fn, sc, cam, prm, w, h)
        print(f"{name:9s} {w}x{h}: call {call:8.3f} ms, kernel {k:8.3f} ms, beyond the kernel {call - k:7.3f} ms ({w * h * 4 / 1e6:.1f} MB frame)", flush=True)
for w, h in ((784, 561), (1920, 1080), (3840, 2160)):
    sc, cam, prm = scenes.scene04(64, seed=1)
    ctx.progressive_begin(sc, cam, prm, w, h)
    for chunk in (1, 1, 1, 4, 4):
        t0 = time.perf_counter(); r = ctx.progressive_pass(chunk, want_rgb=False); dt = (time.perf_counter() - t0) * 1e3
        print(f"progressive 04 {w}x{h} +{chunk} spp with preview: call {dt:7.3f} ms, kernel {r['kernel_ms']:7.3f} ms", flush=True)
    t0 = time.perf_counter(); r = ctx.progressive_pass(4, preview=False); dt = (time.perf_counter() - t0) * 1e3
    print(f"progressive 04 {w}x{h} +4 spp, no preview:   call {dt:7.3f} ms, kernel {r['kernel_ms']:7.3f} ms", flush=True)
    ctx.progressive_end()
