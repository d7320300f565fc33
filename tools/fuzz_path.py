"""Random path-traced scenes (03 sky estimator / 04 lights estimator) on the GPU against the oracle at equal seeds.
python3 tools/fuzz_path.py [n_scenes] [first_seed]
Per scene: scene queries per sample (HIP, oracle, relative difference), share of pixels with the oracle's own query count, relative
difference of the channel means.  Exit status 1 when queries/sample differ by more than 2e-3 or a channel mean by more than 1e-2."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from rtamd import abi, lib, scenes
import orc as oracle

def random_scene(seed):
    g = np.random.default_rng(1000 + seed)
    U = lambda a, b: float(g.uniform(a, b))
    s = scenes.Scene()
    variant = abi.RT_PATH_SKY if seed % 2 else abi.RT_PATH_LIGHTS
    mats = [s.diffuse(tuple(g.uniform(0.1, 0.9, 3))) for _ in range(int(g.integers(1, 4)))]
    mats += [s.metal(tuple(g.uniform(0.5, 0.95, 3)), float(g.choice([0.0, 0.1, 0.4, 1.0]))) for _ in range(int(g.integers(0, 3)))]
    if g.random() < 0.7: mats.append(s.dielectric(float(g.choice([0.6, 0.66, 0.8, 1.5]))))
    pick = lambda: int(g.choice(mats))
    angles = lambda: (float(g.choice([0, 0, 90, U(-180, 180)])), float(g.choice([0, 0, 90, U(-180, 180)])), float(g.choice([0, 0, U(-180, 180)])))
    if variant == abi.RT_PATH_LIGHTS:     # a room (open towards the camera) or a floor, one or two lights
        lm = s.light_material(tuple(g.uniform(4, 20, 3)))
        s.plane(pick(), (0, -5, 0), (0, 1, 0))
        if g.random() < 0.7:
            s.plane(pick(), (0, 5, 0), (0, -1, 0)); s.plane(pick(), (0, 0, -5), (0, 0, 1)); s.plane(pick(), (-5, 0, 0), (1, 0, 0)); s.plane(pick(), (5, 0, 0), (-1, 0, 0))
        s.rect(lm, (U(-2, 2), 4.95, U(-2, 2)), (U(1, 4), U(1, 4)), (90, 0, 0))
        if g.random() < 0.4: s.sphere(lm, (U(-4, 4), U(-2, 3), U(-4, 2)), U(0.3, 1.0))
        cam = scenes.camera(U(60, 100), (U(-1, 1), U(-1, 1), 10), (U(-5, 5), U(-5, 5), 0))
        lo, hi = (-4, -4.5, -4), (4, 3, 4)
    else:                                 # objects on a large ground sphere or plane under the sky
        if g.random() < 0.5: s.sphere(pick(), (0, -100.5, -1), 100.0)
        else: s.plane(pick(), (0, -0.5, 0), (0, 1, 0))
        cam = scenes.camera(U(40, 90), (U(-2, 2), U(0.2, 2), U(1, 3)), (U(-30, 0), U(-40, 40), 0))
        lo, hi = (-2.5, -0.4, -4), (2.5, 1.5, 0)
    P = lambda: tuple(float(g.uniform(a, b)) for a, b in zip(lo, hi))
    scale = 1.0 if variant == abi.RT_PATH_LIGHTS else 0.3
    for _ in range(int(g.integers(2, 7))):
        kind = g.choice(["sphere", "sphere", "box", "rect", "tri", "shell"])
        if kind == "sphere": s.sphere(pick(), P(), U(0.5, 2.0) * scale)
        elif kind == "shell":
            c = P(); r = U(0.8, 2.0) * scale; gl = s.dielectric(0.625)
            s.sphere(gl, c, r); s.sphere(gl, c, r * 0.8, flip=True)
        elif kind == "box": s.box(pick(), P(), tuple(g.uniform(0.5, 3.0, 3) * scale), angles(), flip=bool(g.random() < 0.15))
        elif kind == "rect": s.rect(pick(), P(), (U(1, 4) * scale, U(1, 4) * scale), angles())
        else:
            a = np.array(P()); s.triangle(pick(), tuple(a), tuple(a + g.uniform(-3, 3, 3) * scale), tuple(a + g.uniform(-3, 3, 3) * scale))
    return s, cam, variant

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = lib.Context()
    if os.environ.get("FUZZ_ARITH") == "ref": ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_REFERENCE)      # the reference's arithmetic for rays and spheres
    W, H, SPP = 96, 64, 64
    bad = 0; worst_q = 0.0; worst_m = 0.0
    for seed in range(first, first + n):
        s, cam, variant = random_scene(seed)
        prm = scenes.path_params(variant, SPP, seed=seed)
        o = oracle.render_path_px(s, cam, prm, W, H, threads=16)
        line = f"seed {seed:3d} {'03' if variant == abi.RT_PATH_SKY else '04'} prims {len(s.prims):2d} oracle q/s {o['rays'] / (W * H * SPP):8.4f}:"
        for name, flags in (("generic", 8), ("generated", 16)):
            acc = torch.zeros(H, W, 4, device="cuda")
            ctx.set_debug_flags(4 | flags)
            try:
                ctx.path_accumulate_device(s, cam, prm, W, H, acc.data_ptr()); st = ctx.sync()
            finally: ctx.set_debug_flags(0)
            a = acc.cpu().numpy()
            relq = (st["rays"] - o["rays"]) / o["rays"]
            same = float((a[:, :, 3].astype(np.int64) == o["px_rays"]).mean())
            mo = o["sum"].reshape(-1, 3).mean(axis=0).astype(np.float64); mg = a[:, :, :3].reshape(-1, 3).mean(axis=0).astype(np.float64)
            relm = float(np.abs((mg - mo) / np.maximum(np.abs(mo), 1e-6)).max())
            line += f"  {name} k{ctx.last_path_kernel()[0]} rel q {relq:+.1e} px= {same:.4f} mean {relm:.1e}"
            worst_q = max(worst_q, abs(relq)); worst_m = max(worst_m, relm)
            if abs(relq) > 2e-3 or relm > 1e-2: bad += 1; line += " <<<"
        print(line, flush=True)
    print(f"largest |rel queries/sample| {worst_q:.2e}, largest rel channel-mean difference {worst_m:.2e}; FAILURES {bad}")
    return 1 if bad else 0
sys.exit(main())
