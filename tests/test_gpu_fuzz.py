"""A slice of the random-scene fuzz (tools/fuzz_path.py, tools/fuzz_whitted.py; profiles/r03_path_fuzz.txt, r03_whitted_fuzz.txt) kept in the suite: a few seeds of every
family through the generic kernel, the scene-specialised kernel, the item form and the BVH walk against the CPU oracle at equal seeds.  The bars are the ones the full
runs (~1,300 scenes) met after round 3's fixes; a scene outside them means an arithmetic path has left the reference's."""
import os
import sys

import numpy as np
import pytest
import torch

from rtamd import abi, scenes
import orc as oracle

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import fuzz_scenes  # noqa: E402

pytestmark = pytest.mark.gpu
W, H = 96, 64


@pytest.mark.parametrize("family,seeds,spp,bar", [("a", range(1, 13), 64, 1.5e-3), ("c", range(1, 9), 64, 2e-3), ("d", range(1, 5), 16, 4e-3)])
def test_path_fuzz_slice(ctx, family, seeds, spp, bar):
    for seed in seeds:
        s, cam, variant = fuzz_scenes.FAMILIES[family](seed)
        prm = scenes.path_params(variant, spp, seed=seed)
        o = oracle.render_path_px(s, cam, prm, W, H, threads=8)
        counts = {}
        for name, flags in (("generic", 8), ("specialised", 16)):
            acc = torch.zeros(H, W, 4, device="cuda")
            ctx.set_debug_flags(4 | flags)
            try:
                ctx.path_accumulate_device(s, cam, prm, W, H, acc.data_ptr()); st = ctx.sync()
            finally:
                ctx.set_debug_flags(0)
            counts[name] = st["rays"]
            a = acc.cpu().numpy()
            assert abs(st["rays"] / o["rays"] - 1.0) <= bar, (family, seed, name, st["rays"] / o["rays"] - 1.0)
            mo = o["sum"].reshape(-1, 3).mean(axis=0).astype(np.float64); mg = a[:, :, :3].reshape(-1, 3).mean(axis=0).astype(np.float64)
            lit = mo >= 0.01 * spp
            if lit.any(): assert np.abs(mg[lit] / mo[lit] - 1.0).max() <= 5e-3, (family, seed, name)
        assert counts["generic"] == counts["specialised"], (family, seed)
        ctx.set_debug_flags(8 | 1024)                                  # the item form: the same paths
        try: it = ctx.render_path(s, cam, prm, W, H)
        finally: ctx.set_debug_flags(0)
        assert it["rays"] == counts["generic"] and (ctx.last_path_form()[0] == 2 or family == "d"), (family, seed)      # scenes beyond LDS keep one lane per pixel
        ctx.set_debug_flags(8 | 512)                                   # the BVH walk: the brute-force image bit for bit
        try: bf = ctx.render_path(s, cam, prm, W, H)
        finally: ctx.set_debug_flags(0)
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
        try: bv = ctx.render_path(s, cam, prm, W, H)
        finally: ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
        assert bv["rays"] == bf["rays"] and np.array_equal(bv["rgb"], bf["rgb"]), (family, seed)


def test_whitted_fuzz_slice(ctx):
    """Random Whitted scenes (chains and branching trees, 1-3 lights, hard and soft shadows, a fractional shininess among the materials): ray and shadow-ray counts equal to
    the oracle's, the monolithic kernels (generic and specialised) identical to each other, and their floats the oracle's but for a last bit of the specular power
    (<= 0.5 % of a scene's pixels when a fractional exponent is about, a handful otherwise) and, with soft shadows, a few pixels where one fast-arithmetic shadow sample falls
    the other way at a silhouette."""
    w, h = 128, 96
    for seed in range(1, 13):
        s, cam, depth, samples = fuzz_scenes.random_whitted_scene(seed)
        prm = scenes.whitted_params(depth, samples, seed=seed)
        o = oracle.render_whitted(s, cam, prm, w, h, threads=8)
        frames = []
        for flags in (8 | 4096, 16):
            ctx.set_debug_flags(flags)
            try: g = ctx.render_whitted(s, cam, prm, w, h)
            finally: ctx.set_debug_flags(0)
            assert (g["rays"], g["shadow_rays"]) == (o["rays"], o["shadow_rays"]), seed
            frames.append(g["rgb"])
        assert ctx.last_whitted_kernel() == 2 and np.array_equal(frames[0], frames[1]), seed
        d = np.abs(frames[0].astype(np.float64) - o["rgb"]).reshape(-1, 3).max(axis=1)
        fractional = any(m.shininess != int(m.shininess) for m in s.materials)
        assert (d > 0).sum() <= (0.05 if fractional else 0.002) * w * h, (seed, int((d > 0).sum()))
        assert (d > 4e-7).sum() <= (12 if samples > 0 else 0), (seed, int((d > 4e-7).sum()))
