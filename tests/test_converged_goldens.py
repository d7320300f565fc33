"""Converged-image parity against goldens made by the REFERENCE's own code (tests/golden/converged_0{2,3,3r,4}.npz, written by
oracle/gen_converged.py from oracle/_ref/libref_e{2,3,4}.so: the reference's List/Scene::intersectsRay, materials, samplers
and <random> engines, driven by a loop that follows its Render/TraceTay).  SURVEY.md 8(c) bound, in linear pre-gamma colour:

    RMSE(image, golden) <= 1.1 * sqrt(mean(var) * (1/N + 1/N_golden))     var = the golden's per-pixel sample variance
    per-channel image mean within 1 % of the golden's                     (the bias check: a "fixed" sampler fails it)

PSNR of the gamma-encoded 8-bit images is printed, not asserted (the 04 estimator has no next-event estimation and is noisy).
CPU tests pin the oracle (rt_oracle.c); the gpu tests pin the HIP kernels, through the C ABI."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from rtamd import abi, scenes
import orc as oracle

W, H = 196, 140


def _golden(which):
    g = np.load(os.path.join(GOLDEN, f"converged_{which}.npz"))
    return g["mean"].astype(np.float64), g["var"].astype(np.float64), int(g["spp"]), float(g["rays_per_sample"])


def _psnr8(a, b, gamma):
    enc = lambda x: np.floor(np.clip(np.sqrt(np.maximum(x, 0)) if gamma else x, 0, 1) * 255.0)
    mse = ((enc(a) - enc(b)) ** 2).mean()
    return 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))


def _check(name, img, n, which, rays_per_sample=None, gamma=True, mean_rtol=0.01):
    mean, var, n_gold, rps = _golden(which)
    img = np.asarray(img, np.float64)
    bound = 1.1 * np.sqrt(var.mean() * (1.0 / n + 1.0 / n_gold))
    rmse = np.sqrt(((img - mean) ** 2).mean())
    ratio = img.mean(axis=(0, 1)) / mean.mean(axis=(0, 1))
    print(f"{name} vs reference golden {which} ({n_gold} spp): N {n}, RMSE {rmse:.5f} <= bound {bound:.5f}; channel means x{ratio}; "
          f"PSNR(8-bit) {_psnr8(img, mean, gamma):.1f} dB")
    assert rmse <= bound, (rmse, bound)
    assert np.allclose(ratio, 1.0, rtol=mean_rtol, atol=0), ratio
    if rays_per_sample is not None:
        # scene queries per sample: a path statistic no colour bound sees.  At the goldens' own sample counts the noise of the
        # ratio is ~1e-4; 5e-4 there caught the hardware square root's bias on 03's ground sphere (+1.1e-3, rt_device.h)
        tol = 5e-4 if n >= 8192 else 0.004
        assert abs(rays_per_sample - rps) <= tol * rps, (rays_per_sample, rps, rays_per_sample / rps - 1.0)


def _scene(which, spp, seed):
    if which == "04":
        return scenes.scene04(spp, seed=seed)
    return scenes.scene03(spp, seed=seed, readme_variant=(which == "03r"))


# ---- the oracle against the reference (CPU) ------------------------------------------------------------------------------
@pytest.mark.parametrize("which,n", [("04", 512), ("03", 256), ("03r", 256)])
def test_oracle_matches_reference_converged_image(which, n):
    s, cam, prm = _scene(which, n, seed=31)
    r = oracle.render_path(s, cam, prm, W, H, threads=8)
    _check("oracle", r["sum"] / n, n, which, r["rays"] / (W * H * n))


def test_oracle_matches_reference_mean_image_02():
    """02: the mean of independent evaluations (different seeds) of the deterministic-at-a-seed frame."""
    n = 32
    acc = np.zeros((H, W, 3), np.float64)
    for k in range(n):
        s, cam, prm = scenes.scene02(16, seed=1000 + k)
        r = oracle.render_whitted(s, cam, prm, W, H, threads=8)
        assert r["rays"] == W * H and r["shadow_rays"] == 16 * W * H
        acc += r["rgb"]
    _check("oracle", acc / n, n, "02", 17.0, gamma=False)


# ---- the HIP kernels against the reference (GPU, through the C ABI) ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("which,n", [("04", 16384), ("03", 16384), ("03r", 8192)])
def test_gpu_matches_reference_converged_image(ctx, which, n):
    s, cam, prm = _scene(which, n, seed=47)
    g = ctx.render_path(s, cam, prm, W, H)
    _check("HIP", g["rgb"], n, which, g["rays"] / (W * H * n))
    # both kernels a small scene can run: the generic one and the scene-specialised one
    ctx.set_debug_flags(16)
    try:
        j = ctx.render_path(s, cam, prm, W, H)
        assert ctx.last_path_kernel()[0] == 2
    finally:
        ctx.set_debug_flags(0)
    _check("HIP (scene-specialised)", j["rgb"], n, which, j["rays"] / (W * H * n))


@pytest.mark.gpu
def test_gpu_matches_reference_mean_image_02(ctx):
    n = 128
    acc = np.zeros((H, W, 3), np.float64)
    for k in range(n):
        s, cam, prm = scenes.scene02(16, seed=5000 + k)
        g = ctx.render_whitted(s, cam, prm, W, H)
        acc += g["rgb"]
    _check("HIP", acc / n, n, "02", 17.0, gamma=False)


# ---- BASELINE config 2: the shadow-sample sweep against the reference (tests/golden/converged_02_sweep.npz, 98x70) ----------
def _check_sweep(name, render, shadow, n):
    """mean of n evaluations (different seeds) of the frame with 'shadow' samples per light vs the reference's 256 evaluations"""
    g = np.load(os.path.join(GOLDEN, "converged_02_sweep.npz"))
    w, h, n_gold = int(g["width"]), int(g["height"]), int(g["evals"])
    mean, var = g[f"mean_s{shadow}"].astype(np.float64), g[f"var_s{shadow}"].astype(np.float64)
    acc = np.zeros((h, w, 3), np.float64)
    for k in range(n):
        r = render(scenes.scene02(shadow, seed=7000 + 131 * shadow + k), w, h)
        assert r["rays"] == w * h and r["shadow_rays"] == shadow * w * h
        acc += r["rgb"]
    img = acc / n
    # sampling noise of both means + the 1e-4 float tolerance of the frame's deterministic part (tests/test_gpu_parity.py)
    # 1.25 instead of SURVEY 8(c)'s 1.1: only the penumbra's few hundred pixels of this 98x70 frame carry variance, so the RMSE
    # itself scatters by ~ +-15 % from sample count to sample count (measured: 0.87 ... 1.15 of its expectation)
    bound = 1.25 * np.sqrt((var * (1.0 / n + 1.0 / n_gold) + (1e-4 * mean) ** 2).mean())
    rmse = np.sqrt(((img - mean) ** 2).mean())
    ratio = img.mean(axis=(0, 1)) / mean.mean(axis=(0, 1))
    print(f"{name} vs reference, 02 with {shadow} shadow samples: RMSE {rmse:.5f} <= {bound:.5f}; channel means x{ratio}")
    assert rmse <= bound, (shadow, rmse, bound)
    assert np.allclose(ratio, 1.0, rtol=2e-3, atol=0), ratio


@pytest.mark.parametrize("shadow", [1, 8, 64])
def test_oracle_matches_reference_over_the_shadow_sample_sweep(shadow):
    _check_sweep("oracle", lambda scp, w, h: oracle.render_whitted(*scp, w, h, threads=8), shadow, 32 if shadow < 64 else 8)


@pytest.mark.gpu
@pytest.mark.parametrize("shadow", [1, 2, 4, 8, 32, 64, 128, 256])
def test_gpu_matches_reference_over_the_shadow_sample_sweep(ctx, shadow):
    _check_sweep("HIP", lambda scp, w, h: ctx.render_whitted(*scp, w, h), shadow, 256)
