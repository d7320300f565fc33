"""The allocation-free entry point (rt_path_accumulate_device_noalloc) against rt_path_accumulate_device, through the C ABI.

Both run the same generic kernels on the same counter-RNG, so wherever they cut the samples into the same number of slices the
accumulators must be IDENTICAL, bit for bit, and so must the ray counts; where the fixed scratch forces fewer slices only the
order of the float additions differs (1e-5 relative, ray counts still identical).  Two calls in a row, the second smaller
than the first: the scratch keeps the first call's contents and nothing of them may leak into the second result."""
import ctypes as C

import numpy as np
import pytest

from rtamd import abi, lib, scenes

pytestmark = pytest.mark.gpu


def _both(ctx, s, cam, prm, w, h):
    import torch
    ref = torch.full((h * w * 4,), -7.0, dtype=torch.float32, device="cuda")
    got = torch.full((h * w * 4,), -9.0, dtype=torch.float32, device="cuda")
    ctx.set_debug_flags(8)                       # the existing entry point on its generic kernel too (no scene-specialised build)
    try:
        ctx.path_accumulate_device(s, cam, prm, w, h, ref.data_ptr())
        st_ref = ctx.sync()
    finally:
        ctx.set_debug_flags(0)
    ctx.path_accumulate_device_noalloc(s, cam, prm, w, h, got.data_ptr())
    st_got = ctx.sync()
    return ref.cpu().numpy().reshape(h, w, 4), got.cpu().numpy().reshape(h, w, 4), st_ref, st_got


def test_noalloc_matches_the_allocating_entry_point_twice_in_a_row(ctx):
    L = lib.load()
    sb, pb = C.c_int32(), C.c_int32()
    assert L.rt_noalloc_limits(C.byref(sb), C.byref(pb)) == 0 and sb.value == 64 * 1024 and pb.value == 32 << 20
    # 1st call: 04 scene, 512x384, 2 samples -> both entry points cut the samples into 2 slices (2 x 3 MiB of partial sums)
    s, cam, _ = scenes.scene04(2, seed=5)
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, 2, seed=5)
    ref, got, a, b = _both(ctx, s, cam, prm, 512, 384)
    assert a["rays"] == b["rays"] and a["samples"] == b["samples"] == 512 * 384 * 2
    assert np.array_equal(ref, got)
    assert (got[:, :, 3] == 2).all() and got[:, :, :3].max() > 0
    # 2nd call, smaller input, another scene and variant: 03 scene, 200x120, 3 samples -> 3 slices in both
    s2, cam2, _ = scenes.scene03(3, seed=6)
    prm2 = scenes.path_params(abi.RT_PATH_SKY, 3, seed=6)
    ref2, got2, a2, b2 = _both(ctx, s2, cam2, prm2, 200, 120)
    assert a2["rays"] == b2["rays"] and a2["samples"] == b2["samples"] == 200 * 120 * 3
    assert np.array_equal(ref2, got2)
    assert (got2[:, :, 3] == 3).all()


def test_noalloc_limits_and_capped_slices(ctx):
    import torch
    # more slices than the fixed scratch holds: 640x400 at 16 spp wants 9 slices (36.9 MB); the scratch holds 8 -> same paths
    # (identical ray counts, identical sample counts), sums equal up to the order of the float additions
    s, cam, _ = scenes.scene04(16, seed=2)
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, 16, seed=2)
    ref, got, a, b = _both(ctx, s, cam, prm, 640, 400)
    assert a["rays"] == b["rays"] and (got[:, :, 3] == 16).all()
    assert np.allclose(got[:, :, :3], ref[:, :, :3], rtol=1e-5, atol=1e-6)
    # a sample range, and an image whose single slice alone exceeds the scratch (written straight into d_accum)
    prm_r = scenes.path_params(abi.RT_PATH_LIGHTS, 16, seed=2, sample_begin=3, sample_end=4)
    ref, got, a, b = _both(ctx, s, cam, prm_r, 2048, 1100)
    assert a["rays"] == b["rays"] and np.array_equal(ref, got)
    # a scene beyond the fixed scene scratch is refused with a clear error, and the context keeps working
    big, bcam, bprm = scenes.scene_stress(n_spheres=1200, grid=2, spp=1, seed=1)
    acc = torch.zeros(64 * 48 * 4, dtype=torch.float32, device="cuda")
    with pytest.raises(lib.RtError) as e:
        ctx.path_accumulate_device_noalloc(big, bcam, bprm, 64, 48, acc.data_ptr())
    assert e.value.code == abi.RT_ERR_TOO_LARGE and "scratch" in str(e.value)
    ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
    try:
        with pytest.raises(lib.RtError) as e:
            ctx.path_accumulate_device_noalloc(s, cam, prm, 64, 48, acc.data_ptr())
        assert e.value.code == abi.RT_ERR_UNSUPPORTED
    finally:
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
    ctx.path_accumulate_device_noalloc(s, cam, prm, 64, 48, acc.data_ptr())
    assert ctx.sync()["samples"] == 64 * 48 * 16
