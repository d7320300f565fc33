"""The item form of the bounce loop (path_items_body: persistent waves pulling units of work from a counter, lanes decoupled
from pixels, 32.32 fixed-point per-pixel sums in LDS) against the one-lane-per-pixel form (path_body) and the oracle.
Same counter RNG, same camera / query / shading code => the SAME PATHS: scene-query counts are equal exactly, per pixel too;
colours differ only by the rounding of the per-pixel sum (exact fixed point rounded once vs a running float): <= 1e-6 relative.
Debug flags: 1024 = item form whatever the frame size (by default only frames that can occupy the chip take it), 512 = one lane
per pixel, 4 = the accumulator's w channel holds the pixel's scene-query count."""
import os

import numpy as np
import pytest

from conftest import frac_within, nest_scene
from rtamd import abi, lib, scenes
import orc as oracle

pytestmark = pytest.mark.gpu


def _render(ctx, flags, s, cam, prm, w, h):
    ctx.set_debug_flags(flags)
    try:
        r = ctx.render_path(s, cam, prm, w, h)
        r["form"] = ctx.last_path_form(); r["kernel"] = ctx.last_path_kernel()[0]
    finally:
        ctx.set_debug_flags(0)
    return r


def _px_counts(ctx, flags, s, cam, prm, w, h):
    import torch
    acc = torch.zeros(h, w, 4, device="cuda")
    ctx.set_debug_flags(flags | 4)
    try:
        ctx.path_accumulate_device(s, cam, prm, w, h, acc.data_ptr())
        st = ctx.sync()
    finally:
        ctx.set_debug_flags(0)
    return acc[:, :, 3].cpu().numpy().astype(np.int64), st["rays"]


def _cases():
    ns, ncam = nest_scene()
    return {
        "04": (scenes.scene04(16), (196, 140)),
        "03": (scenes.scene03(16), (196, 140)),
        "03 readme": (scenes.scene03(8, readme_variant=True), (98, 70)),
        "04 odd size": (scenes.scene04(5), (61, 37)),
        "03 one sample": (scenes.scene03(1), (98, 70)),
        "nest": ((ns, ncam, scenes.path_params(abi.RT_PATH_LIGHTS, 12, seed=5)), (98, 70)),
        "mid-size (150 slots)": (scenes.scene_stress(n_spheres=140, grid=2, spp=4, seed=3), (96, 54)),
    }


@pytest.mark.parametrize("name", list(_cases()))
@pytest.mark.parametrize("kernel", ["generic", "specialised", "global"])
def test_item_form_traces_the_same_paths(ctx, name, kernel):
    (s, cam, prm), (w, h) = _cases()[name]
    extra = {"generic": 8, "specialised": 16, "global": 2}[kernel]
    a = _render(ctx, extra | 1024, s, cam, prm, w, h)
    b = _render(ctx, extra | 512, s, cam, prm, w, h)
    assert a["form"][0] == 2 and b["form"][0] == 0, (a["form"], b["form"])
    assert a["kernel"] == b["kernel"]
    assert a["rays"] == b["rays"] and a["samples"] == b["samples"]
    assert frac_within(a["rgb"], b["rgb"], 1e-6, floor=1e-2) == 1.0
    assert (a["bgra"] != b["bgra"]).mean() < 1e-4
    ca, ra = _px_counts(ctx, extra | 1024, s, cam, prm, w, h)
    cb, rb = _px_counts(ctx, extra | 512, s, cam, prm, w, h)
    assert ra == rb == ca.sum() and np.array_equal(ca, cb)        # the same scene queries, pixel by pixel


def test_item_form_is_independent_of_unit_size_and_reproducible(ctx):
    """Fixed-point sums: integer addition is order-independent, so the frame is the same BIT FOR BIT whatever the unit size
    (1 ... 64 pixels per unit: which lanes trace which paths changes completely), with sample slices, and from run to run."""
    s, cam, prm = scenes.scene03(24, seed=4)
    w, h = 131, 77
    frames = []
    try:
        for k in ("6", "3", "0", "5"):
            os.environ["RT_MI355_UNIT_LOG2"] = k
            r = _render(ctx, 16 | 1024, s, cam, prm, w, h)
            assert r["form"] == (2, int(k))
            frames.append(r)
        os.environ["RT_MI355_SLICES"] = "3"
        frames.append(_render(ctx, 16 | 1024, s, cam, prm, w, h))
    finally:
        os.environ.pop("RT_MI355_UNIT_LOG2", None); os.environ.pop("RT_MI355_SLICES", None)
    frames.append(_render(ctx, 16 | 1024, s, cam, prm, w, h))
    for r in frames[1:4] + frames[5:]:
        assert r["rays"] == frames[0]["rays"] and np.array_equal(r["rgb"], frames[0]["rgb"])
    # slices: three partial sums added as floats - equal to rounding
    assert frames[4]["rays"] == frames[0]["rays"] and frac_within(frames[4]["rgb"], frames[0]["rgb"], 1e-6, floor=1e-2) == 1.0


def test_item_form_against_the_oracle(ctx):
    """HIP item form vs oracle at equal seeds (the bounds of tests/test_gpu_parity.py for the one-lane-per-pixel form)."""
    for (s, cam, prm), need in ((scenes.scene04(8), 0.995), (scenes.scene03(8), 0.97)):
        w, h = 196, 140
        g = _render(ctx, 1024, s, cam, prm, w, h)
        assert g["form"][0] == 2
        o = oracle.render_path(s, cam, prm, w, h, threads=16)
        assert abs(g["rays"] - o["rays"]) <= 2e-3 * o["rays"]
        assert frac_within(g["rgb"], o["sum"] / prm.spp, 1e-3) >= need


def test_large_frames_take_the_item_form_small_ones_do_not(ctx):
    """The default policy (plan_items, rt_api.hip): a frame takes the item form when its units can be long (>= 2048 paths, 32 per lane)
    and numerous (>= the resident waves); the frame the bench times does, an example-size frame keeps one lane per pixel."""
    import torch
    s, cam, prm = scenes.scene03(64)
    acc = torch.zeros(1080, 1920, 4, device="cuda")
    ctx.path_accumulate_device(s, cam, prm, 1920, 1080, acc.data_ptr()); ctx.sync()
    form, unit_log2 = ctx.last_path_form()
    assert form == 2 and 2048 <= (64 << unit_log2) <= 4096                          # units of 2048..4096 paths: 32 or 64 pixels at 64 spp
    assert float(acc[:, :, 3].min()) == 64.0 == float(acc[:, :, 3].max())          # every pixel owns all its samples
    s, cam, prm = scenes.scene04(32)
    r = ctx.render_path(s, cam, prm, 784, 561)
    assert ctx.last_path_form()[0] == 0
    s, cam, prm = scenes.scene04(1024)
    small = torch.zeros(64, 64, 4, device="cuda")
    ctx.path_accumulate_device(s, cam, prm, 64, 64, small.data_ptr()); ctx.sync()
    assert ctx.last_path_form()[0] == 0                                              # 4096 pixels cannot occupy the chip as units


def test_colours_outside_the_fixed_point_range_keep_float_sums(ctx):
    """Unsigned 32.32 sums cannot hold a negative contribution or an albedo above 1: such scenes render through path_body."""
    s, cam, prm = scenes.scene04(4)
    s.materials[1].albedo[0] = 1.5; s._c = None
    r = _render(ctx, 1024, s, cam, prm, 98, 70)
    assert r["form"][0] == 0 and np.isfinite(r["rgb"]).all()


@pytest.mark.parametrize("name", ["04", "03", "03 readme", "04 odd size", "nest"])
@pytest.mark.parametrize("kernel", ["generic", "specialised"])
def test_outcome_sorted_form_renders_the_item_form_frame_bit_for_bit(ctx, name, kernel):
    """path_sorted_body (debug flag 2048 / RT_MI355_PATH=s: the workgroup sorts its paths by what the query found, settles ended
    paths before the exchange, shades scattering ones in full waves) traces the same paths and adds into the same fixed-point
    sums as the item form: identical ray counts, identical frames.  It lost its A/B on the GPU (profiles/r03_pmc_forms.txt)."""
    (s, cam, prm), (w, h) = _cases()[name]
    extra = {"generic": 8, "specialised": 16}[kernel]
    a = _render(ctx, extra | 1024 | 2048, s, cam, prm, w, h)
    b = _render(ctx, extra | 1024, s, cam, prm, w, h)
    assert a["form"][0] == 3 and b["form"][0] == 2
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["bgra"], b["bgra"])
    # a depth limit that cuts paths short: the sorted form settles "would be cut" before the exchange
    prm2 = scenes.path_params(prm.variant, prm.spp, max_depth=2, seed=7)
    a = _render(ctx, extra | 1024 | 2048, s, cam, prm2, w, h)
    b = _render(ctx, extra | 1024, s, cam, prm2, w, h)
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"])
