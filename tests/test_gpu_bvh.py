"""GPU tests of the optional BVH scene query (rt_set_option(RT_OPT_ACCEL, RT_ACCEL_BVH)).

The reference has no acceleration structure: its scene query is the brute-force loop (03 Utils.h:255-271, 04
Utils.h:362-390).  The BVH mode therefore has one bar: the SAME image as the brute-force kernels, bit for bit and ray for
ray, on scenes that exercise what makes that hard - exact ties (the last element in scene order wins), the sphere test's
"hit at tMax" rule for rays that start inside overlapping spheres (Common/Ray.hpp:127-134), rectangles and box faces at
arbitrary orientations, triangles, planes (unbounded: outside the tree), empty scenes."""
import numpy as np
import pytest

from conftest import nest_scene
from rtamd import abi, scenes
import orc as oracle

pytestmark = pytest.mark.gpu


def _both(ctx, s, cam, prm, w, h):
    ctx.set_debug_flags(8)              # brute force through the generic kernels (no scene-specialised build)
    try:
        a = ctx.render_path(s, cam, prm, w, h)
        assert ctx.last_path_kernel()[0] in (0, 1)
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
        try:
            b = ctx.render_path(s, cam, prm, w, h)
            assert ctx.last_path_kernel()[0] == 3
        finally:
            ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
    finally:
        ctx.set_debug_flags(0)
    return a, b


def _same(a, b):
    assert a["rays"] == b["rays"] and a["samples"] == b["samples"]
    assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["bgra"], b["bgra"])


def test_bvh_matches_brute_force_on_the_reduced_stress_scene(ctx):
    """600 overlapping spheres (a quarter of them refractive: rays start inside spheres that intersect other spheres),
    288 triangles, a ground plane, an emissive rectangle; 03 and 04 estimator."""
    s, cam, prm = scenes.scene_stress(n_spheres=600, grid=12, spp=8, seed=3)
    a, b = _both(ctx, s, cam, prm, 160, 90)
    _same(a, b)
    assert a["rgb"].mean() > 0.01
    prm3 = scenes.path_params(abi.RT_PATH_SKY, 4, max_depth=12, seed=5)
    a, b = _both(ctx, s, cam, prm3, 96, 54)
    _same(a, b)
    o = oracle.render_path(s, cam, prm3, 96, 54, threads=8)          # and both agree with the oracle as closely as the brute-force kernel does
    assert abs(b["rays"] - o["rays"]) <= 2e-2 * o["rays"]           # (03 paths are long: one differing bounce changes many rays)


def test_bvh_ties_and_rays_starting_inside_overlapping_spheres(ctx):
    """Coincident duplicates (exact ties: the LAST one in scene order must win - they differ in material), glass spheres
    nested and intersecting so that rays inside one sphere find another surface before its far wall (the tMax rule hands
    the hit to the enclosing sphere when it comes later in scene order), rotated boxes, a rectangle coplanar with a box face."""
    s, cam = nest_scene()
    for variant, depth in ((abi.RT_PATH_LIGHTS, 8), (abi.RT_PATH_SKY, 16)):
        prm = scenes.path_params(variant, 16, max_depth=depth, seed=9)
        a, b = _both(ctx, s, cam, prm, 160, 120)
        _same(a, b)
    o = oracle.render_path(s, cam, prm, 160, 120, threads=8)                     # the brute-force semantics themselves are the oracle's
    assert abs(b["rays"] - o["rays"]) <= 2e-2 * o["rays"]


def test_bvh_small_and_degenerate_scenes(ctx):
    """The reference scenes (few primitives, mostly planes), a scene with no bounded primitive, an empty scene, one primitive."""
    for fn in (scenes.scene04, scenes.scene03):
        s, cam, prm = fn(8, seed=2)
        _same(*_both(ctx, s, cam, prm, 128, 96))
    s = scenes.Scene(); w = s.diffuse((0.5, 0.5, 0.5))
    cam = scenes.camera(60.0, (0, 1, 3), (0, 0, 0)); prm = scenes.path_params(abi.RT_PATH_SKY, 4, seed=1)
    _same(*_both(ctx, s, cam, prm, 40, 30))                                      # empty
    s.plane(w, (0, 0, 0), (0, 1, 0)); s.plane(w, (0, 0, -8), (0.1, 0.2, 1.0))
    _same(*_both(ctx, s, cam, prm, 40, 30))                                      # planes only
    s.sphere(w, (0, 1, -3), 1.0)
    _same(*_both(ctx, s, cam, prm, 40, 30))                                      # a single leaf
    for (wd, ht) in [(1, 1), (17, 33)]:
        _same(*_both(ctx, s, cam, prm, wd, ht))


def test_bvh_full_stress_scene_is_far_faster_than_brute_force(ctx):
    """SURVEY.md 8(d) config 5 geometry at full count (10,000 spheres, 8,192 triangles): the walk must agree with the
    brute-force kernel ray for ray and be at least 8x faster (measured: 18x at 1080p)."""
    s, cam, prm = scenes.scene_stress(spp=2, seed=1)
    a, b = _both(ctx, s, cam, prm, 320, 180)
    _same(a, b)
    assert b["kernel_ms"] * 8 < a["kernel_ms"], (a["kernel_ms"], b["kernel_ms"])


def test_bvh_scheduled_walk_matches_the_function_form(ctx):
    """k_path_bvh schedules stages over a per-lane state machine (rt_bvh_device.h); debug flag 32 runs the plain form (one
    walk per lane as a function).  Per lane the operations are the same, so the frames are bit-identical."""
    s, cam, prm = scenes.scene_stress(n_spheres=900, grid=16, spp=6, seed=13)
    ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
    try:
        a = ctx.render_path(s, cam, prm, 200, 120)
        ctx.set_debug_flags(32)
        b = ctx.render_path(s, cam, prm, 200, 120)
    finally:
        ctx.set_debug_flags(0)
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
    _same(a, b)
    assert a["rays"] > 200 * 120 * 6


def _random_scene(rng, n):
    """n primitives of every bounded and unbounded kind at random places, sizes and orientations around the origin, every
    material kind, one light; sizes span three orders of magnitude so that padded bounds, early-outs and boxes all get thin,
    huge, nested and overlapping cases."""
    s = scenes.Scene()
    mats = [s.diffuse(tuple(rng.uniform(0.2, 0.9, 3))) for _ in range(3)] + [s.metal(tuple(rng.uniform(0.5, 0.9, 3)), float(rng.uniform(0, 0.5))),
            s.dielectric(float(rng.uniform(0.5, 0.9))), s.light_material((6.0, 6.0, 6.0))]
    s.plane(mats[0], (0, -6, 0), (0, 1, 0))
    for i in range(n):
        kind = rng.integers(0, 5)
        m = int(mats[rng.integers(0, len(mats))]) if rng.random() > 0.05 else -1
        c = tuple(rng.uniform(-8, 8, 3) + np.array([0, 0, -12]))
        size = float(10 ** rng.uniform(-1.5, 0.8))
        if kind == 0:
            s.sphere(m, c, size, flip=bool(rng.random() < 0.1))
        elif kind == 1:
            s.rect(m, c, (size, size * float(rng.uniform(0.3, 3))), tuple(rng.uniform(-180, 180, 3)))
        elif kind == 2:
            s.box(m, c, tuple(size * rng.uniform(0.3, 2, 3)), tuple(rng.uniform(-90, 90, 3)), flip=bool(rng.random() < 0.1))
        elif kind == 3:
            v = [tuple(np.array(c) + rng.uniform(-size, size, 3)) for _ in range(3)]
            s.triangle(m, *v)
        else:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            if rng.random() < 0.5:
                nrm = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1.0, 1.0])
            s.plane(m, tuple(np.array(c) + nrm * -14.0), tuple(nrm))          # far behind the cloud, facing it
    s.rect(mats[-1], (0, 9, -12), (12, 12), (90, 0, 0))
    return s


@pytest.mark.parametrize("seed,n", [(1, 12), (2, 40), (3, 70), (4, 150), (5, 400), (6, 900), (7, 25), (8, 58), (9, 66), (10, 100), (11, 250), (12, 600), (13, 1500), (14, 5)])
def test_random_scenes_every_kernel_agrees(ctx, seed, n):
    """Differential test over random scenes: the generic kernel (LDS-resident, with early-outs above 64 slots, or tiled), the
    plain global-memory kernel, the scene-specialised build (up to 64 slots) and both BVH kernels must produce the same frame
    bit for bit - they differ only in how they walk the scene."""
    rng = np.random.default_rng(1000 + seed)
    s = _random_scene(rng, n)
    cam = scenes.camera(70.0, (0.3, 0.5, 6.0), (-3.0, 2.0, 1.0))
    for variant, depth in ((abi.RT_PATH_LIGHTS, 6), (abi.RT_PATH_SKY, 10)):
        prm = scenes.path_params(variant, 4, max_depth=depth, seed=seed)
        frames = {}
        for name, flags, accel in (("generic", 8, 0), ("global", 2, 0), ("specialised", 16, 0), ("bvh", 8, 1), ("bvh-simple", 8 | 32, 1)):
            ctx.set_debug_flags(flags)
            ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH if accel else abi.RT_ACCEL_NONE)
            try:
                frames[name] = ctx.render_path(s, cam, prm, 96, 72)
                kind = ctx.last_path_kernel()[0]
            finally:
                ctx.set_debug_flags(0); ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
            if name == "specialised":
                assert kind == (2 if len(s.prims) + 5 * sum(p.type == abi.RT_PRIM_BOX for p in s.prims) <= 64 else kind)
        ref = frames["global"]
        assert ref["rays"] >= 96 * 72 * 4 and ref["samples"] == 96 * 72 * 4
        for name, f in frames.items():
            assert f["rays"] == ref["rays"], name
            assert np.array_equal(f["rgb"], ref["rgb"]), name


@pytest.mark.parametrize("seed,n", [(21, 30), (22, 120), (23, 500)])
def test_random_scenes_agree_with_the_oracle(ctx, seed, n):
    """The same random scenes against the CPU oracle (same seeds, same counter-based draws): equal ray counts to 1 %, the
    same paths on nearly every pixel (a path that grazes an edge differently diverges completely: chaotic, not a bias)."""
    rng = np.random.default_rng(1000 + seed)
    s = _random_scene(rng, n)
    cam = scenes.camera(70.0, (0.3, 0.5, 6.0), (-3.0, 2.0, 1.0))
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, 4, max_depth=6, seed=seed)
    g = ctx.render_path(s, cam, prm, 96, 72)
    o = oracle.render_path(s, cam, prm, 96, 72, threads=8)
    om = o["sum"] / 4
    assert abs(g["rays"] - o["rays"]) <= 1e-2 * o["rays"]
    assert np.mean(np.all(np.abs(g["rgb"] - om) <= 1e-3 * np.maximum(np.abs(om), 1e-3), axis=-1)) >= 0.95
    assert np.allclose(g["rgb"].mean(axis=(0, 1)), om.mean(axis=(0, 1)), rtol=5e-2, atol=1e-3)
