"""GPU tests of the cases where the last bits of the sphere test decide what a path does next: large spheres seen from
arbitrary cameras in the default (fast) arithmetic, and RT_OPT_ARITHMETIC = RT_ARITH_REFERENCE for small spheres far from the rays'
origins.  The oracle is the reference's arithmetic (oracle/rt_oracle.c, pinned by the reference's own headers); equal seeds throughout.
Found by tools/fuzz_path.py (random scenes); the figures quoted are profiles/r03_path_fuzz.txt's."""
import os
import sys

import numpy as np
import pytest

from rtamd import abi, scenes
import orc as oracle

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))

pytestmark = pytest.mark.gpu


def _rel_queries(ctx, s, cam, prm, w, h, flags):
    ctx.set_debug_flags(flags)
    try: g = ctx.render_path(s, cam, prm, w, h)
    finally: ctx.set_debug_flags(0)
    return g


@pytest.mark.parametrize("camera", [((-0.18, 1.51, 1.73), (-16.9, 29.2, 0.0), 88.8), ((1.28, 1.85, 3.0), (-13.9, 36.5, 0.0), 61.8), ((-0.89, 1.34, 2.52), (-15.3, -19.0, 0.0), 53.6)])
def test_ground_sphere_from_other_cameras_fast_arithmetic(ctx, camera):
    """03's scene (ground sphere r = 100, tMin = 0.001) seen from cameras that look further across the ground than the reference's.
    Leaving a sphere of that size, c = oc.oc - r*r cancels from 1e4 to ~1e-3 and decides whether the ray re-hits the ground above
    tMin; the reference does that in about one bounce of a hundred there.  With the coefficients fused (round 2) HIP traced up to
    1.3 % FEWER queries than the oracle from these cameras; PRIM_SPHERE_BIG slots now compute them in the reference's operation
    order (rt_device.h big_sphere_roots): within +-3e-4, generic and specialised kernel on the same paths."""
    pos, ori, fov = camera
    s, _, _ = scenes.scene03(64, seed=5)
    cam = scenes.camera(fov, pos, ori)
    prm = scenes.path_params(abi.RT_PATH_SKY, 64, seed=5)
    o = oracle.render_path(s, cam, prm, 192, 128, threads=8)
    a = _rel_queries(ctx, s, cam, prm, 192, 128, 8)
    b = _rel_queries(ctx, s, cam, prm, 192, 128, 16)
    assert ctx.last_path_kernel()[0] == 2
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"])
    assert abs(a["rays"] / o["rays"] - 1.0) <= 1e-3, a["rays"] / o["rays"] - 1.0
    mo = (o["sum"] / 64).reshape(-1, 3).mean(axis=0)
    assert np.abs(a["rgb"].reshape(-1, 3).mean(axis=0) / mo - 1.0).max() <= 5e-4


def test_rough_metal_ground_sphere_crossed_from_inside(ctx):
    """A ground sphere of rough metal (03 Materials.hpp: the cone around the mirror direction reaches below the surface) sends rays INTO
    the sphere; they come out at the far root ~2r.  With a = d.d taken as 1 that root is off by 2r(|d|^2 - 1) ~ 2e-5, the far hit
    point lies that much off the surface and the next bounce re-hits it too often: +3e-3 scene queries against the oracle, 18 % of the
    pixels with another ray count (fuzz seeds 61, 231).  big_sphere_roots solves with a: within 1e-3, generic = specialised."""
    s = scenes.Scene()
    s.sphere(s.metal((0.8, 0.7, 0.6), 1.0), (0, -100.5, -1.0), 100.0)
    s.sphere(s.diffuse((0.2, 0.3, 0.7)), (-0.6, 0.04, -3.0), 0.56); s.rect(s.diffuse((0.7, 0.3, 0.2)), (-1.36, 0.48, -0.38), (0.84, 0.76), (0, 0, 0))
    cam = scenes.camera(42.5, (-0.2, 0.9, 1.13), (-23.0, 2.8, 0.0))
    prm = scenes.path_params(abi.RT_PATH_SKY, 64, seed=61)
    o = oracle.render_path_px(s, cam, prm, 192, 128, threads=8)
    a = _rel_queries(ctx, s, cam, prm, 192, 128, 8)
    b = _rel_queries(ctx, s, cam, prm, 192, 128, 16)
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"])
    assert o["rays"] / (192 * 128 * 64) > 2.5                      # the paths really run on inside the sphere
    assert abs(a["rays"] / o["rays"] - 1.0) <= 1e-3, a["rays"] / o["rays"] - 1.0


def test_giant_glass_ground_sphere(ctx):
    """A glass ground of r = 2500 at tMin = 0.001 (fuzz family B, seed 33): r*r = 6e6 carries rounding noise of 0.5, i.e. distances of 1e-4/cos(theta) against tMin,
    so every refraction into and reflection inside the sphere is decided by last bits - in the reference too (3.4 queries per sample on a scene of 4 primitives).
    With directions that are unit only by construction HIP traced +37 % queries; scenes with a sphere beyond RT_GIANT_SPHERE_RATIO x tMin now normalise scattered
    directions in the reference's operation order (PathArgs::norm_dirs): within 1 %, generic and specialised kernel on the same paths."""
    s = scenes.Scene()
    glass = s.dielectric(0.66)
    s.sphere(glass, (0, -2525.0, -50.0), 2500.0)
    s.sphere(s.dielectric(0.625), (-110.5, 23.3, -190.2), 12.93); s.sphere(s.dielectric(0.625), (-110.5, 23.3, -190.2), 10.34, flip=True)
    s.triangle(glass, (-114.5, 14.2, -198.0), (-133.4, 25.0, -177.4), (-78.0, 8.3, -183.3))
    cam = scenes.camera(64.1, (4.8, 89.6, 81.9), (-4.65, -28.35, 0.0))
    prm = scenes.path_params(abi.RT_PATH_SKY, 64, seed=33)
    o = oracle.render_path(s, cam, prm, 192, 128, threads=8)
    a = _rel_queries(ctx, s, cam, prm, 192, 128, 8)
    b = _rel_queries(ctx, s, cam, prm, 192, 128, 16)
    assert ctx.last_path_kernel()[0] == 2
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"])
    assert o["rays"] / (192 * 128 * 64) > 2.5
    assert abs(a["rays"] / o["rays"] - 1.0) <= 1.5e-2, a["rays"] / o["rays"] - 1.0


def test_glass_box_standing_on_a_box(ctx):
    """Two boxes of one orientation, the glass one standing on the diffuse one: top face and bottom face coincide, and the reference decides between them by the rounding of
    the same formula about two centres (04 Scene/Rectangle.hpp:58-70).  The strict slots of round 3 apply that formula - but to a ray ORIGIN that a fused o + d*t had put
    closer to its face than the reference's two-rounding hit point: +1.3e-2 scene queries (fuzz family C, seed 73; -5e-4 with the coplanar detection off).  Scenes with
    strict slots now round hit points as the reference does (PathArgs::strict_point): within 1.5e-3, generic and specialised kernel on the same paths."""
    s = scenes.Scene()
    s.plane(s.diffuse((0.6, 0.6, 0.5)), (0, -0.5, 0), (0, 1, 0))
    s.box(s.diffuse((0.7, 0.3, 0.3)), (0.066, -0.253, -0.762), (0.714, 0.494, 0.614), (0, 0, 0))
    s.box(s.dielectric(1.5), (0.066, -0.5 + 0.494 + 0.2425, -0.762), (0.428, 0.485, 0.368), (0, 0, 0))
    cam = scenes.camera(53.5, (-0.27, 1.09, 2.6), (-6.4, -9.0, 0.0))
    prm = scenes.path_params(abi.RT_PATH_SKY, 256, seed=73)
    o = oracle.render_path(s, cam, prm, 96, 64, threads=8)
    a = _rel_queries(ctx, s, cam, prm, 96, 64, 8)
    b = _rel_queries(ctx, s, cam, prm, 96, 64, 16)
    assert ctx.last_path_kernel()[0] == 2
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"])
    assert abs(a["rays"] / o["rays"] - 1.0) <= 1.5e-3, a["rays"] / o["rays"] - 1.0


def test_glass_ground_with_a_short_tmax(ctx):
    """The sphere tMax rule (Common/Ray.hpp:127-134) inside a glass sphere wider than tMax: the far root lies beyond tMax, the element reports a hit AT tMax, the path refracts at
    that virtual surface and goes on - the same way again and again (11 scene queries per sample here).  Refract assumes |d| = 1 and multiplies what |d|^2 is off by with eta^2
    each time (2.8^n at eta = 1/0.6); the reference normalises every ray, the fast kernels did not: +2.4 % scene queries, channel means off by 7 % (fuzz family A, seed 7 at
    tMax = 12).  Scenes where that can accumulate now normalise scattered directions (path_accumulate: norm_dirs): within 1e-3, generic = specialised."""
    s = scenes.Scene()
    s.sphere(s.dielectric(0.6), (0, -100.5, -1.0), 100.0)
    s.rect(s.diffuse((0.7, 0.4, 0.3)), (1.66, 0.75, -3.09), (1.05, 0.6), (48.3, 0.0, -30.9))
    s.sphere(s.diffuse((0.3, 0.4, 0.8)), (0.37, 1.36, -1.44), 0.5)
    cam = scenes.camera(49.9, (-0.57, 0.28, 2.63), (-15.5, 24.7, 0.0))
    prm = scenes.path_params(abi.RT_PATH_SKY, 64, seed=7, t_max=12.0)
    o = oracle.render_path(s, cam, prm, 192, 128, threads=8)
    a = _rel_queries(ctx, s, cam, prm, 192, 128, 8)
    b = _rel_queries(ctx, s, cam, prm, 192, 128, 16)
    assert ctx.last_path_kernel()[0] == 2
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"])
    assert o["rays"] / (192 * 128 * 64) > 6.0                      # the paths really run on inside the sphere
    assert abs(a["rays"] / o["rays"] - 1.0) <= 1e-3, a["rays"] / o["rays"] - 1.0
    mo = (o["sum"] / 64).reshape(-1, 3).mean(axis=0)
    assert np.abs(a["rgb"].reshape(-1, 3).mean(axis=0) / mo - 1.0).max() <= 5e-3


def _far_small_spheres(n=600):
    """Diffuse and metal spheres of radius 0.3-1.5 up to 160 units away, nothing else (tools/bias_scan_stress.py: 'no mesh, no plane')."""
    g = scenes._PCG32(0x5EED); s = scenes.Scene()
    for i in range(n):
        c = (g.u(-60, 60), g.u(-4, 40), g.u(-160, -10)); r = g.u(0.3, 1.5)
        alb = (g.u(0.2, 0.9), g.u(0.2, 0.9), g.u(0.2, 0.9))
        s.sphere(s.metal(alb, 0.2) if i % 2 else s.diffuse(alb), c, r)
    return s, scenes.camera(60.0, (0, 5, 20), (-10, 0, 0))


@pytest.mark.parametrize("n_spheres", [300, 600])      # 33 KB of slots and materials: the LDS-resident kernel with early-outs; 67 KB: the LDS-tiled kernel
def test_reference_arithmetic_for_far_small_spheres(ctx, n_spheres):
    """Small spheres far from where the rays start, 03 estimator at tMin = 0.001: b*b - 4ac cancels from |oc|^2 ~ 1e4 to r^2 ~ 1, so
    the hit point lies up to ~1e-3 inside or outside the surface and the next ray re-hits it or not by rounding.  The fast arithmetic
    traces +0.5...0.9 % more queries than the reference's (documented); RT_ARITH_REFERENCE - directions normalised as
    Ray's constructor does, Ray::intersectsSphere's own operations with a = d.d, hit points rounded as there - follows it to a few 1e-4.
    Also: the option is refused together with the BVH, and switching it back restores the fast kernels' paths."""
    s, cam = _far_small_spheres(n_spheres)
    prm = scenes.path_params(abi.RT_PATH_SKY, 64, max_depth=8, seed=3)
    w, h = 196, 140
    o = oracle.render_path(s, cam, prm, w, h, threads=8)
    fast = ctx.render_path(s, cam, prm, w, h)
    ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_REFERENCE)
    try:
        ref = ctx.render_path(s, cam, prm, w, h)
        assert ctx.last_path_form()[0] == 0                      # one lane per pixel
        assert ctx.last_path_kernel()[0] == (1 if n_spheres == 300 else 0)
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
        with pytest.raises(RuntimeError, match="brute force"):
            ctx.render_path(s, cam, prm, w, h)
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
    finally:
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE); ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_FAST)
    again = ctx.render_path(s, cam, prm, w, h)
    assert again["rays"] == fast["rays"] and np.array_equal(again["rgb"], fast["rgb"])
    assert abs(ref["rays"] / o["rays"] - 1.0) <= 1e-3, ref["rays"] / o["rays"] - 1.0
    if n_spheres == 600: assert 2e-3 <= fast["rays"] / o["rays"] - 1.0 <= 1.5e-2, fast["rays"] / o["rays"] - 1.0      # the documented deviation of the fast arithmetic, pinned
    mo = (o["sum"] / 64).reshape(-1, 3).mean(axis=0)
    assert np.abs(ref["rgb"].reshape(-1, 3).mean(axis=0) / mo - 1.0).max() <= 1e-3


@pytest.mark.parametrize("which", ["03", "04", "nest"])
def test_reference_arithmetic_kernel_forms_agree(ctx, which):
    """RT_ARITH_REFERENCE: the generic kernel, the scene-specialised kernel and (with a scene cut off beyond 48 KB) the tiled kernel
    trace the same paths, and they are the oracle's to the tolerance of the default arithmetic's own tests."""
    from conftest import nest_scene
    if which == "03": s, cam, prm = scenes.scene03(16, seed=4)
    elif which == "04": s, cam, prm = scenes.scene04(16, seed=4)
    else:
        s, cam = nest_scene(); prm = scenes.path_params(abi.RT_PATH_LIGHTS, 16, seed=4)
    w, h = 160, 120
    o = oracle.render_path(s, cam, prm, w, h, threads=8)
    ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_REFERENCE)
    try:
        a = _rel_queries(ctx, s, cam, prm, w, h, 8)
        assert ctx.last_path_kernel()[0] == 1
        b = _rel_queries(ctx, s, cam, prm, w, h, 16)
        assert ctx.last_path_kernel()[0] == 2, ctx.last_path_kernel()[1]
    finally:
        ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_FAST)
    assert a["rays"] == b["rays"] and np.array_equal(a["rgb"], b["rgb"])
    assert abs(a["rays"] / o["rays"] - 1.0) <= (5e-3 if which == "nest" else 5e-4), a["rays"] / o["rays"] - 1.0
