"""Coplanar elements of different arithmetic (VERDICT r02 "weak" 1a) and the far-small-sphere figure (1b).

The reference evaluates a box face / rectangle in the object frame (04 Scene/Rectangle.hpp:58-70, Scene/Box.hpp:62-131) and a plane
or triangle in the world frame (Common/Ray.hpp:150-249).  Where two such elements share a plane the two distances are equal in exact
arithmetic and the ROUNDING of the two formulas decides which one a ray sees.  The library's fast slot forms would make that an exact
tie (scene order decides): a different image.  So the scene compiler finds such pairs and gives their slots the reference's own
operation sequence (rt_device.h PRIM_*_STRICT).  CPU part: the detection.  GPU part: HIP vs the oracle at equal seeds."""
import ctypes as C
import os

import numpy as np
import pytest

from rtamd import abi, lib, scenes
import orc as oracle


def _base():
    s = scenes.Scene()
    white = s.diffuse((0.8, 0.8, 0.8)); lm = s.light_material((9, 9, 9)); glass = s.dielectric(0.66)
    s.plane(white, (0, -2, 0), (0, 1, 0))
    return s, white, lm, glass


def box_on_floor(size=(1.2, 1.6, 1.2), orient=(0, 0, 0)):
    """A glass box resting on the floor plane y = -2 (centre y = -2 + height/2), lit by a rectangle: the most ordinary scene a
    user of Box.hpp builds; profiles/r02_bias_scan.txt:50 measured it at -5.3e-3 queries, -1.1e-2 image mean before the fix."""
    s, white, lm, glass = _base()
    s.box(glass, (-1.5, -2.0 + size[1] / 2.0, -4), size, orient)
    s.rect(lm, (0, 6, -6), (6, 6), (90, 0, 0))
    return s, scenes.camera(70.0, (0, 0.5, 1.0), (-3, 0, 0))


def _strict(L, s, variant=4):
    L.rt_debug_strict_slots.argtypes = [C.POINTER(abi.rt_scene), C.c_int32, C.POINTER(C.c_int32)]
    sc = s.c(); out = (C.c_int32 * 3)()
    n = L.rt_debug_strict_slots(C.byref(sc), variant, out)
    return n, list(out)


def test_detection_of_coplanar_pairs():
    """Host logic, no GPU: which slots take the reference's operation sequence."""
    L = lib.load()
    assert _strict(L, box_on_floor()[0]) == (2, [1, 1, 0])                       # the floor and the box's bottom face
    assert _strict(L, box_on_floor(orient=(0, 25, 0))[0]) == (2, [1, 1, 0])      # turned about the vertical: still resting on it
    s, white, lm, glass = _base(); s.box(glass, (-1.5, -1.1, -4), (1.2, 1.6, 1.2))
    assert _strict(L, s)[0] == 0                                                 # 0.1 above the floor: nothing coplanar
    s, white, lm, glass = _base(); s.box(glass, (-1.5, -1.2, -4), (1.2, 1.6, 1.2), (10, 0, 0))
    assert _strict(L, s)[0] == 0                                                 # tilted: the bottom face crosses the floor, no shared plane
    s, white, lm, glass = _base(); s.rect(white, (0, 0, -4), (2, 2)); s.rect(glass, (0, 0, -4), (2, 2))
    assert _strict(L, s)[0] == 0                                                 # identical arithmetic: an exact tie in the reference too (last wins)
    s, white, lm, glass = _base(); s.rect(white, (0, 0, -4), (2, 2)); s.rect(glass, (0.5, 0, -4), (2, 2))
    assert _strict(L, s) == (2, [0, 2, 0])                                       # same plane, other centre: (o - pos) rounds differently
    s, white, lm, glass = _base(); s.rect(white, (0, 0, -4), (2, 2)); s.rect(glass, (5, 0, -4), (2, 2))
    assert _strict(L, s)[0] == 0                                                 # same plane, but they do not overlap
    s, white, lm, glass = _base(); s.rect(white, (0, 0, -4), (2, 2)); s.triangle(glass, (-1, -1, -4), (1, -1, -4), (0, 1, -4))
    assert _strict(L, s) == (2, [0, 1, 1])
    s, white, lm, glass = _base(); s.triangle(white, (0, 0, -4), (1, 0, -4), (1, 1, -4)); s.triangle(white, (0, 0, -4), (1, 1, -4), (0, 1, -4))
    assert _strict(L, s)[0] == 0                                                 # a quad's two triangles touch along a line only
    s, white, lm, glass = _base(); s.triangle(glass, (-1, -2, -4), (1, -2, -4), (0, -2, -6))
    assert _strict(L, s) == (2, [1, 0, 1])                                       # a triangle lying in the floor
    # none of the reference's scenes, nor the stress scene, has such a pair: their kernels are untouched
    assert _strict(L, scenes.scene04(1)[0])[0] == 0 and _strict(L, scenes.scene03(1)[0], 3)[0] == 0
    assert _strict(L, scenes.scene03(1, readme_variant=True)[0], 3)[0] == 0
    assert _strict(L, scenes.scene_stress(spp=1)[0])[0] == 0
    # and the scene-specialised source with strict slots compiles (loading it needs a GPU)
    L.rt_debug_jit_probe.argtypes = [C.POINTER(abi.rt_scene), C.c_int32, C.c_char_p, C.c_int32]
    s, white, lm, glass = _base(); s.box(glass, (-1.5, -1.2, -4), (1.2, 1.6, 1.2)); s.triangle(glass, (-1, -2, -4), (1, -2, -4), (0, -2, -6))
    note = C.create_string_buffer(8192); sc = s.c()
    rc = L.rt_debug_jit_probe(C.byref(sc), 4, note, 8192)
    assert rc in (0, 1) and b"compile failed" not in note.value and b"hiprtcCreateProgram" not in note.value, note.value.decode()


W, H, SPP = 196, 140, 64


def _compare(ctx, s, cam, variant=abi.RT_PATH_LIGHTS, depth=6):
    prm = scenes.path_params(variant, SPP, max_depth=depth, seed=3)
    g = ctx.render_path(s, cam, prm, W, H)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    om = o["sum"] / SPP
    q = g["rays"] / o["rays"] - 1.0
    m = g["rgb"].reshape(-1, 3).mean(axis=0) / om.reshape(-1, 3).mean(axis=0) - 1.0
    return g, q, m


@pytest.mark.gpu
@pytest.mark.parametrize("size,orient", [((1.2, 1.6, 1.2), (0, 0, 0)), ((1.2, 1.2, 1.2), (0, 0, 0)), ((1.2, 1.6, 1.2), (0, 25, 0))])
def test_glass_box_on_a_floor_plane(ctx, size, orient):
    """HIP vs oracle at equal seeds: scene queries per sample within 5e-4, channel means within 2e-3;
    every kernel form renders the same frame bit for bit."""
    s, cam = box_on_floor(size, orient)
    os.environ["RT_MI355_JIT"] = "0"
    try:
        g, q, m = _compare(ctx, s, cam)
        assert ctx.last_path_kernel()[0] == 1
        os.environ["RT_MI355_JIT"] = "2"
        prm = scenes.path_params(abi.RT_PATH_LIGHTS, SPP, seed=3)
        j = ctx.render_path(s, cam, prm, W, H)
        assert ctx.last_path_kernel()[0] == 2, ctx.last_path_kernel()
    finally:
        os.environ.pop("RT_MI355_JIT", None)
    print(f"glass box {size} {orient} on the floor: queries rel {q:+.2e}, channel means rel {m}")
    assert abs(q) <= 5e-4, q
    assert np.abs(m).max() <= 2e-3, m
    assert j["rays"] == g["rays"] and np.array_equal(j["rgb"], g["rgb"])        # generated == generic
    ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
    try:
        b = ctx.render_path(s, cam, prm, W, H)
    finally:
        ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
    assert b["rays"] == g["rays"] and np.array_equal(b["rgb"], g["rgb"])        # BVH walk == brute force
    ctx.set_debug_flags(2)                                                      # the global-memory kernel (scalar loads)
    try:
        k = ctx.render_path(s, cam, prm, W, H)
    finally:
        ctx.set_debug_flags(0)
    assert k["rays"] == g["rays"] and np.array_equal(k["rgb"], g["rgb"])


@pytest.mark.gpu
def test_the_fast_forms_alone_would_miss_it():
    """The same scene with the detection switched off (RT_MI355_STRICT_COPLANAR=0, a fresh context: the switch is read when a scene
    is compiled) shows the deviation the strict slots remove - so the test above has teeth."""
    s, cam = box_on_floor()
    os.environ["RT_MI355_STRICT_COPLANAR"] = "0"
    c2 = lib.Context()
    try:
        g, q, m = _compare(c2, s, cam)
    finally:
        os.environ.pop("RT_MI355_STRICT_COPLANAR", None)
        c2.close()
    print(f"fast forms only: queries rel {q:+.2e}, channel means rel {m}")
    assert abs(q) > 2e-3 and np.abs(m).max() > 4e-3


@pytest.mark.gpu
def test_rectangle_in_a_triangle_plane_and_on_a_plane(ctx):
    """The other class pairs: a glass rectangle lying in a diffuse triangle's plane, a glass triangle lying in the floor."""
    s, white, lm, glass = _base()
    s.triangle(white, (-3, -1.5, -5), (1, -1.5, -5), (-1, 1.5, -5)); s.rect(glass, (-1, -0.5, -5), (1.5, 1.5))
    s.triangle(glass, (1, -2, -3), (3, -2, -3), (2, -2, -5))
    s.rect(lm, (0, 6, -6), (6, 6), (90, 0, 0))
    cam = scenes.camera(70.0, (0, 0.5, 1.0), (-3, 0, 0))
    assert _strict(lib.load(), s) == (4, [1, 1, 2])
    g, q, m = _compare(ctx, s, cam)
    print(f"rect in triangle plane, triangle in floor: queries rel {q:+.2e}, channel means rel {m}")
    assert abs(q) <= 5e-4 and np.abs(m).max() <= 2e-3, (q, m)


@pytest.mark.gpu
def test_far_small_spheres_at_config5_settings(ctx):
    """VERDICT r02 "weak" 1b, pinned: BASELINE config 5's scene (all 18,194 primitives: 10,000 small spheres up to 160 away) under
    its own settings (04 estimator, tMin 0.01, depth 8), HIP vs oracle at equal seeds.  Hit points on far small spheres carry
    +-1e-4 of noise along the ray in the reference and here alike, and how often a path then self-hits depends on the rounding
    sequence: profiles/r02_bias_scan.txt measured +2.4e-4 ... +6.6e-4 in scene queries per sample at 32 spp.  Bound: 1e-3 in
    queries per sample.  (The image mean at this sample count is noise - a few hundred light hits - and is bounded at full
    size by tests/test_gpu_config5.py.)  (A reduced scene - 1,500 spheres - sits at +1.0e-3: the figure
    belongs to the scene, so the stated one is the one pinned.)"""
    s, cam, prm = scenes.scene_stress(spp=8, seed=2)
    w, h = 192, 108
    g = ctx.render_path(s, cam, prm, w, h)
    o = oracle.render_path(s, cam, prm, w, h, threads=16)
    q = g["rays"] / o["rays"] - 1.0
    m = g["rgb"].mean() / (o["sum"] / prm.spp).mean() - 1.0
    print(f"far small spheres, config 5 scene and settings: queries rel {q:+.2e}, mean rel {m:+.2e}")
    assert abs(q) <= 1e-3, q
    assert abs(m) <= 0.3, m
