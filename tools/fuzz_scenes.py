"""Random path-traced scenes for the fuzz tools (tools/fuzz_path*.py) and tests/test_gpu_fuzz.py.  Families: a - the ingredients of the reference scenes at their scales;
b - other scales, giant ground spheres, the camera inside glass or a box; c - coplanar constructions; d - 80 ... 1500 primitives.  Each returns (scene, camera, variant)."""
import numpy as np
from rtamd import abi, scenes


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

def random_scene_b(seed):
    """Family B (FUZZ_FAMILY=b): the same ingredients at scales the reference scenes do not have - the whole scene times 1 / 10 / 50, sphere radii log-uniform over
    0.02 ... 30, ground spheres of radius 20 ... 1000, thin boxes, the camera sometimes inside a glass sphere or a box."""
    g = np.random.default_rng(5000 + seed)
    U = lambda a, b: float(g.uniform(a, b))
    s = scenes.Scene()
    variant = abi.RT_PATH_SKY if seed % 2 else abi.RT_PATH_LIGHTS
    k = float(g.choice([1.0, 1.0, 10.0, 50.0]))
    mats = [s.diffuse(tuple(g.uniform(0.1, 0.9, 3))) for _ in range(int(g.integers(1, 4)))]
    mats += [s.metal(tuple(g.uniform(0.5, 0.95, 3)), float(g.choice([0.0, 0.1, 0.4, 1.0]))) for _ in range(int(g.integers(0, 3)))]
    if g.random() < 0.7: mats.append(s.dielectric(float(g.choice([0.6, 0.66, 0.8, 1.5]))))
    pick = lambda: int(g.choice(mats))
    angles = lambda: (float(g.choice([0, 0, 90, U(-180, 180)])), float(g.choice([0, 0, 90, U(-180, 180)])), float(g.choice([0, 0, U(-180, 180)])))
    if variant == abi.RT_PATH_LIGHTS:
        lm = s.light_material(tuple(g.uniform(4, 20, 3)))
        s.plane(pick(), (0, -5 * k, 0), (0, 1, 0))
        if g.random() < 0.5:
            s.plane(pick(), (0, 5 * k, 0), (0, -1, 0)); s.plane(pick(), (0, 0, -5 * k), (0, 0, 1)); s.plane(pick(), (-5 * k, 0, 0), (1, 0, 0)); s.plane(pick(), (5 * k, 0, 0), (-1, 0, 0))
        s.rect(lm, (U(-2, 2) * k, 4.95 * k, U(-2, 2) * k), (U(1, 4) * k, U(1, 4) * k), (90, 0, 0))
        cam_pos = (U(-1, 1) * k, U(-1, 1) * k, 10 * k); cam_ori = (U(-5, 5), U(-5, 5), 0); fov = U(60, 100)
        lo, hi = (-4 * k, -4.5 * k, -4 * k), (4 * k, 3 * k, 4 * k)
    else:
        R = float(g.choice([20.0, 50.0, 100.0, 1000.0])) * k
        if g.random() < 0.6: s.sphere(pick(), (0, -0.5 * k - R, -1 * k), R)
        else: s.plane(pick(), (0, -0.5 * k, 0), (0, 1, 0))
        cam_pos = (U(-2, 2) * k, U(0.2, 2) * k, U(1, 3) * k); cam_ori = (U(-30, 0), U(-40, 40), 0); fov = U(40, 90)
        lo, hi = (-2.5 * k, -0.4 * k, -4 * k), (2.5 * k, 1.5 * k, 0)
    P = lambda: tuple(float(g.uniform(a, b)) for a, b in zip(lo, hi))
    scale = k if variant == abi.RT_PATH_LIGHTS else 0.3 * k
    for _ in range(int(g.integers(2, 8))):
        kind = g.choice(["sphere", "sphere", "box", "thin", "rect", "tri", "shell"])
        if kind == "sphere": s.sphere(pick(), P(), float(np.exp(g.uniform(np.log(0.02), np.log(30.0)))) * scale / 3)
        elif kind == "shell":
            c = P(); r = U(0.8, 2.0) * scale; gl = s.dielectric(0.625)
            s.sphere(gl, c, r); s.sphere(gl, c, r * 0.8, flip=True)
        elif kind == "box": s.box(pick(), P(), tuple(g.uniform(0.5, 3.0, 3) * scale), angles(), flip=bool(g.random() < 0.15))
        elif kind == "thin": s.box(pick(), P(), (U(0.5, 3) * scale, 0.002 * scale, U(0.5, 3) * scale), angles())
        elif kind == "rect": s.rect(pick(), P(), (U(1, 4) * scale, U(1, 4) * scale), angles())
        else:
            a = np.array(P()); s.triangle(pick(), tuple(a), tuple(a + g.uniform(-3, 3, 3) * scale), tuple(a + g.uniform(-3, 3, 3) * scale))
    r = g.random()
    if r < 0.15: s.sphere(s.dielectric(0.66), cam_pos, 0.5 * scale)                                  # the camera inside a glass sphere
    elif r < 0.3: s.box(pick(), cam_pos, (3 * scale, 3 * scale, 3 * scale), angles(), flip=True)     # ... inside a box that faces inwards
    return s, scenes.camera(fov, cam_pos, cam_ori), variant

def random_scene_c(seed):
    """Family C (FUZZ_FAMILY=c): coplanar elements of different arithmetic - boxes resting exactly on the floor plane (turned about the vertical), boxes stacked on boxes,
    rectangles lying on box tops and on the floor, triangles in the floor plane, a box against a wall; glass among them.  In exact arithmetic those distances tie; the
    reference decides by the rounding of two formulas, the scene compiler must find every such pair (rt_api.hip find_coplanar) and test it in the reference's sequence."""
    g = np.random.default_rng(9000 + seed)
    U = lambda a, b: float(g.uniform(a, b))
    s = scenes.Scene()
    variant = abi.RT_PATH_SKY if seed % 2 else abi.RT_PATH_LIGHTS
    mats = [s.diffuse(tuple(g.uniform(0.2, 0.9, 3))) for _ in range(3)] + [s.metal(tuple(g.uniform(0.5, 0.95, 3)), float(g.choice([0.0, 0.3]))), s.dielectric(0.66), s.dielectric(1.5)]
    pick = lambda: int(g.choice(mats))
    if variant == abi.RT_PATH_LIGHTS:
        floor = -5.0
        s.plane(pick(), (0, floor, 0), (0, 1, 0)); s.plane(pick(), (0, 5, 0), (0, -1, 0)); s.plane(pick(), (0, 0, -5), (0, 0, 1)); s.plane(pick(), (-5, 0, 0), (1, 0, 0)); s.plane(pick(), (5, 0, 0), (-1, 0, 0))
        s.rect(s.light_material((15, 15, 15)), (0, 4.95, 0), (3, 3), (90, 0, 0))
        cam = scenes.camera(90.0, (0, 0, 10), (0, 0, 0)); span, sz = 3.5, (0.8, 2.5)
    else:
        floor = -0.5
        s.plane(pick(), (0, floor, 0), (0, 1, 0))
        cam = scenes.camera(U(50, 80), (U(-1, 1), U(0.5, 1.5), U(1.5, 3)), (U(-25, -5), U(-20, 20), 0)); span, sz = 2.0, (0.3, 0.9)
    for _ in range(int(g.integers(1, 4))):
        w, h, d = U(*sz), U(*sz), U(*sz); x, z = U(-span, span), U(-span, span) - (0 if variant == abi.RT_PATH_LIGHTS else 2.5)
        yaw = float(g.choice([0.0, 0.0, 90.0, U(-90, 90)]))
        s.box(pick(), (x, floor + h / 2, z), (w, h, d), (0, yaw, 0))                              # rests on the floor
        r = g.random()
        if r < 0.35:
            h2 = U(*sz) * 0.6; s.box(pick(), (x, floor + h + h2 / 2, z), (w * 0.6, h2, d * 0.6), (0, yaw, 0))      # stacked: shares the top face's plane
        elif r < 0.6: s.rect(pick(), (x, floor + h, z), (w * 0.7, d * 0.7), (90, yaw, 0))          # a rectangle lying on the top face
    if g.random() < 0.5: s.rect(pick(), (U(-span, span), floor, U(-span, span)), (U(*sz), U(*sz)), (90, 0, 0))      # a rectangle lying on the floor
    if g.random() < 0.5:
        a = np.array([U(-span, span), floor, U(-span, span)])
        s.triangle(pick(), tuple(a), tuple(a + np.array([U(0.5, 2), 0, U(-1, 1)])), tuple(a + np.array([U(-1, 1), 0, U(0.5, 2)])))    # a triangle in the floor plane
    if variant == abi.RT_PATH_LIGHTS and g.random() < 0.5:
        w, h, d = U(*sz), U(*sz), U(*sz); s.box(pick(), (-5 + w / 2, floor + h / 2, U(-3, 3)), (w, h, d), (0, 0, 0))    # in the corner: floor AND wall
    s.sphere(pick(), (U(-span, span), floor + 0.5 * sz[0], U(-span, span) - (0 if variant == abi.RT_PATH_LIGHTS else 2.5)), 0.5 * sz[0])
    return s, cam, variant

def random_scene_d(seed):
    """Family D (FUZZ_FAMILY=d): 80 ... 1500 primitives - the kernels beyond the scene-specialised builds: LDS-resident with wave-level early-outs (from 64 slots), tiled
    through LDS (beyond 48 KB) - and the BVH over them.  Small spheres, triangles, rotated rectangles and boxes in a volume, on a floor; both estimators."""
    g = np.random.default_rng(13000 + seed)
    U = lambda a, b: float(g.uniform(a, b))
    s = scenes.Scene()
    variant = abi.RT_PATH_SKY if seed % 2 else abi.RT_PATH_LIGHTS
    mats = [s.diffuse(tuple(g.uniform(0.2, 0.9, 3))) for _ in range(4)] + [s.metal(tuple(g.uniform(0.5, 0.95, 3)), float(g.choice([0.0, 0.3]))), s.dielectric(0.66)]
    pick = lambda: int(g.choice(mats))
    s.plane(pick(), (0, -2, 0), (0, 1, 0))
    if variant == abi.RT_PATH_LIGHTS: s.rect(s.light_material((12, 12, 12)), (0, 6, -6), (8, 8), (90, 0, 0))
    n = int(np.exp(g.uniform(np.log(80), np.log(1500))))
    for _ in range(n):
        c = (U(-6, 6), U(-1.8, 4), U(-14, -3)); k = g.random()
        if k < 0.55: s.sphere(pick(), c, U(0.05, 0.5))
        elif k < 0.85:
            a = np.array(c); s.triangle(pick(), tuple(a), tuple(a + g.uniform(-0.8, 0.8, 3)), tuple(a + g.uniform(-0.8, 0.8, 3)))
        elif k < 0.95: s.rect(pick(), c, (U(0.2, 1.0), U(0.2, 1.0)), (U(-180, 180), U(-180, 180), 0))
        else: s.box(pick(), c, tuple(g.uniform(0.2, 0.8, 3)), (U(-180, 180), U(-180, 180), 0))
    return s, scenes.camera(U(50, 80), (U(-1, 1), U(0, 2), U(0, 2)), (U(-15, 5), U(-15, 15), 0)), variant


def random_scene_e(seed):
    """Family E: the 04 estimator (lights, tMin 0.01) on a ground SPHERE of radius 100 ... 10000 under a light rectangle and a light sphere - big and giant spheres with
    the estimator whose attenuation carries the cosine; objects as in family A."""
    g = np.random.default_rng(21000 + seed)
    U = lambda a, b: float(g.uniform(a, b))
    s = scenes.Scene()
    mats = [s.diffuse(tuple(g.uniform(0.1, 0.9, 3))) for _ in range(int(g.integers(1, 4)))]
    mats += [s.metal(tuple(g.uniform(0.5, 0.95, 3)), float(g.choice([0.0, 0.2, 1.0]))) for _ in range(int(g.integers(0, 3)))]
    if g.random() < 0.6: mats.append(s.dielectric(float(g.choice([0.6, 0.66, 1.5]))))
    pick = lambda: int(g.choice(mats))
    R = float(g.choice([100.0, 1000.0, 10000.0]))
    s.sphere(pick(), (0, -5.0 - R, 0), R)
    lm = s.light_material(tuple(g.uniform(6, 20, 3)))
    s.rect(lm, (U(-2, 2), 6.0, U(-2, 2)), (U(3, 8), U(3, 8)), (90, 0, 0))
    if g.random() < 0.5: s.sphere(lm, (U(-6, 6), U(0, 4), U(-6, 2)), U(0.5, 1.5))
    for _ in range(int(g.integers(2, 6))):
        kind = g.choice(["sphere", "sphere", "box", "rect", "tri"])
        P = (U(-4, 4), U(-4.5, 2), U(-4, 4))
        if kind == "sphere": s.sphere(pick(), P, U(0.5, 2.0))
        elif kind == "box": s.box(pick(), P, tuple(g.uniform(0.5, 3.0, 3)), (0, U(-180, 180), 0))
        elif kind == "rect": s.rect(pick(), P, (U(1, 4), U(1, 4)), (U(-180, 180), U(-180, 180), 0))
        else:
            a = np.array(P); s.triangle(pick(), tuple(a), tuple(a + g.uniform(-3, 3, 3)), tuple(a + g.uniform(-3, 3, 3)))
    return s, scenes.camera(U(60, 100), (U(-1, 1), U(-1, 2), 10), (U(-10, 5), U(-8, 8), 0)), abi.RT_PATH_LIGHTS

def random_whitted_scene(seed, K=1.0):
    """Whitted scene (01 / 02): -> (scene, camera, max_depth, shadow_samples); K scales the whole scene (the kernels' tMin and shadow offsets stay)."""
    g = np.random.default_rng(seed)
    s = scenes.Scene()
    mats = []
    for _ in range(int(g.integers(2, 5))):
        r2r = float(g.choice([1.0, 1.0, 0.0, 0.5])) if seed % 3 else float(g.choice([1.0, 0.0]))     # every third seed: chains only
        mats.append(s.phong(tuple(g.uniform(0.1, 1.0, 3)), float(g.uniform(0, 1)), float(g.choice([1, 2, 8, 16, 50, 64.5, 300, 1000])),
                            float(g.choice([1.0, 0.8, 0.4, 0.0])), r2r, float(g.choice([1.0, 0.7, 1.5]))))
    pick = lambda: int(g.choice(mats + ([-1] if g.random() < 0.1 else [])))
    s.plane(pick(), (0, K * float(g.uniform(-5, -2)), 0), (0, 1, 0))
    if g.random() < 0.6: s.plane(pick(), (0, 0, K * float(g.uniform(-40, -20))), (float(g.uniform(-.3, .3)), float(g.uniform(-.3, .3)), 1))
    for _ in range(int(g.integers(1, 6))):
        s.sphere(pick(), (K * float(g.uniform(-6, 6)), K * float(g.uniform(-2, 4)), K * float(g.uniform(-18, -6))), K * float(g.uniform(0.5, 3)))
    for _ in range(int(g.integers(0, 3))):
        v = g.uniform(-8, 8, (3, 3)); v[:, 2] = g.uniform(-22, -8, 3)
        s.triangle(pick(), *[tuple(map(float, r * K)) for r in v])
    for _ in range(int(g.integers(1, 4))):
        s.light((K * float(g.uniform(-10, 10)), K * float(g.uniform(3, 12)), K * float(g.uniform(-10, 4))), tuple(g.uniform(0.2, 0.9, 3)), K * float(g.uniform(0.3, 2.0)))
    return s, scenes.camera(float(g.uniform(40, 90))), int(g.integers(1, 7)), int(g.choice([0, 0, 1, 3, 8]))


FAMILIES = {"": random_scene, "a": random_scene, "b": random_scene_b, "c": random_scene_c, "d": random_scene_d, "e": random_scene_e}
