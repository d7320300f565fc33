"""Flat scene descriptions (the literals of the four reference examples, and the stress scene).

Each builder cites the lines of the reference's main() that hold the literals
(paths relative to /root/reference/Sources)."""
import ctypes as C
import math

import numpy as np

from . import abi


class Scene:
    """Owns ctypes arrays of rt_prim / rt_material / rt_light and the rt_scene that points at them."""

    def __init__(self):
        self.prims, self.materials, self.lights = [], [], []
        self._c = None

    # --- materials -----------------------------------------------------------------------
    def phong(self, albedo=(1, 1, 1), specular_intensity=1.0, shininess=16.0,
              primary_to_secondary=1.0, reflect_to_refract=1.0, refraction_eta=1.0):
        """01_Basic/Types.h:9-23 aggregate (defaults are the member initialisers)."""
        m = abi.rt_material(type=abi.RT_MAT_PHONG)
        m.albedo[:] = albedo
        m.specular_intensity, m.shininess = specular_intensity, shininess
        m.primary_to_secondary, m.reflect_to_refract, m.refraction_eta = primary_to_secondary, reflect_to_refract, refraction_eta
        return self._add_mat(m)

    def diffuse(self, albedo):
        m = abi.rt_material(type=abi.RT_MAT_DIFFUSE); m.albedo[:] = albedo
        return self._add_mat(m)

    def metal(self, albedo, roughness=0.0):
        m = abi.rt_material(type=abi.RT_MAT_METAL); m.albedo[:] = albedo; m.roughness = roughness
        return self._add_mat(m)

    def dielectric(self, refraction_index):
        m = abi.rt_material(type=abi.RT_MAT_DIELECTRIC); m.refraction_index = refraction_index
        return self._add_mat(m)

    def light_material(self, color):
        m = abi.rt_material(type=abi.RT_MAT_LIGHT); m.albedo[:] = color
        return self._add_mat(m)

    def _add_mat(self, m):
        self.materials.append(m); self._c = None
        return len(self.materials) - 1

    # --- primitives ----------------------------------------------------------------------
    def _add_prim(self, type_, material, vals, flags=0):
        p = abi.rt_prim(type=type_, material=material, flags=flags)
        for i, v in enumerate(vals):
            p.f[i] = v
        self.prims.append(p); self._c = None
        return len(self.prims) - 1

    def sphere(self, material, center, radius, flip=False):
        return self._add_prim(abi.RT_PRIM_SPHERE, material, [*center, radius], abi.RT_PRIM_FLAG_FLIP_NORMALS if flip else 0)

    def plane(self, material, point, normal):
        # the reference's Plane constructors normalise the normal in float (01 Plane.hpp:27)
        n = np.asarray(normal, np.float32)
        ln = np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])  # float32 throughout
        n = n / ln if ln > 0 else n * 0
        return self._add_prim(abi.RT_PRIM_PLANE, material, [*point, *[float(x) for x in n]])

    def rect(self, material, position, sizes=(1, 1), orientation=(0, 0, 0)):
        return self._add_prim(abi.RT_PRIM_RECT, material, [*position, *sizes, *orientation])

    def box(self, material, position, sizes=(1, 1, 1), orientation=(0, 0, 0), flip=False):
        """04 Scene/Box.hpp:34-40"""
        return self._add_prim(abi.RT_PRIM_BOX, material, [*position, *sizes, *orientation], abi.RT_PRIM_FLAG_FLIP_NORMALS if flip else 0)

    def triangle(self, material, v0, v1, v2):
        return self._add_prim(abi.RT_PRIM_TRIANGLE, material, [*v0, *v1, *v2])

    def light(self, position, color, radius=0.0):
        l = abi.rt_light(); l.position[:] = position; l.color[:] = color; l.radius = radius
        self.lights.append(l); self._c = None

    # --- ctypes view ---------------------------------------------------------------------
    def c(self):
        if self._c is None:
            pa = (abi.rt_prim * max(1, len(self.prims)))(*self.prims)
            ma = (abi.rt_material * max(1, len(self.materials)))(*self.materials)
            la = (abi.rt_light * max(1, len(self.lights)))(*self.lights)
            sc = abi.rt_scene(C.cast(pa, C.POINTER(abi.rt_prim)), len(self.prims),
                              C.cast(ma, C.POINTER(abi.rt_material)), len(self.materials),
                              C.cast(la, C.POINTER(abi.rt_light)), len(self.lights))
            self._c = (sc, pa, ma, la)
        return self._c[0]


def camera(fov, position=(0, 0, 0), orientation=(0, 0, 0)):
    cam = abi.rt_camera(fov_deg=fov)
    cam.position[:] = position
    cam.orientation_deg[:] = orientation
    return cam


def whitted_params(max_depth=4, shadow_samples=0, seed=1):
    return abi.rt_whitted_params(max_depth=max_depth, shadow_samples=shadow_samples, seed=seed)


def path_params(variant, spp, max_depth=None, seed=1, sample_begin=0, sample_end=0, t_min=None, t_max=1000.0):
    if max_depth is None:
        max_depth = 24 if variant == abi.RT_PATH_SKY else 6
    if t_min is None:
        t_min = 0.001 if variant == abi.RT_PATH_SKY else 0.01
    return abi.rt_path_params(variant=variant, spp=spp, max_depth=max_depth, seed=seed,
                              sample_begin=sample_begin, sample_end=sample_end, t_min=t_min, t_max=t_max)


# ---- the four examples ---------------------------------------------------------------------
def scene01():
    """01_Basic/Main.cpp:182-203; camera: Render(&fb, 90.0f, ...) :206."""
    s = Scene()
    red = s.phong((1, 0, 0), 0.0); green = s.phong((0, 1, 0), 0.0); white = s.phong((1, 1, 1), 0.0)
    rubber = s.phong((0.6, 0.2, 0.2), 0.1, 16.0)
    mirror = s.phong((0.1, 0.1, 0.1), 1.0, 512.0, 0.25, 1.0)
    glass = s.phong((0.1, 0.1, 0.1), 1.0, 128.0, 0.25, 0.0, 0.6)
    s.plane(white, (0, -10, 0), (0, 1, 0)); s.plane(white, (0, 10, 0), (0, -1, 0))
    s.plane(white, (0, 0, -20), (0, 0, 1)); s.plane(white, (0, 0, 0.001), (0, 0, -1))
    s.plane(green, (10, 0, 0), (-1, 0, 0)); s.plane(red, (-10, 0, 0), (1, 0, 0))
    s.sphere(glass, (0, -7, -12), 3.0); s.sphere(rubber, (7.5, -7, -15), 2.0); s.sphere(mirror, (-7.5, -7, -15), 2.0)
    s.light((0, 8.5, -10), (0.9, 0.9, 0.9))
    return s, camera(90.0), whitted_params(4, 0)


def scene02(shadow_samples=16, seed=1):
    """02_SoftShadows/Main.cpp:201-218; MAX_SHADOW_SAMPLES 16 (:15)."""
    s = Scene()
    red = s.phong((1, 0, 0), 0.0); green = s.phong((0, 1, 0), 0.0); white = s.phong((1, 1, 1), 0.0)
    rubber = s.phong((0.6, 0.2, 0.2), 0.1, 16.0)
    s.plane(white, (0, -10, 0), (0, 1, 0)); s.plane(white, (0, 10, 0), (0, -1, 0))
    s.plane(white, (0, 0, -20), (0, 0, 1)); s.plane(white, (0, 0, 0.001), (0, 0, -1))
    s.plane(green, (10, 0, 0), (-1, 0, 0)); s.plane(red, (-10, 0, 0), (1, 0, 0))
    s.sphere(rubber, (0, -6.5, -15), 3.0)
    s.light((0, 6.5, -10), (0.9, 0.9, 0.9), 3.0)
    return s, camera(90.0), whitted_params(4, shadow_samples, seed)


def scene03(spp=16, seed=1, readme_variant=False):
    """03_PathTracingBasics/Main.cpp:172-189.  readme_variant=True builds the scene the README
    screenshot shows: the commented-out hollow glass sphere (:182-183) instead of the left metal one."""
    s = Scene()
    ground = s.diffuse((0.8, 0.8, 0.0)); center = s.diffuse((0.1, 0.2, 0.5))
    left = s.metal((0.8, 0.8, 0.8), 0.3); right = s.metal((0.8, 0.6, 0.2), 0.4)
    glass = s.dielectric(0.625)
    s.sphere(ground, (0, -100.5, -1.0), 100.0)
    s.sphere(center, (0, 0, -1), 0.5)
    if readme_variant:
        s.sphere(glass, (-1, 0, -1), 0.5); s.sphere(glass, (-1, 0, -1), 0.4, flip=True)
    else:
        s.sphere(left, (-1, 0, -1), 0.5)
    s.sphere(right, (1, 0, -1), 0.5)
    return s, camera(50.0, (-2.0, 1.5, 1.0), (-30.0, -42.0, 0.0)), path_params(abi.RT_PATH_SKY, spp, seed=seed)


def scene04(spp=32, seed=1):
    """04_PathTracingLights/Main.cpp:181-203."""
    s = Scene()
    red = s.diffuse((0.65, 0.05, 0.05)); white = s.diffuse((0.73, 0.73, 0.73)); green = s.diffuse((0.12, 0.45, 0.15))
    light = s.light_material((15.0, 15.0, 15.0)); metal = s.metal((0.8, 0.8, 0.8), 0.2)
    ball = s.diffuse((0.1, 0.2, 0.5)); glass = s.dielectric(0.6)
    s.plane(white, (0, 5, 0), (0, -1, 0)); s.plane(white, (0, -5, 0), (0, 1, 0)); s.plane(white, (0, 0, -5), (0, 0, 1))
    s.plane(red, (-5, 0, 0), (1, 0, 0)); s.plane(green, (5, 0, 0), (-1, 0, 0))
    s.sphere(metal, (0, -3, -1), 2.0); s.sphere(ball, (-2, -4, 2.5), 1.0); s.sphere(glass, (2.5, -3.5, 3.0), 1.5)
    s.rect(light, (0, 4.95, 0), (3, 3), (90, 0, 0))
    return s, camera(90.0, (0, 0, 10), (0, 0, 0)), path_params(abi.RT_PATH_LIGHTS, spp, seed=seed)


# ---- stress scene (the reference defines none) -----------------------
class _PCG32:
    def __init__(self, seed, seq=54):
        self.state, self.inc = 0, ((seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
        self.next(); self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF; self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def u(self, a=0.0, b=1.0):
        return a + (b - a) * ((self.next() >> 8) / 16777216.0)


def scene_stress(n_spheres=10000, grid=64, spp=4096, seed=1):
    """10k spheres + (grid x grid x 2)-triangle height field + ground plane + emissive rectangle.
    Draw order per sphere: cx, cy, cz, radius, albedo r,g,b.  Then per height-field vertex (row-major): height."""
    g = _PCG32(0x5EED)
    s = Scene()
    for i in range(n_spheres):
        c = (g.u(-60, 60), g.u(-4, 40), g.u(-160, -10)); r = g.u(0.3, 1.5)
        alb = (g.u(0.2, 0.9), g.u(0.2, 0.9), g.u(0.2, 0.9))
        k = i % 4
        m = s.metal(alb, 0.2) if k == 1 else s.dielectric(0.6) if k == 2 else s.diffuse(alb)
        s.sphere(m, c, r)
    ground = s.diffuse((0.5, 0.5, 0.5))
    s.plane(ground, (0, -5, 0), (0, 1, 0))
    hf = s.diffuse((0.4, 0.6, 0.3))
    hs = [[-5.0 + 3.0 * g.u() for _ in range(grid + 1)] for _ in range(grid + 1)]
    x0, x1, z0, z1 = -60.0, 60.0, -160.0, -10.0
    for j in range(grid):
        for i in range(grid):
            xa, xb = x0 + (x1 - x0) * i / grid, x0 + (x1 - x0) * (i + 1) / grid
            za, zb = z0 + (z1 - z0) * j / grid, z0 + (z1 - z0) * (j + 1) / grid
            p00, p10 = (xa, hs[j][i], za), (xb, hs[j][i + 1], za)
            p01, p11 = (xa, hs[j + 1][i], zb), (xb, hs[j + 1][i + 1], zb)
            s.triangle(hf, p00, p10, p11); s.triangle(hf, p00, p11, p01)
    lm = s.light_material((15, 15, 15))
    s.rect(lm, (0, 45, -85), (20, 20), (90, 0, 0))
    cam = camera(60.0, (0, 5, 20), (-10, 0, 0))
    return s, cam, path_params(abi.RT_PATH_LIGHTS, spp, max_depth=8, seed=seed)


# algorithmic FLOPs per ray: 17/sphere, 12/plane, 38/rect, 45/triangle + shade
def flops_per_ray(scene, shade):
    n = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
    for p in scene.prims:
        n[p.type] += 1
    return 17 * n[0] + 12 * n[1] + 38 * (n[2] + 6 * n[4]) + 45 * n[3] + shade     # a box is six rectangle tests
