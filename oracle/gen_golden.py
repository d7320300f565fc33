#!/usr/bin/env python3
"""Generates tests/golden/kat_*.npz by calling the REFERENCE's own header-only code (compiled from
/root/reference into oracle/_ref/libref_e{1,3,4}.so by oracle/Makefile) on seeded inputs.

Run in the build container only (needs /root/reference); the .npz files are data (inputs + expected
outputs) and are committed, so the tests need neither the reference nor oracle/_ref.
Deterministic routines are stored as exact vectors; the reference's samplers/materials draw from
clock-seeded engines that cannot be seeded from outside, so they are stored as statistics."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
FP = C.POINTER(C.c_float)


def fp(a):
    return a.ctypes.data_as(FP)


def load(n):
    return C.CDLL(os.path.join(HERE, "_ref", f"libref_e{n}.so"))


def unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def gen_math(L, rng):
    n = 1000
    a = rng.normal(size=(n, 3)).astype(np.float32) * 3
    a[0] = 0; a[1] = (1e-20, 0, 0); a[2] = (1, 1, 1)
    nrm = unit(rng, n)
    d = unit(rng, n)
    eta = rng.uniform(0.3, 2.5, n).astype(np.float32)
    deg = rng.uniform(-180, 180, (n, 3)).astype(np.float32)
    deg[0] = 0; deg[1] = (90, 0, 0); deg[2] = (-90, 0, 0); deg[3] = (0, 90, 0); deg[4] = (-30, -42, 0); deg[5] = (0, 0, 45)
    out = {k: np.zeros((n, 3), np.float32) for k in ("normalize", "reflect", "refract_tir", "refract_zero", "mix", "ray_dir")}
    rot = np.zeros((n, 9), np.float32)
    ratio = rng.uniform(-0.5, 1.5, n).astype(np.float32)
    clampv = np.zeros(n, np.float32)
    L.ref_clamp.restype = C.c_float
    L.ref_clamp.argtypes = [C.c_float] * 3
    L.ref_refract.argtypes = [FP, FP, C.c_float, C.c_int, FP]
    L.ref_mix.argtypes = [FP, FP, C.c_float, FP]
    for i in range(n):
        L.ref_normalize(fp(a[i]), fp(out["normalize"][i]))
        L.ref_reflect(fp(d[i]), fp(nrm[i]), fp(out["reflect"][i]))
        L.ref_refract(fp(d[i]), fp(nrm[i]), float(eta[i]), 1, fp(out["refract_tir"][i]))
        L.ref_refract(fp(d[i]), fp(nrm[i]), float(eta[i]), 0, fp(out["refract_zero"][i]))
        L.ref_rotation_mat(fp(deg[i]), fp(rot[i]))
        L.ref_mix(fp(a[i]), fp(d[i]), float(ratio[i]), fp(out["mix"][i]))
        L.ref_ray_ctor(fp(nrm[i]), fp(a[i]), fp(out["ray_dir"][i]))
        clampv[i] = L.ref_clamp(float(a[i, 0]), 0.0, 1.0)
    np.savez_compressed(os.path.join(OUT, "kat_math.npz"), a=a, n=nrm, d=d, eta=eta, deg=deg, ratio=ratio, rot=rot, clamp=clampv, **out)


def gen_intersect(L, rng):
    n = 4000
    o = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    d = unit(rng, n)
    c = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    r = rng.uniform(0.2, 6, n).astype(np.float32)
    tmin = rng.choice(np.array([0.0, 0.001, 0.01], np.float32), n)
    tmax = rng.choice(np.array([1000.0, 5.0, 2.0, 0.5], np.float32), n)
    # hand-made: inside the sphere with the far root beyond tMax (hit AT tMax), tangent, behind
    o[:3] = 0; d[:3] = (0, 0, -1); c[0] = (0, 0, 0); r[0] = 2; tmin[0] = 0.01; tmax[0] = 1.5
    c[1] = (1, 0, -5); r[1] = 1; tmin[1] = 0; tmax[1] = 1000
    c[2] = (0, 0, 5); r[2] = 1; tmin[2] = 0; tmax[2] = 1000
    L.ref_ray_sphere.argtypes = [FP, FP, FP, C.c_float, C.c_float, C.c_float, FP]
    L.ref_ray_plane.argtypes = [FP, FP, FP, FP, C.c_float, C.c_float, FP]
    L.ref_ray_rect_xy.argtypes = [FP, FP] + [C.c_float] * 7 + [FP]
    L.ref_ray_triangle.argtypes = [FP] * 5 + [C.c_float, C.c_float, FP, FP]
    sph_hit = np.zeros(n, np.int32); sph_t = np.zeros(n, np.float32)
    pn = unit(rng, n); pn[:200] = np.round(pn[:200]); p0 = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    pl_hit = np.zeros(n, np.int32); pl_t = np.zeros(n, np.float32)
    rect = np.stack([rng.uniform(-3, 3, n), rng.uniform(-4, 0, n), rng.uniform(0, 4, n), rng.uniform(-4, 0, n), rng.uniform(0, 4, n)], 1).astype(np.float32)
    dz0 = d.copy(); dz0[:50, 2] = 0    # d.z == 0: division by zero inside intersectsAARectangleXy (Ray.hpp:266)
    rc_hit = np.zeros(n, np.int32); rc_t = np.zeros(n, np.float32)
    v0 = rng.uniform(-4, 4, (n, 3)).astype(np.float32); v1 = v0 + rng.uniform(-4, 4, (n, 3)).astype(np.float32); v2 = v0 + rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    tr_hit = np.zeros(n, np.int32); tr_t = np.zeros(n, np.float32); tr_b = np.zeros((n, 2), np.float32)
    # aim half of the triangle rays at a point inside the triangle so hits are common
    w = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    target = v0 * w[:, :1] + v1 * w[:, 1:2] + v2 * w[:, 2:3]
    dt = target - o; dt = (dt / np.linalg.norm(dt, axis=1, keepdims=True)).astype(np.float32)
    dtri = np.where((np.arange(n) % 2 == 0)[:, None], dt, d).astype(np.float32)
    t = C.c_float(0)
    b2 = (C.c_float * 2)()
    for i in range(n):
        t.value = 0
        sph_hit[i] = L.ref_ray_sphere(fp(o[i]), fp(d[i]), fp(c[i]), float(r[i]), float(tmin[i]), float(tmax[i]), C.byref(t)); sph_t[i] = t.value if sph_hit[i] else 0
        t.value = 0
        pl_hit[i] = L.ref_ray_plane(fp(o[i]), fp(d[i]), fp(pn[i]), fp(p0[i]), float(tmin[i]), float(tmax[i]), C.byref(t)); pl_t[i] = t.value if pl_hit[i] else 0
        t.value = 0
        rc_hit[i] = L.ref_ray_rect_xy(fp(o[i]), fp(dz0[i]), *[float(x) for x in rect[i]], float(tmin[i]), float(tmax[i]), C.byref(t)); rc_t[i] = t.value if rc_hit[i] else 0
        t.value = 0
        tr_hit[i] = L.ref_ray_triangle(fp(o[i]), fp(dtri[i]), fp(v0[i]), fp(v1[i]), fp(v2[i]), float(tmin[i]), 1000.0, C.byref(t), b2)
        tr_t[i] = t.value if tr_hit[i] else 0; tr_b[i] = (b2[0], b2[1]) if tr_hit[i] else (0, 0)
    np.savez_compressed(os.path.join(OUT, "kat_intersect.npz"), o=o, d=d, c=c, r=r, tmin=tmin, tmax=tmax, sph_hit=sph_hit, sph_t=sph_t,
                        pn=pn, p0=p0, pl_hit=pl_hit, pl_t=pl_t, rect=rect, dz0=dz0, rc_hit=rc_hit, rc_t=rc_t,
                        v0=v0, v1=v1, v2=v2, dtri=dtri, tr_hit=tr_hit, tr_t=tr_t, tr_b=tr_b)
    print("intersect hits: sphere", sph_hit.mean(), "plane", pl_hit.mean(), "rect", rc_hit.mean(), "tri", tr_hit.mean())
    # 01/02 element wrappers (normals)
    m = 1000
    sh = np.zeros(m, np.int32); st = np.zeros(m, np.float32); sn = np.zeros((m, 3), np.float32)
    ph = np.zeros(m, np.int32); pt = np.zeros(m, np.float32); pnn = np.zeros((m, 3), np.float32)
    praw = (rng.normal(size=(m, 3)) * 2).astype(np.float32)      # un-normalised: the constructor normalises (Plane.hpp:27)
    L.ref_e1_sphere_hit.argtypes = [FP, C.c_float, FP, FP, C.c_float, C.c_float, FP, FP]
    L.ref_e1_plane_hit.argtypes = [FP, FP, FP, FP, C.c_float, C.c_float, FP, FP]
    for i in range(m):
        t.value = 0
        sh[i] = L.ref_e1_sphere_hit(fp(c[i]), float(r[i]), fp(o[i]), fp(d[i]), 0.001, 1000.0, C.byref(t), fp(sn[i])); st[i] = t.value if sh[i] else 0
        t.value = 0
        ph[i] = L.ref_e1_plane_hit(fp(p0[i]), fp(praw[i]), fp(o[i]), fp(d[i]), 0.001, 1000.0, C.byref(t), fp(pnn[i])); pt[i] = t.value if ph[i] else 0
    np.savez_compressed(os.path.join(OUT, "kat_e1_elements.npz"), o=o[:m], d=d[:m], c=c[:m], r=r[:m], sh=sh, st=st, sn=sn, p0=p0[:m], praw=praw, ph=ph, pt=pt, pn=pnn)


def gen_scenes(L3, L4, rng):
    ns, nr = 40, 200
    # 03: 4 spheres (+ inverted ones, + an exact duplicate to exercise "last wins")
    sph = np.zeros((ns, 5, 5), np.float32)
    sph[:, :, :3] = rng.uniform(-3, 3, (ns, 5, 3)); sph[:, :, 3] = rng.uniform(0.3, 2.5, (ns, 5)); sph[:, :, 4] = rng.random((ns, 5)) < 0.25
    sph[:, 4, :4] = sph[:, 1, :4]; sph[:, 4, 4] = 0           # element 4 duplicates element 1
    o = rng.uniform(-4, 4, (ns, nr, 3)).astype(np.float32); d = unit(rng, ns * nr).reshape(ns, nr, 3)
    out = np.zeros((ns, nr, 8), np.float32); hit = np.zeros((ns, nr), np.int32); which = np.full((ns, nr), -1, np.int32)
    L3.ref_e3_scene_hit.argtypes = [FP, C.c_int, FP, FP, C.c_float, C.c_float, FP, C.POINTER(C.c_int)]
    w = C.c_int(0)
    for s in range(ns):
        for k in range(nr):
            hit[s, k] = L3.ref_e3_scene_hit(fp(sph[s]), 5, fp(o[s, k]), fp(d[s, k]), 0.001, 1000.0, fp(out[s, k]), C.byref(w))
            which[s, k] = w.value if hit[s, k] else -1
    np.savez_compressed(os.path.join(OUT, "kat_scene_e3.npz"), spheres=sph, o=o, d=d, hit=hit, out=out * hit[..., None], which=which)
    # 04: spheres, planes, rectangles (single- and two-axis orientations)
    npr = 8
    prims = np.zeros((ns, npr, 13), np.float32)
    for s in range(ns):
        for p in range(npr):
            kind = rng.integers(0, 3)
            f = np.zeros(11, np.float32)
            if kind == 0:
                f[:3] = rng.uniform(-3, 3, 3); f[3] = rng.uniform(0.3, 2.0); flip = float(rng.random() < 0.25)
            elif kind == 1:
                f[:3] = rng.uniform(-5, 5, 3); nn = rng.normal(size=3); f[3:6] = nn / np.linalg.norm(nn); flip = 0.0
            else:
                f[:3] = rng.uniform(-3, 3, 3); f[3:5] = rng.uniform(0.5, 6, 2)
                f[5:8] = [(90, 0, 0), (0, 90, 0), (0, 0, 30), (rng.uniform(-180, 180), 0, 0), tuple(rng.uniform(-90, 90, 3))][rng.integers(0, 5)]
                flip = 0.0
            prims[s, p, 0] = kind; prims[s, p, 1] = flip; prims[s, p, 2:] = f
        prims[s, npr - 1] = prims[s, 0]                       # duplicate: the LAST one must win the tie
    out4 = np.zeros((ns, nr, 8), np.float32); hit4 = np.zeros((ns, nr), np.int32); which4 = np.full((ns, nr), -1, np.int32)
    L4.ref_e4_scene_hit.argtypes = [FP, C.c_int, FP, FP, C.c_float, C.c_float, FP, C.POINTER(C.c_int)]
    for s in range(ns):
        for k in range(nr):
            hit4[s, k] = L4.ref_e4_scene_hit(fp(prims[s]), npr, fp(o[s, k]), fp(d[s, k]), 0.01, 1000.0, fp(out4[s, k]), C.byref(w))
            which4[s, k] = w.value if hit4[s, k] else -1
    np.savez_compressed(os.path.join(OUT, "kat_scene_e4.npz"), prims=prims, o=o, d=d, hit=hit4, out=out4 * hit4[..., None], which=which4)
    print("scene hits: e3", hit.mean(), "e4", hit4.mean())


def angle_stats(dirs, axis, theta_max_deg, bins=30):
    """histogram of the polar angle to `axis` (degrees, [0,theta_max]) and of the azimuth, plus the mean vector."""
    axis = axis / np.linalg.norm(axis)
    cosang = np.clip(dirs @ axis, -1, 1)
    ang = np.degrees(np.arccos(cosang))
    h_ang = np.histogram(ang, bins=bins, range=(0, max(theta_max_deg, 1e-3) * 1.0001))[0]
    ref = np.cross(axis, np.array([0.3, -0.5, 0.8])); ref /= np.linalg.norm(ref); ref2 = np.cross(axis, ref)
    az = np.degrees(np.arctan2(dirs @ ref2, dirs @ ref)) % 360
    h_az = np.histogram(az, bins=12, range=(0, 360))[0]
    return h_ang.astype(np.int64), h_az.astype(np.int64), dirs.mean(axis=0).astype(np.float64), np.abs(np.linalg.norm(dirs, axis=1) - 1).max()


def gen_stats(L3, L4, rng):
    n = 400000
    axes = np.array([[0, 1, 0], [0, 0, 1], [0.6, -0.64, 0.48], [-1, 0, 0]], np.float32)
    axes = (axes / np.linalg.norm(axes, axis=1, keepdims=True)).astype(np.float32)
    thetas = np.array([90.0, 18.0, 24.0, 12.0], np.float32)
    L3.ref_e3_rnd_hemisphere.argtypes = [FP, C.c_float, C.c_int, FP]
    L4.ref_e4_rnd_hemisphere3.argtypes = [FP, C.c_float, C.c_int, FP]
    res = {}
    buf = np.zeros((n, 3), np.float32)
    for tag, fn in (("e3", L3.ref_e3_rnd_hemisphere), ("e4", L4.ref_e4_rnd_hemisphere3)):
        for i, (ax, th) in enumerate(zip(axes, thetas)):
            fn(fp(ax), float(th), n, fp(buf))
            dd = buf.astype(np.float64)
            if tag == "e3":
                dd = dd / np.linalg.norm(dd, axis=1, keepdims=True)     # RndHemisphereVec output is normalised later by setDirection
            ha, hz, mv, _ = angle_stats(dd, ax.astype(np.float64), float(th))
            res[f"{tag}_ang_{i}"] = ha; res[f"{tag}_az_{i}"] = hz; res[f"{tag}_mean_{i}"] = mv
    # materials: attenuation rule and reflect/refract split
    m = 200000
    dirs = np.zeros((m, 3), np.float32); atts = np.zeros((m, 3), np.float32)
    nrm = np.array([0, 1, 0], np.float32); pt = np.array([0.5, 0, -1], np.float32)
    ind = np.array([0.6, -0.8, 0.0], np.float32)
    alb = np.array([0.8, 0.6, 0.2], np.float32)
    flags = (C.c_int * 2)(); em = np.zeros(3, np.float32); att1 = np.zeros(3, np.float32)
    L4.ref_e4_scatter.argtypes = [C.c_int, FP, C.c_float, FP, FP, FP, C.c_int, C.c_int, FP, FP, C.POINTER(C.c_int), FP]
    L3.ref_e3_scatter.argtypes = [C.c_int, FP, C.c_float, FP, FP, FP, C.c_int, C.c_int, FP, FP]
    for kind, param, name in ((1, 0.0, "diffuse"), (2, 0.2, "metal")):
        L4.ref_e4_scatter(kind, fp(alb), param, fp(ind), fp(pt), fp(nrm), 1, m, fp(dirs), fp(atts), flags, fp(em))
        res[f"e4_{name}_att_mean"] = atts.astype(np.float64).mean(axis=0)
        res[f"e4_{name}_att_over_cos_max_err"] = np.abs(atts.astype(np.float64) - alb[None] * np.maximum(dirs.astype(np.float64) @ nrm, 0)[:, None]).max()
        res[f"e4_{name}_dir_mean"] = dirs.astype(np.float64).mean(axis=0)
        res[f"e4_{name}_flags"] = np.array([flags[0], flags[1]])
        L3.ref_e3_scatter(kind, fp(alb), param, fp(ind), fp(pt), fp(nrm), 1, m, fp(dirs), fp(att1))
        res[f"e3_{name}_att"] = att1.copy(); res[f"e3_{name}_dir_mean"] = dirs.astype(np.float64).mean(axis=0)
    L3.ref_e3_scatter(2, fp(alb), 0.0, fp(ind), fp(pt), fp(nrm), 1, 4, fp(dirs), fp(att1))          # roughness 0: exact mirror
    res["e3_metal_mirror_dir"] = dirs[:4].copy()
    # dielectric at several incidences, front and back faces: fraction of refracted outcomes + the two exact directions
    cosines = np.array([0.05, 0.2, 0.5, 0.8, 0.99], np.float32)
    frac = np.zeros((2, 2, len(cosines))); dset = np.zeros((2, 2, len(cosines), 2, 3), np.float32)
    for li, (Ls, is4) in enumerate(((L3, False), (L4, True))):
        for front in (1, 0):
            for ci, cc in enumerate(cosines):
                vin = np.array([np.sqrt(1 - cc * cc), -cc, 0], np.float32)
                if is4:
                    Ls.ref_e4_scatter(3, fp(alb), 0.6, fp(vin), fp(pt), fp(nrm), front, 20000, fp(dirs), fp(atts), flags, fp(em))
                else:
                    Ls.ref_e3_scatter(3, fp(alb), 0.6, fp(vin), fp(pt), fp(nrm), front, 20000, fp(dirs), fp(att1))
                dd = dirs[:20000]
                refl = dd[:, 1] > 0            # reflected rays leave on the normal's side
                frac[li, 1 - front, ci] = 1.0 - refl.mean()
                dset[li, 1 - front, ci, 0] = dd[refl][0] if refl.any() else 0
                dset[li, 1 - front, ci, 1] = dd[~refl][0] if (~refl).any() else 0
    res["dielectric_cosines"] = cosines; res["dielectric_refract_fraction"] = frac; res["dielectric_dirs"] = dset
    L4.ref_e4_scatter(4, fp(alb), 0.0, fp(ind), fp(pt), fp(nrm), 1, 1, fp(dirs), fp(atts), flags, fp(em))
    res["e4_light_front"] = np.array([flags[0], flags[1]]); res["e4_light_emitted"] = em.copy()
    L4.ref_e4_scatter(4, fp(alb), 0.0, fp(ind), fp(pt), fp(nrm), 0, 1, fp(dirs), fp(atts), flags, fp(em))
    res["e4_light_back"] = np.array([flags[0], flags[1]])
    np.savez_compressed(os.path.join(OUT, "stat_samplers_materials.npz"), axes=axes, thetas=thetas, n=n, **res)


def gen_boxes(L4):
    """scene::Box (04 Scene/Box.hpp) alone and among spheres/rectangles; rays from outside and from inside the boxes."""
    rng = np.random.default_rng(4040)
    ns, nr, npr = 40, 200, 5
    prims = np.zeros((ns, npr, 13), np.float32)
    orients = [(0, 0, 0), (0, 0, 0), (90, 0, 0), (0, 45, 0), (0, 0, 30), None]
    for s in range(ns):
        for p in range(npr):
            f = np.zeros(11, np.float32)
            kind = 4 if p < 3 else int(rng.integers(0, 3))
            flip = 0.0
            if kind == 4:
                f[:3] = rng.uniform(-3, 3, 3); f[3:6] = rng.uniform(0.5, 4, 3)
                oc = orients[int(rng.integers(0, len(orients)))]
                f[6:9] = oc if oc is not None else tuple(rng.uniform(-90, 90, 3))
                flip = float(rng.random() < 0.3)
            elif kind == 0:
                f[:3] = rng.uniform(-3, 3, 3); f[3] = rng.uniform(0.3, 2.0)
            elif kind == 1:
                f[:3] = rng.uniform(-5, 5, 3); nn = rng.normal(size=3); f[3:6] = nn / np.linalg.norm(nn)
            else:
                f[:3] = rng.uniform(-3, 3, 3); f[3:5] = rng.uniform(0.5, 6, 2); f[5:8] = (rng.uniform(-180, 180), 0, 0)
            prims[s, p, 0] = kind; prims[s, p, 1] = flip; prims[s, p, 2:] = f
    o = rng.uniform(-4, 4, (ns, nr, 3)).astype(np.float32); d = unit(rng, ns * nr).reshape(ns, nr, 3)
    o[:, : nr // 4] = prims[:, 0, 2:5][:, None, :] + rng.uniform(-0.2, 0.2, (ns, nr // 4, 3)).astype(np.float32)   # origins inside box 0
    out = np.zeros((ns, nr, 8), np.float32); hit = np.zeros((ns, nr), np.int32); which = np.full((ns, nr), -1, np.int32)
    L4.ref_e4_scene_hit.argtypes = [FP, C.c_int, FP, FP, C.c_float, C.c_float, FP, C.POINTER(C.c_int)]
    w = C.c_int(0)
    for s in range(ns):
        for k in range(nr):
            hit[s, k] = L4.ref_e4_scene_hit(fp(prims[s]), npr, fp(o[s, k]), fp(d[s, k]), 0.01, 1000.0, fp(out[s, k]), C.byref(w))
            which[s, k] = w.value if hit[s, k] else -1
    np.savez_compressed(os.path.join(OUT, "kat_box_e4.npz"), prims=prims, o=o, d=d, hit=hit, out=out * hit[..., None], which=which)
    print("box scenes: hit rate", hit.mean(), "box hits", np.isin(which, [0, 1, 2]).mean())


def gen_frame01(L):
    """01 is deterministic: whole frames by the reference's own classes (ref_e1_golden_frame follows Render/TraceTay,
    01 Main.cpp:324-496) - linear float colour at 196x140, Render's bytes at 196x140 and 784x561, TraceTay call counts."""
    out = {}
    for tag, (w, h) in (("196", (196, 140)), ("784", (784, 561))):
        rgb = np.zeros((h, w, 3), np.float32); bgra = np.zeros((h, w, 4), np.uint8); rays = C.c_ulonglong(0)
        L.ref_e1_golden_frame(w, h, rgb.ctypes.data_as(C.c_void_p), bgra.ctypes.data_as(C.c_void_p), C.byref(rays))
        if tag == "196": out["rgb" + tag] = rgb          # the float frame only at the small size (fixture size)
        out["bgra" + tag] = bgra; out["rays" + tag] = np.uint64(rays.value)
    np.savez_compressed(os.path.join(OUT, "frame_01.npz"), **out)
    print("01 frames: rays", int(out["rays196"]), int(out["rays784"]))


if __name__ == "__main__":
    import sys
    if sys.argv[1:] == ["frame01"]:
        gen_frame01(load(1)); raise SystemExit
    if sys.argv[1:] == ["boxes"]:          # added later: leaves the other fixtures (one shared generator state) untouched
        gen_boxes(load(4)); raise SystemExit
    rng = np.random.default_rng(20201)
    L1, L3, L4 = load(1), load(3), load(4)
    gen_math(L1, rng)
    gen_intersect(L1, rng)
    gen_scenes(L3, L4, rng)
    gen_stats(L3, L4, rng)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
