"""Per-pixel diagnostics: scene-query counts of one pixel, HIP (debug flag 4) vs the oracle."""
import sys, ctypes as C, math
import torch
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'software-raytracing_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
from rtamd import scenes, abi
import orc as oracle
f32=np.float32
L=oracle.lib()
FP=C.POINTER(C.c_float)
L.orc_scene_hit_path.argtypes=[C.POINTER(abi.rt_scene),FP,FP,C.c_float,C.c_float,FP,C.POINTER(C.c_int)]
L.orc_scatter_explicit.argtypes=[C.c_int,C.POINTER(abi.rt_material),FP,FP,FP,C.c_int,C.c_float,C.c_float,C.c_float,FP,FP]
L.orc_rng_draw.restype=C.c_uint32
def fp(a): return a.ctypes.data_as(FP)
full, cam, _ = scenes.scene_stress(n_spheres=600, grid=12, spp=4, seed=3)
s = scenes.Scene()
sph=[p for p in full.prims if p.type==0]
for p in sph: s.sphere(s.diffuse((0.5,0.5,0.5)),(p.f[0],p.f[1],p.f[2]),p.f[3])
sc=s.c()
w,h=96,54
def lowbias32(x):
    x&=0xffffffff; x^=x>>16; x=(x*0x7feb352d)&0xffffffff; x^=x>>15; x=(x*0x846ca68b)&0xffffffff; x^=x>>16; return x
def trace(py,px,seed=3,maxd=8):
    pixel=py*w+px
    pkey=lowbias32(pixel+0x9E3779B9*seed+0x85EBCA6B); skey=lowbias32(pkey^0)
    fov=f32(cam.fov_deg/(180.0/math.pi)); th=f32(math.tan(fov/f32(2)))
    x=f32((f32(2)*(f32(px)+f32(.5))/f32(w)-f32(1))*th*f32(w)/f32(h)); y=f32(-(f32(2)*(f32(py)+f32(.5))/f32(h)-f32(1))*th)
    rot=np.zeros(9,f32); L.orc_rotation_mat(fp(np.array(list(cam.orientation_deg),f32)),fp(rot))
    v=np.array([x,y,-1],f32); d=(rot.reshape(3,3)@v).astype(f32); dd=np.zeros(3,f32); L.orc_normalize(fp(d),fp(dd)); d=dd
    o=np.array(list(cam.position),f32)
    for depth in range(maxd+1):
        out=np.zeros(8,f32); mat=C.c_int(0)
        hit=L.orc_scene_hit_path(C.byref(sc),fp(o),fp(d),0.01,1000.0,fp(out),C.byref(mat))
        if not hit: print(" depth",depth,"miss"); return
        idx=mat.value; p=out[1:4].copy(); n=out[4:7].copy()
        c=np.array(sph[idx].f[0:3],np.float64); r=float(sph[idx].f[3])
        delta=np.linalg.norm(p.astype(np.float64)-c)-r
        print(f" depth {depth} hit sphere {idx} t={out[0]:.6f} front={int(out[7])} dist_from_surface={delta:+.6f} r={r:.3f} |o-c|-r={np.linalg.norm(o.astype(np.float64)-c)-r:+.6f} cos={float(-(d*n).sum()):+.4f}")
        r32=lowbias32(skey+(4*(depth+1)+1)*0x9E3779B9)
        ua=f32(r32&0xffff)/f32(65536); ub=f32(r32>>16)/f32(65536)
        od=np.zeros(3,f32); oa=np.zeros(3,f32)
        L.orc_scatter_explicit(abi.RT_PATH_LIGHTS,C.byref(s.materials[idx]),fp(d),fp(p),fp(n),int(out[7]),float(ua),float(ub),0.5,fp(od),fp(oa))
        o=p; d=od

from rtamd import lib
ctx = lib.Context()
def probe_scene(flip):
    ps = scenes.Scene()
    for i, p in enumerate(sph):
        ps.sphere(ps.light_material((float(i + 1), 0, 0)), (p.f[0], p.f[1], p.f[2]), p.f[3], flip=flip)
    return ps
PS = [probe_scene(False), probe_scene(True)]
def gpu_probe(o, d):
    """index of the sphere the GPU kernel reports for the ray (o, d) and whether it was a front face"""
    ax = math.asin(float(d[1])); cx = math.cos(ax)
    ay = math.atan2(-float(d[0]) / cx, -float(d[2]) / cx)
    camp = scenes.camera(1.0, tuple(float(v) for v in o), (math.degrees(ax), math.degrees(ay), 0.0))
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, 1, max_depth=0, seed=1)
    res = []
    for flip in (0, 1):
        g = ctx.render_path(PS[flip], camp, prm, 1, 1)
        res.append(int(round(float(g["rgb"][0, 0, 0]))) - 1)
    return ("front", res[0]) if res[0] >= 0 else (("back", res[1]) if res[1] >= 0 else ("miss", -1))
_orig_hit = L.orc_scene_hit_path
def trace2(py, px, seed=3, maxd=8):
    pixel = py * w + px
    pkey = lowbias32(pixel + 0x9E3779B9 * seed + 0x85EBCA6B); skey = lowbias32(pkey ^ 0)
    fov = f32(cam.fov_deg / (180.0 / math.pi)); th = f32(math.tan(fov / f32(2)))
    x = f32((f32(2) * (f32(px) + f32(.5)) / f32(w) - f32(1)) * th * f32(w) / f32(h)); y = f32(-(f32(2) * (f32(py) + f32(.5)) / f32(h) - f32(1)) * th)
    rot = np.zeros(9, f32); L.orc_rotation_mat(fp(np.array(list(cam.orientation_deg), f32)), fp(rot))
    v = np.array([x, y, -1], f32); d = (rot.reshape(3, 3) @ v).astype(f32); dd = np.zeros(3, f32); L.orc_normalize(fp(d), fp(dd)); d = dd
    o = np.array(list(cam.position), f32)
    for depth in range(maxd + 1):
        out = np.zeros(8, f32); mat = C.c_int(0)
        hit = L.orc_scene_hit_path(C.byref(sc), fp(o), fp(d), 0.01, 1000.0, fp(out), C.byref(mat))
        print(f" depth {depth}: cpu {'hit %d front=%d t=%.5f' % (mat.value, int(out[7]), out[0]) if hit else 'miss'} | gpu {gpu_probe(o, d)}")
        if not hit: return
        idx = mat.value; p = out[1:4].copy(); n = out[4:7].copy()
        r32 = lowbias32(skey + (4 * (depth + 1) + 1) * 0x9E3779B9)
        ua = f32(r32 & 0xffff) / f32(65536); ub = f32(r32 >> 16) / f32(65536)
        od = np.zeros(3, f32); oa = np.zeros(3, f32)
        L.orc_scatter_explicit(abi.RT_PATH_LIGHTS, C.byref(s.materials[idx]), fp(d), fp(p), fp(n), int(out[7]), float(ua), float(ub), 0.5, fp(od), fp(oa))
        o = p; d = od

import torch
for px in [(4, 39), (4, 60)]:
    print(px); trace2(*px)
