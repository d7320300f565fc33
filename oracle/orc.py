"""orc - ctypes wrapper of oracle/librt_oracle.so, the CPU restatement.  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path (it lives outside the
product package for that reason; it borrows the package's ctypes record definitions, rtamd.abi)."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(_ROOT, "software-raytracing_amd"))
from rtamd import abi  # noqa: E402

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, "oracle", "librt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        fp, u8p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
        L.orc_render_whitted.argtypes = [C.POINTER(abi.rt_scene), C.POINTER(abi.rt_camera), C.POINTER(abi.rt_whitted_params),
                                         C.c_int, C.c_int, C.c_int, C.c_int, fp, u8p, u64p]
        L.orc_render_path.argtypes = [C.POINTER(abi.rt_scene), C.POINTER(abi.rt_camera), C.POINTER(abi.rt_path_params),
                                      C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, u8p, u64p]
        L.orc_rng_draw.restype = C.c_uint32
        L.orc_rng_draw.argtypes = [C.c_uint32] * 4
        L.orc_resolve.argtypes = [fp, C.c_int, C.c_float, C.c_int, u8p]
        L.orc_reflectance.restype = C.c_float
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _run_rows(fn, height, threads):
    if threads <= 1:
        fn(0, height)
        return
    bounds = np.linspace(0, height, threads * 4 + 1).astype(int)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda i: fn(int(bounds[i]), int(bounds[i + 1])), range(len(bounds) - 1)))


def render_whitted(scene, cam, prm, width, height, threads=1):
    """-> dict(rgb float32 [H,W,3], bgra uint8 [H,W,4], rays, shadow_rays)"""
    L = lib()
    rgb = np.zeros((height, width, 3), np.float32)
    bgra = np.zeros((height, width, 4), np.uint8)
    sc = scene.c()

    def rows(y0, y1):
        cnt = (C.c_uint64 * 2)(0, 0)
        rc = L.orc_render_whitted(C.byref(sc), C.byref(cam), C.byref(prm), width, height, y0, y1,
                                  _fp(rgb), bgra.ctypes.data_as(C.POINTER(C.c_uint8)), cnt)
        assert rc == 0
        return cnt[0], cnt[1]

    totals = []
    _run_rows(lambda a, b: totals.append(rows(a, b)), height, threads)
    return dict(rgb=rgb, bgra=bgra, rays=sum(t[0] for t in totals), shadow_rays=sum(t[1] for t in totals))


def render_path(scene, cam, prm, width, height, threads=1, want_sumsq=False):
    """-> dict(sum float32 [H,W,3] per-pixel SUM over the rendered samples, sumsq, bgra, rays)"""
    L = lib()
    s = np.zeros((height, width, 3), np.float32)
    s2 = np.zeros((height, width, 3), np.float32) if want_sumsq else None
    bgra = np.zeros((height, width, 4), np.uint8)
    sc = scene.c()
    totals = []

    def rows(y0, y1):
        cnt = (C.c_uint64 * 2)(0, 0)
        rc = L.orc_render_path(C.byref(sc), C.byref(cam), C.byref(prm), width, height, y0, y1,
                               _fp(s), _fp(s2), bgra.ctypes.data_as(C.POINTER(C.c_uint8)), cnt)
        assert rc == 0
        totals.append(cnt[0])

    _run_rows(rows, height, threads)
    return dict(sum=s, sumsq=s2, bgra=bgra, rays=sum(totals))


def render_path_px(scene, cam, prm, width, height, threads=1):
    """-> dict(sum, rays, px_rays uint32 [H,W]: scene queries per pixel) - debugging aid for parity tests"""
    L = lib()
    L.orc_render_path_px.argtypes = [C.POINTER(abi.rt_scene), C.POINTER(abi.rt_camera), C.POINTER(abi.rt_path_params),
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    s = np.zeros((height, width, 3), np.float32)
    px = np.zeros((height, width), np.uint32)
    sc = scene.c()
    totals = []

    def rows(y0, y1):
        cnt = (C.c_uint64 * 2)(0, 0)
        rc = L.orc_render_path_px(C.byref(sc), C.byref(cam), C.byref(prm), width, height, y0, y1, _fp(s), cnt, px.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert rc == 0
        totals.append(cnt[0])

    _run_rows(rows, height, threads)
    return dict(sum=s, rays=sum(totals), px_rays=px)


def resolve(rgb, divisor, gamma):
    L = lib()
    rgb = np.ascontiguousarray(rgb, np.float32)
    n = rgb.size // 3
    out = np.zeros((n, 4), np.uint8)
    L.orc_resolve(_fp(rgb), n, divisor, int(gamma), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(rgb.shape[:-1] + (4,))
