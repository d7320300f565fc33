"""ctypes binding of librt_mi355.so (the product).  Mirrors include/rt_mi355.h one to one; there is
no CPU fallback: a missing library or a missing gfx950 device raises."""
import ctypes as C
import os
import sys

import numpy as np

from . import abi

_HERE = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB_PATH = os.environ.get("RT_MI355_LIB") or os.path.join(_HERE, "librt_mi355.so")   # override: A/B builds in tools/
_LIB = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt_mi355 error {code}: {msg}")
        self.code = code


COMGR_SONAME_MAJOR = 3       # libamd_comgr.so.3: the ABI the HIP runtime PyTorch bundles was linked against
SYSTEM_COMGR = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", f"libamd_comgr.so.{COMGR_SONAME_MAJOR}")


def prefer_system_compiler():
    """Call BEFORE `import torch`.  The scene-specialised kernels are compiled at run time through hiprtc -> libamd_comgr of
    whichever ROCm runtime the process loaded first; PyTorch bundles an older one (ROCm 7.0) than the image ships (7.2), and the
    older compiler's kernel is 3 % slower (tools/runtime_probe.py).  Loading the system's compiler library first, with global
    visibility, makes it the provider of amd_comgr_* for the whole process (same soname, same ABI).  It is only taken when it
    reports the soname's major version through amd_comgr_get_version (an ABI check, not a bet on the file name); otherwise, and
    with RT_MI355_SYSTEM_COMGR=0, the process keeps whatever compiler its HIP runtime brings (the GPU suite passes either way:
    tests/test_gpu_bundled_compiler.py).  A host that never imports torch gets the system compiler anyway.
    -> a description of the compiler in use."""
    if os.environ.get("RT_MI355_SYSTEM_COMGR", "1") != "0" and "torch" not in sys.modules:
        try:
            probe = C.CDLL(SYSTEM_COMGR)          # local first: nothing binds to it unless the version check passes
            major, minor = C.c_size_t(0), C.c_size_t(0)
            probe.amd_comgr_get_version.restype = None
            probe.amd_comgr_get_version(C.byref(major), C.byref(minor))
            if major.value != COMGR_SONAME_MAJOR:
                return f"the ROCm runtime the process loads ({SYSTEM_COMGR} reports comgr {major.value}.{minor.value}, not {COMGR_SONAME_MAJOR}.x: not used)"
            C.CDLL(SYSTEM_COMGR, mode=C.RTLD_GLOBAL)
            return f"{SYSTEM_COMGR} (comgr {major.value}.{minor.value}, loaded before torch)"
        except (OSError, AttributeError):
            pass
    return "the ROCm runtime bundled with PyTorch" if "torch" in sys.modules else "the first ROCm runtime the process loads"


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C software-raytracing_amd` "
                               "(or __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        L.rt_create.restype = C.c_void_p
        L.rt_create.argtypes = [C.c_int32]
        L.rt_create_multi.restype = C.c_void_p
        L.rt_create_multi.argtypes = [P(C.c_int32), C.c_int32]
        L.rt_destroy.argtypes = [C.c_void_p]
        L.rt_last_error.restype = C.c_char_p
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_get_stats.argtypes = [C.c_void_p, P(abi.rt_stats)]
        L.rt_whitted_params_default.argtypes = [C.c_int32, P(abi.rt_whitted_params)]
        L.rt_path_params_default.argtypes = [C.c_int32, P(abi.rt_path_params)]
        L.rt_render_whitted.argtypes = [C.c_void_p, P(abi.rt_scene), P(abi.rt_camera), P(abi.rt_whitted_params), P(abi.rt_image)]
        L.rt_render_path.argtypes = [C.c_void_p, P(abi.rt_scene), P(abi.rt_camera), P(abi.rt_path_params), P(abi.rt_image)]
        L.rt_progressive_begin.argtypes = [C.c_void_p, P(abi.rt_scene), P(abi.rt_camera), P(abi.rt_path_params), C.c_int32, C.c_int32]
        L.rt_progressive_pass.argtypes = [C.c_void_p, C.c_int32, P(abi.rt_image), P(C.c_int32)]
        L.rt_progressive_end.argtypes = [C.c_void_p]
        L.rt_path_accumulate_device.argtypes = [C.c_void_p, P(abi.rt_scene), P(abi.rt_camera), P(abi.rt_path_params),
                                                C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rt_path_accumulate_device_noalloc.argtypes = L.rt_path_accumulate_device.argtypes
        L.rt_noalloc_limits.argtypes = [P(C.c_int32), P(C.c_int32)]
        L.rt_whitted_device.argtypes = [C.c_void_p, P(abi.rt_scene), P(abi.rt_camera), P(abi.rt_whitted_params),
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rt_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.rt_sync.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.rt_debug_set_regen_threshold.argtypes = [C.c_void_p, C.c_int32]
        L.rt_debug_set_flags.argtypes = [C.c_void_p, C.c_int32]
        L.rt_debug_last_path_kernel.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        L.rt_debug_last_whitted_kernel.argtypes = [C.c_void_p]
        L.rt_debug_last_whitted_form.argtypes = [C.c_void_p]
        L.rt_debug_last_path_form.argtypes = [C.c_void_p, P(C.c_int32)]
        _LIB = L
    return _LIB


class Context:
    """rt_ctx owner.  One per process/device (the four Render wrappers share one)."""

    def __init__(self, device_id=-1, devices=None):
        """devices: a list of HIP device ids -> a device group (rt_create_multi); else one device"""
        self.L = load()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            self.h = self.L.rt_create_multi(arr, len(devices))
        else:
            self.h = self.L.rt_create(device_id)
        if not self.h:
            raise RtError(abi.RT_ERR_NO_DEVICE, self.L.rt_last_error(None).decode())

    def close(self):
        if self.h:
            self.L.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, self.L.rt_last_error(self.h).decode())

    def stats(self):
        st = abi.rt_stats()
        self._chk(self.L.rt_get_stats(self.h, C.byref(st)))
        return dict(rays=st.rays, shadow_rays=st.shadow_rays, samples=st.samples, kernel_ms=st.kernel_ms, total_ms=st.total_ms, allreduce_ms=st.allreduce_ms)

    # ---- host-buffer entry points ---------------------------------------------------------
    def _image(self, width, height, want_rgb=True):
        bgra = np.zeros((height, width, 4), np.uint8)
        rgb = np.zeros((height, width, 3), np.float32) if want_rgb else None
        img = abi.rt_image(bgra.ctypes.data, rgb.ctypes.data if want_rgb else None, width, height, 0)
        return img, bgra, rgb

    def render_whitted(self, scene, cam, prm, width, height):
        img, bgra, rgb = self._image(width, height)
        sc = scene.c()
        self._chk(self.L.rt_render_whitted(self.h, C.byref(sc), C.byref(cam), C.byref(prm), C.byref(img)))
        return dict(bgra=bgra, rgb=rgb, **self.stats())

    def render_path(self, scene, cam, prm, width, height):
        img, bgra, rgb = self._image(width, height)
        sc = scene.c()
        self._chk(self.L.rt_render_path(self.h, C.byref(sc), C.byref(cam), C.byref(prm), C.byref(img)))
        return dict(bgra=bgra, rgb=rgb, **self.stats())

    # ---- progressive frame: sums stay on the device, one preview per pass --------------------
    def progressive_begin(self, scene, cam, prm, width, height):
        sc = scene.c()
        self._chk(self.L.rt_progressive_begin(self.h, C.byref(sc), C.byref(cam), C.byref(prm), width, height))
        self._prog_size = (width, height)

    def progressive_pass(self, n_samples, preview=True, want_rgb=True):
        """-> dict(done=..., bgra=..., rgb=..., **stats of the pass); preview=False adds samples without reading a frame back"""
        done = C.c_int32(0)
        if preview:
            img, bgra, rgb = self._image(*self._prog_size, want_rgb=want_rgb)
            self._chk(self.L.rt_progressive_pass(self.h, n_samples, C.byref(img), C.byref(done)))
        else:
            bgra = rgb = None
            self._chk(self.L.rt_progressive_pass(self.h, n_samples, None, C.byref(done)))
        return dict(done=done.value, bgra=bgra, rgb=rgb, **self.stats())

    def progressive_end(self):
        self._chk(self.L.rt_progressive_end(self.h))

    # ---- device-buffer entry points (raw device pointers, e.g. torch tensor .data_ptr()) -----
    def path_accumulate_device(self, scene, cam, prm, width, height, d_accum_ptr, stream=None):
        sc = scene.c()
        self._chk(self.L.rt_path_accumulate_device(self.h, C.byref(sc), C.byref(cam), C.byref(prm), width, height, d_accum_ptr, stream))

    def path_accumulate_device_noalloc(self, scene, cam, prm, width, height, d_accum_ptr, stream=None):
        sc = scene.c()
        self._chk(self.L.rt_path_accumulate_device_noalloc(self.h, C.byref(sc), C.byref(cam), C.byref(prm), width, height, d_accum_ptr, stream))

    def whitted_device(self, scene, cam, prm, width, height, d_color_ptr, stream=None):
        sc = scene.c()
        self._chk(self.L.rt_whitted_device(self.h, C.byref(sc), C.byref(cam), C.byref(prm), width, height, d_color_ptr, stream))

    def resolve_device(self, d_accum_ptr, width, height, divisor, gamma, d_bgra_ptr, stride=0, d_rgb_ptr=None, stream=None):
        self._chk(self.L.rt_resolve_device(self.h, d_accum_ptr, width, height, divisor, int(gamma), d_bgra_ptr, stride, d_rgb_ptr, stream))

    def sync(self, stream=None):
        self._chk(self.L.rt_sync(self.h, stream))
        return self.stats()

    def set_option(self, option, value):
        self._chk(self.L.rt_set_option(self.h, option, value))

    def set_regen_threshold(self, n):
        self._chk(self.L.rt_debug_set_regen_threshold(self.h, n))

    def last_path_kernel(self):
        """(kind, note): 2 scene-specialised (hiprtc), 1 generic LDS-resident, 0 tiled / global; note = why not specialised"""
        buf = C.create_string_buffer(2048)
        kind = self.L.rt_debug_last_path_kernel(self.h, buf, 2048)
        return kind, buf.value.decode(errors="replace")

    def last_path_form(self):
        """(form, unit_log2): 0 one lane per pixel (per-lane regeneration), 1 queue form, 2 item form (persistent waves, work counter)"""
        k = C.c_int32(0)
        return self.L.rt_debug_last_path_form(self.h, C.byref(k)), k.value

    def last_whitted_form(self):
        """1 wavefront form (walk / shadow rays / combine), 0 monolithic kernel"""
        return self.L.rt_debug_last_whitted_form(self.h)

    def last_whitted_kernel(self):
        """2 scene-specialised (02 only), 1 generic"""
        return self.L.rt_debug_last_whitted_kernel(self.h)

    def set_debug_flags(self, f):
        self._chk(self.L.rt_debug_set_flags(self.h, f))
