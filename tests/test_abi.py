"""The C-ABI library loads and exports every symbol include/rt_mi355.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from rtamd import abi, lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    text = open(os.path.join(ROOT, "include", "rt_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = lib.load()
    for name in _declared():
        assert hasattr(L, name), name
    assert L.rt_abi_version() == 1


def test_record_sizes_match_header():
    assert C.sizeof(abi.rt_prim) == 64 and C.sizeof(abi.rt_material) == 48 and C.sizeof(abi.rt_light) == 32
    assert C.sizeof(abi.rt_path_params) == 32 and C.sizeof(abi.rt_whitted_params) == 16


def test_param_defaults_are_the_examples_constants():
    L = lib.load()
    w = abi.rt_whitted_params()
    assert L.rt_whitted_params_default(2, C.byref(w)) == 0 and (w.max_depth, w.shadow_samples) == (4, 16)
    assert L.rt_whitted_params_default(1, C.byref(w)) == 0 and (w.max_depth, w.shadow_samples) == (4, 0)
    p = abi.rt_path_params()
    assert L.rt_path_params_default(3, C.byref(p)) == 0 and (p.spp, p.max_depth) == (16, 24) and abs(p.t_min - 0.001) < 1e-9
    assert L.rt_path_params_default(4, C.byref(p)) == 0 and (p.spp, p.max_depth) == (32, 6) and abs(p.t_min - 0.01) < 1e-9
    assert L.rt_path_params_default(7, C.byref(p)) == abi.RT_ERR_BAD_ARG


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.RtError):
        lib.Context()


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "software-raytracing_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".hpp")) or f in ("lib.py", "abi.py", "scenes.py"):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "librt_oracle" not in text and "orc_" not in text, os.path.join(dirpath, f)


def test_scene_specialised_kernel_source_compiles_for_every_primitive_type():
    """rt_jit.hip generates the scene query of a small scene as literal straight-line code and compiles it with hiprtc.
    The compile needs no GPU (only loading the module does), so the generated source is checked here for every primitive
    type and both path variants; the images are compared with the generic kernel in the GPU tests."""
    import ctypes as C
    from rtamd import abi, lib, scenes
    L = lib.load()
    L.rt_debug_jit_probe.argtypes = [C.POINTER(abi.rt_scene), C.c_int32, C.c_char_p, C.c_int32]
    s = scenes.Scene()
    d = s.diffuse((0.5, 0.5, 0.5)); m = s.metal((0.8, 0.8, 0.8), 0.2); g = s.dielectric(0.6); lm = s.light_material((5, 5, 5))
    s.plane(d, (0, -3, 0), (0, 1, 0)); s.plane(d, (0, 0, -12), (0.2, 0.1, 1.0)); s.sphere(g, (0, 0, -5), 1.0); s.sphere(m, (1, 0, -5), 0.5, flip=True)
    s.triangle(m, (-6, -3, -9), (-1, -3, -11), (-4, 3, -10)); s.rect(lm, (0, 5, -6), (4, 3), (70, 20, 0)); s.box(d, (3, -2, -7), (2, 2, 2), (0, 30, 10)); s.sphere(-1, (3, 1, -7), 0.5)
    for scene, variant in ((s, 4), (scenes.scene04(1)[0], 4), (scenes.scene03(1)[0], 3), (scenes.scene03(1, readme_variant=True)[0], 3)):
        note = C.create_string_buffer(8192)
        sc = scene.c()
        rc = L.rt_debug_jit_probe(C.byref(sc), variant, note, 8192)
        text = note.value.decode(errors="replace")
        assert rc in (0, 1) and "compile failed" not in text and "hiprtcCreateProgram" not in text, text
    # beyond 64 primitives the generic kernels are used
    big = scenes.scene_stress(n_spheres=80, grid=2, spp=1)[0]
    note = C.create_string_buffer(1024); sc = big.c()
    assert L.rt_debug_jit_probe(C.byref(sc), 4, note, 1024) == 0 and b"specialised up to 64" in note.value
