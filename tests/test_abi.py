"""The C-ABI library loads and exports every symbol include/rt_mi355.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from rtamd import abi, lib, scenes

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
    assert L.rt_abi_version() == 2


def test_every_exported_symbol_is_declared_in_a_header():
    """The diagnostic exports (rt_debug_*) are declared in include/rt_mi355_debug.h, apart from the boundary: every rt_* symbol
    the library exports has a declaration in one of the two headers, and every declared one is exported."""
    import subprocess
    text = open(os.path.join(ROOT, "include", "rt_mi355_debug.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    dbg = sorted(set(re.findall(r"\b(rt_debug_[a-z_]+)\s*\(", text)))
    assert dbg and all(n.startswith("rt_debug_") for n in dbg)
    L = lib.load()
    for name in dbg:
        assert hasattr(L, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("rt_") and ln.split()[-2] in "TW"})
    assert exported == sorted(set(dbg) | set(_declared())), set(exported) ^ (set(dbg) | set(_declared()))


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
            if f.endswith((".hip", ".h", ".cpp", ".hpp", ".py", ".inc")):
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
    # the 02 kernel (variant 2): its shadow samples use the same generated query
    s2 = scenes.scene02(16)[0]
    s2.triangle(0, (-1, 0, -4), (1, 0, -4), (0, 1, -4))
    sc = s2.c(); note = C.create_string_buffer(8192)
    rc = L.rt_debug_jit_probe(C.byref(sc), 2, note, 8192)
    assert rc in (0, 1) and b"compile failed" not in note.value and b"hiprtcCreateProgram" not in note.value, note.value.decode()


def test_bvh_builder_covers_every_bounded_slot_exactly_once():
    """Host logic of the optional BVH (rt_bvh.h), no GPU: rt_debug_bvh_probe builds the tree and checks that every bounded
    slot sits in exactly one leaf whose box contains it, child boxes nest, planes stay outside the tree; here on the stress
    scene, the reference scenes, boxes (six rectangle slots each), a single primitive, planes only and an empty scene."""
    L = lib.load()
    L.rt_debug_bvh_probe.argtypes = [C.POINTER(abi.rt_scene), C.c_int32, C.POINTER(C.c_int32)]

    def probe(s):
        sc = s.c(); out = (C.c_int32 * 5)()
        assert L.rt_debug_bvh_probe(C.byref(sc), 4, out) == 0
        return dict(nodes=out[0], refs=out[1], unbounded=out[2], depth=out[3], slots=out[4])

    r = probe(scenes.scene_stress(spp=1)[0])
    assert r["slots"] == 10000 + 1 + 8192 + 1 and r["unbounded"] == 1 and r["refs"] == r["slots"] - 1
    assert r["refs"] / 4 <= r["nodes"] + 1 <= r["refs"] and r["depth"] <= 24
    r = probe(scenes.scene04(4)[0]); assert (r["refs"], r["unbounded"], r["nodes"]) == (4, 5, 1)
    s = scenes.Scene(); w = s.diffuse((0.5, 0.5, 0.5))
    r = probe(s); assert (r["refs"], r["unbounded"], r["nodes"]) == (0, 0, 0)
    s.plane(w, (0, 0, 0), (0, 1, 0)); s.plane(w, (0, 0, -8), (0.1, 0.2, 1.0))
    r = probe(s); assert (r["refs"], r["unbounded"], r["nodes"]) == (0, 2, 0)
    s.sphere(w, (0, 1, -3), 1.0)
    r = probe(s); assert (r["refs"], r["unbounded"], r["nodes"], r["depth"]) == (1, 2, 1, 1)
    for k in range(7):
        s.box(w, (3 * k, 0, -5), (1, 2, 3), (10 * k, 20, 5 * k)); s.triangle(w, (k, 0, 0), (k + 1, 0, 0), (k, 1, -k))
    r = probe(s); assert r["refs"] == 1 + 7 * 6 + 7 and r["unbounded"] == 2
    # many coincident primitives: no spatial split exists, the object-median fallback must still terminate
    s = scenes.Scene(); w = s.diffuse((0.5, 0.5, 0.5))
    for k in range(300):
        s.sphere(w, (1, 2, 3), 0.5)
    r = probe(s); assert r["refs"] == 300 and r["depth"] <= 10


def test_device_group_partition_covers_every_sample_once():
    """rt_create_multi cuts [sample_begin, sample_end) into one contiguous share per device: no gap, no overlap, sizes within one."""
    L = lib.load()
    b, e = C.c_int32(), C.c_int32()
    for (s0, s1) in ((0, 1024), (0, 1), (3, 10), (5, 5), (0, 1000)):
        for n in (1, 2, 3, 4, 8):
            cur, sizes = s0, []
            for i in range(n):
                assert L.rt_debug_group_share(s0, s1, i, n, C.byref(b), C.byref(e)) == 0
                assert b.value == cur and e.value >= b.value
                cur = e.value; sizes.append(e.value - b.value)
            assert cur == s1 and max(sizes) - min(sizes) <= 1
    assert L.rt_debug_group_share(0, 4, 2, 2, C.byref(b), C.byref(e)) == abi.RT_ERR_BAD_ARG


def test_device_group_argument_errors_need_no_gpu():
    """rt_create_multi refuses a null list and counts outside [1, 64] before it touches any device, with a message."""
    L = lib.load()
    ids = (C.c_int32 * 70)(*range(70))
    for args in ((None, 2), (ids, 0), (ids, -1), (ids, 65)):
        assert not L.rt_create_multi(args[0], args[1])
        assert b"rt_create_multi" in L.rt_last_error(None)


def test_prefer_system_compiler_is_safe_to_call_any_time():
    """rtamd.lib.prefer_system_compiler(): conftest.py has called it before anything imported torch (JIT_COMPILER says which
    compiler library serves the process); calling it again, or after torch is in, must neither raise nor load anything new."""
    import conftest
    from rtamd import lib as L
    assert isinstance(conftest.JIT_COMPILER, str) and conftest.JIT_COMPILER
    if os.path.exists(L.SYSTEM_COMGR) and os.environ.get("RT_MI355_SYSTEM_COMGR", "1") != "0":
        assert L.SYSTEM_COMGR in conftest.JIT_COMPILER
    import torch  # noqa: F401
    assert "PyTorch" in L.prefer_system_compiler()


def test_device_code_is_built_with_pragma_honouring_contraction():
    """-ffp-contract=fast lets the backend fuse multiply-adds inside rt_strict.h's '#pragma clang fp contract(off)' arithmetic (found in round 3: 01 then
    differs from the reference-made frame by up to 2.4e-4 instead of 2e-7).  Both builds - the Makefile's and the hiprtc options - must say fast-honor-pragmas."""
    import re
    mk = open(os.path.join(ROOT, "software-raytracing_amd", "Makefile")).read()
    jit = open(os.path.join(ROOT, "software-raytracing_amd", "csrc", "rt_jit.hip")).read()
    flags = [l for l in mk.splitlines() if l.startswith("FLAGS")][0]
    assert "-ffp-contract=fast-honor-pragmas" in flags and not re.search(r"-ffp-contract=fast(\s|$)", flags)
    opts = [l for l in jit.splitlines() if "const char* opts[]" in l][0]
    assert '"-ffp-contract=fast-honor-pragmas"' in opts and '"-ffp-contract=fast"' not in opts and "-fhip-fp32-correctly-rounded-divide-sqrt" in opts
