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
