"""ctypes mirror of include/rt_mi355.h (records and constants only)."""
import ctypes as C

RT_OK = 0
RT_ERR_BAD_ARG = -1
RT_ERR_UNSUPPORTED = -2
RT_ERR_HIP = -3
RT_ERR_NO_DEVICE = -4
RT_ERR_TOO_LARGE = -5

RT_PRIM_SPHERE, RT_PRIM_PLANE, RT_PRIM_RECT, RT_PRIM_TRIANGLE, RT_PRIM_BOX = 0, 1, 2, 3, 4
RT_OPT_ACCEL, RT_ACCEL_NONE, RT_ACCEL_BVH = 1, 0, 1
RT_OPT_ARITHMETIC, RT_ARITH_FAST, RT_ARITH_REFERENCE = 2, 0, 1
RT_PRIM_FLAG_FLIP_NORMALS = 1
RT_MAT_PHONG, RT_MAT_DIFFUSE, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_LIGHT = 0, 1, 2, 3, 4
RT_PATH_SKY, RT_PATH_LIGHTS = 3, 4


class rt_prim(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("flags", C.c_uint32),
                ("reserved", C.c_int32), ("f", C.c_float * 12)]


class rt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("albedo", C.c_float * 3), ("specular_intensity", C.c_float),
                ("shininess", C.c_float), ("primary_to_secondary", C.c_float),
                ("reflect_to_refract", C.c_float), ("refraction_eta", C.c_float),
                ("roughness", C.c_float), ("refraction_index", C.c_float), ("reserved", C.c_float)]


class rt_light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3), ("radius", C.c_float),
                ("reserved", C.c_float)]


class rt_scene(C.Structure):
    _fields_ = [("prims", C.POINTER(rt_prim)), ("n_prims", C.c_int32),
                ("materials", C.POINTER(rt_material)), ("n_materials", C.c_int32),
                ("lights", C.POINTER(rt_light)), ("n_lights", C.c_int32)]


class rt_camera(C.Structure):
    _fields_ = [("fov_deg", C.c_float), ("position", C.c_float * 3), ("orientation_deg", C.c_float * 3)]


class rt_image(C.Structure):
    _fields_ = [("bgra", C.c_void_p), ("rgb", C.c_void_p), ("width", C.c_int32),
                ("height", C.c_int32), ("bgra_stride_bytes", C.c_int32)]


class rt_whitted_params(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("shadow_samples", C.c_int32), ("seed", C.c_uint32),
                ("reserved", C.c_uint32)]


class rt_path_params(C.Structure):
    _fields_ = [("variant", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint32), ("sample_begin", C.c_int32), ("sample_end", C.c_int32),
                ("t_min", C.c_float), ("t_max", C.c_float)]


class rt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("samples", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("allreduce_ms", C.c_double)]


assert C.sizeof(rt_prim) == 64 and C.sizeof(rt_material) == 48 and C.sizeof(rt_light) == 32

# every symbol include/rt_mi355.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "rt_create", "rt_create_multi", "rt_destroy", "rt_last_error", "rt_get_stats", "rt_abi_version",
    "rt_whitted_params_default", "rt_path_params_default",
    "rt_render_whitted", "rt_render_path", "rt_progressive_begin", "rt_progressive_pass", "rt_progressive_end",
    "rt_path_accumulate_device", "rt_path_accumulate_device_noalloc", "rt_noalloc_limits", "rt_whitted_device", "rt_resolve_device", "rt_sync", "rt_set_option",
]
