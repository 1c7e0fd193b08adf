"""ctypes binding of librtc_amd.so (the C ABI in include/rtc.h).

There is no CPU fallback: if the shared library is missing the import fails loudly, and every render
entry point fails with RTC_ERR_NO_DEVICE when no HIP device is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTC_AMD_LIB") or os.path.join(_HERE, "librtc_amd.so")  # override: A/B builds only

OK, ERR_NON_INVERTIBLE, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_IO, ERR_PARSE, ERR_OVERFLOW = range(9)
SPHERE, PLANE, CUBE, CYLINDER, CONE = range(5)
CSG_UNION, CSG_INTERSECTION, CSG_DIFFERENCE = range(3)
PAT_STRIPE, PAT_GRADIENT, PAT_RING, PAT_CHECKER, PAT_RADIAL, PAT_SOLID, PAT_BLENDED, PAT_TEST = range(8)

PD = C.POINTER(C.c_double)
D3 = C.c_double * 3
D16 = C.c_double * 16


class RtcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtc error {code}: {message}")
        self.code = code


class NonInvertibleMatrix(RtcError):
    """RayTracerError::NonInvertibleMatrix (ray_tracer/src/error.rs:3-7)."""


class Material(C.Structure):
    _fields_ = [
        ("color", C.c_double * 3),
        ("ambient", C.c_double),
        ("diffuse", C.c_double),
        ("specular", C.c_double),
        ("shininess", C.c_double),
        ("reflectivity", C.c_double),
        ("transparency", C.c_double),
        ("refractive_index", C.c_double),
        ("casts_shadow", C.c_int32),
        ("pattern", C.c_int32),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("depth", C.c_uint32),
        ("tile_rows", C.c_uint32),
        ("tile_index", C.c_uint32),
        ("tile_count", C.c_uint32),
        ("use_bvh", C.c_uint32),
        ("collect_stats", C.c_uint32),
        ("schedule", C.c_uint32),
        ("exchange", C.c_uint32),
        ("max_rays_per_pixel", C.c_uint32),
        ("fp_mode", C.c_uint32),
        ("tail_polls", C.c_uint32),
        ("reserved", C.c_uint32 * 5),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("pixels", C.c_uint64),
        ("primary_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("secondary_rays", C.c_uint64),
        ("secondary_rays_reference", C.c_uint64),
        ("container_passes", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("prim_tests", C.c_uint64 * 7),
        ("xform_levels", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("light_evals", C.c_uint64),
        ("schlick_evals", C.c_uint64),
        ("refract_evals", C.c_uint64),
        ("overflow", C.c_uint64),
        ("shadow_rays_skipped", C.c_uint64),
        ("diag", C.c_uint64 * 10),
        ("kernel_ms", C.c_double),
        ("algorithmic_flops", C.c_double),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "hsize", "vsize", "n_prims", "n_unbounded", "n_bvh_nodes", "n_xforms", "n_materials", "n_lights",
        "n_patterns", "n_triangles", "bvh_max_depth", "bvh_max_leaf", "committed", "device")] + [
        ("device_bytes", C.c_uint64), ("n_devices", C.c_uint32), ("n_heavy_tiles", C.c_uint32)]


class ItemInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("parent", C.c_int32), ("shape_type", C.c_int32), ("material_id", C.c_int32),
                ("n_children", C.c_uint32), ("n_triangles", C.c_uint32), ("closed", C.c_int32), ("smooth", C.c_int32),
                ("minimum", C.c_double), ("maximum", C.c_double), ("transform", C.c_double * 16)]


class LightInfo(C.Structure):
    _fields_ = [("is_area", C.c_int32), ("usteps", C.c_uint32), ("vsteps", C.c_uint32),
                ("position", C.c_double * 3), ("intensity", C.c_double * 3),
                ("corner", C.c_double * 3), ("uvec", C.c_double * 3), ("vvec", C.c_double * 3)]


class ObjInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("ignored_lines", "vertices", "normals", "triangles", "groups")]


# every symbol include/rtc.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("rtc_last_error", C.c_char_p, []),
    ("rtc_version", C.c_char_p, []),
    ("rtc_material_default", None, [C.POINTER(Material)]),
    ("rtc_render_opts_default", None, [C.POINTER(RenderOpts)]),
    ("rtc_matrix_identity", None, [PD]),
    ("rtc_matrix_translation", None, [C.c_double] * 3 + [PD]),
    ("rtc_matrix_scaling", None, [C.c_double] * 3 + [PD]),
    ("rtc_matrix_rotation_x", None, [C.c_double, PD]),
    ("rtc_matrix_rotation_y", None, [C.c_double, PD]),
    ("rtc_matrix_rotation_z", None, [C.c_double, PD]),
    ("rtc_matrix_shearing", None, [C.c_double] * 6 + [PD]),
    ("rtc_matrix_multiply", None, [PD, PD, PD]),
    ("rtc_matrix_inverse", C.c_int, [PD, PD]),
    ("rtc_view_transform", None, [PD, PD, PD, PD]),
    ("rtc_scene_create", C.c_void_p, []),
    ("rtc_scene_destroy", None, [C.c_void_p]),
    ("rtc_scene_set_camera", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, PD]),
    ("rtc_scene_add_point_light", C.c_int, [C.c_void_p, PD, PD]),
    ("rtc_scene_add_area_light", C.c_int, [C.c_void_p, PD, PD, C.c_uint32, PD, C.c_uint32, PD]),
    ("rtc_scene_add_material", C.c_int, [C.c_void_p, C.POINTER(Material), C.POINTER(C.c_uint32)]),
    ("rtc_scene_add_pattern", C.c_int,
     [C.c_void_p, C.c_int, PD, PD, C.c_int32, C.c_int32, PD, C.POINTER(C.c_uint32)]),
    ("rtc_scene_begin_group", C.c_int, [C.c_void_p, PD, C.c_int32]),
    ("rtc_scene_end_group", C.c_int, [C.c_void_p]),
    ("rtc_scene_begin_csg", C.c_int, [C.c_void_p, C.c_int]),
    ("rtc_scene_end_csg", C.c_int, [C.c_void_p]),
    ("rtc_scene_add_shape", C.c_int, [C.c_void_p, C.c_int, PD, C.c_uint32, C.c_double, C.c_double, C.c_int]),
    ("rtc_scene_add_triangles", C.c_int, [C.c_void_p, C.c_size_t, PD, PD, C.c_uint32]),
    ("rtc_scene_commit", C.c_int, [C.c_void_p, C.c_int]),
    ("rtc_scene_commit_devices", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_uint32]),
    ("rtc_scene_check", C.c_int, [C.c_void_p]),
    ("rtc_render", C.c_int, [C.c_void_p, C.POINTER(RenderOpts), PD, C.POINTER(Stats)]),
    ("rtc_render_device", C.c_int, [C.c_void_p, C.POINTER(RenderOpts), C.c_void_p, C.c_void_p]),
    ("rtc_render_device_static", C.c_int, [C.c_void_p, C.POINTER(RenderOpts), C.c_void_p, C.c_void_p]),
    ("rtc_local_rows", C.c_uint32, [C.c_uint32] * 4),
    ("rtc_local_row_map", C.c_int, [C.c_uint32] * 4 + [C.POINTER(C.c_uint32), C.c_uint32]),
    ("rtc_scene_last_stats", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("rtc_scene_get_info", C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    ("rtc_scene_item_count", C.c_uint32, [C.c_void_p]),
    ("rtc_scene_get_item", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(ItemInfo)]),
    ("rtc_scene_child_count", C.c_uint32, [C.c_void_p, C.c_int32]),
    ("rtc_scene_get_child", C.c_int, [C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("rtc_scene_get_material", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Material)]),
    ("rtc_scene_get_triangles", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, PD, PD]),
    ("rtc_scene_get_light", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(LightInfo)]),
    ("rtc_scene_get_camera", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_double), PD]),
    ("rtc_scene_load_yaml", C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("rtc_scene_load_yaml_file", C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("rtc_scene_add_obj", C.c_int, [C.c_void_p, C.c_char_p, PD, C.c_int32, C.POINTER(ObjInfo)]),
    ("rtc_canvas_to_ppm", C.c_int, [PD, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()). "
                "There is no CPU fallback for the render path.")
        L = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(rc):
    if rc == OK:
        return
    msg = lib().rtc_last_error().decode("utf-8", "replace")
    if rc == ERR_NON_INVERTIBLE:
        raise NonInvertibleMatrix(rc, msg)
    raise RtcError(rc, msg)
