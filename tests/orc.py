"""ctypes binding for the CPU oracle (oracle/liborc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this module.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liborc_oracle.so")

SPHERE, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, SMOOTH_TRIANGLE, GROUP, TEST_SHAPE, CSG = range(10)
CSG_UNION, CSG_INTERSECTION, CSG_DIFFERENCE = range(3)
PAT_STRIPE, PAT_GRADIENT, PAT_RING, PAT_CHECKER, PAT_RADIAL, PAT_SOLID, PAT_BLENDED, PAT_TEST = range(8)
OK, ERR_NON_INVERTIBLE, ERR_INVALID = 0, 1, 2
COMPS_LEN = 27
INF = float("inf")


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


class Stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("secondary_rays", C.c_uint64),
        ("seconds", C.c_double),
        ("diverged_pixels", C.c_uint64),
    ]


def build(force=False):
    """Compile the oracle if the shared object is missing or stale."""
    src = os.path.join(ORACLE_DIR, "rtc_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "rtc_oracle.h")
    if (
        force
        or not os.path.exists(LIB_PATH)
        or (os.path.exists(src) and os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None

D3 = C.c_double * 3
D16 = C.c_double * 16
PD = C.POINTER(C.c_double)
PI32 = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    L.orc_world_new.restype = C.c_void_p
    L.orc_world_free.argtypes = [C.c_void_p]
    L.orc_add_material.argtypes = [C.c_void_p, C.POINTER(Material)]
    L.orc_add_pattern.argtypes = [C.c_void_p, C.c_int, PD, PD, C.c_int, C.c_int, PD]
    L.orc_add_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, PD, C.c_int, C.c_double, C.c_double, C.c_int]
    L.orc_add_triangle.argtypes = [C.c_void_p, C.c_int, PD, PD, PD, PD, PD, PD, C.c_int]
    L.orc_add_point_light.argtypes = [C.c_void_p, PD, PD]
    L.orc_add_area_light.argtypes = [C.c_void_p, PD, PD, C.c_uint32, PD, C.c_uint32, PD]
    L.orc_set_camera.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, PD]
    L.orc_shape_set_material.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_add_csg.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_csg_intersection_allowed.argtypes = [C.c_int] * 4
    L.orc_csg_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, PD, PI32, PI32]
    L.orc_shape_includes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_render.argtypes = [C.c_void_p, C.c_int, PD, C.POINTER(Stats)]
    L.orc_render_subset.argtypes = [
        C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, PD, C.POINTER(Stats)]
    L.orc_render_pixels.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_uint32), PD, C.POINTER(Stats)]
    L.orc_camera_params.argtypes = [C.c_void_p, PD]
    L.orc_ray_for_pixel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, PD, PD]
    L.orc_intersect.argtypes = [C.c_void_p, C.c_int, PD, PD, C.c_int, PD, PI32, PD, PD]
    L.orc_local_intersect.argtypes = [C.c_void_p, C.c_int, PD, PD, C.c_int, PD]
    L.orc_local_normal_at.argtypes = [C.c_void_p, C.c_int, PD, C.c_double, C.c_double, PD]
    L.orc_normal_at.argtypes = [C.c_void_p, C.c_int, PD, C.c_double, C.c_double, PD]
    L.orc_world_to_object.argtypes = [C.c_void_p, C.c_int, PD, PD]
    L.orc_normal_to_world.argtypes = [C.c_void_p, C.c_int, PD, PD]
    L.orc_color_at.argtypes = [C.c_void_p, PD, PD, C.c_int, PD]
    L.orc_is_shadowed.argtypes = [C.c_void_p, PD, PD, PI32]
    L.orc_light_intensity_at.argtypes = [C.c_void_p, C.c_int, PD, PD]
    L.orc_area_light_point.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, PD]
    L.orc_light_info.argtypes = [C.c_void_p, C.c_int, PD, PD, PD, C.POINTER(C.c_uint32)]
    L.orc_lighting.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, PD, PD, PD, C.c_double, PD]
    L.orc_pattern_at.argtypes = [C.c_void_p, C.c_int, PD, PD]
    L.orc_pattern_at_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, PD, PD]
    L.orc_prepare_computations.argtypes = [C.c_void_p, PD, PD, C.c_int, PD, PI32, PD, PD, C.c_int, PD]
    L.orc_hit_index.argtypes = [C.c_int, PD]
    L.orc_shade_hit.argtypes = [C.c_void_p, PD, C.c_int, PD]
    L.orc_reflected_color.argtypes = [C.c_void_p, PD, C.c_int, PD]
    L.orc_refracted_color.argtypes = [C.c_void_p, PD, C.c_int, PD]
    L.orc_schlick.argtypes = [PD]
    L.orc_schlick.restype = C.c_double
    L.orc_scale_and_clamp.argtypes = [C.c_double]
    L.orc_scale_and_clamp.restype = C.c_uint8
    L.orc_canvas_to_ppm.argtypes = [PD, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    L.orc_canvas_to_ppm.restype = C.c_size_t
    L.orc_matrix_determinant.argtypes = [PD]
    L.orc_matrix_determinant.restype = C.c_double
    for name in ("identity",):
        getattr(L, "orc_matrix_" + name).argtypes = [PD]
    L.orc_matrix_mul.argtypes = [PD, PD, PD]
    L.orc_matrix_inverse.argtypes = [PD, PD]
    L.orc_matrix_translation.argtypes = [C.c_double] * 3 + [PD]
    L.orc_matrix_scaling.argtypes = [C.c_double] * 3 + [PD]
    L.orc_matrix_rotation_x.argtypes = [C.c_double, PD]
    L.orc_matrix_rotation_y.argtypes = [C.c_double, PD]
    L.orc_matrix_rotation_z.argtypes = [C.c_double, PD]
    L.orc_matrix_shearing.argtypes = [C.c_double] * 6 + [PD]
    L.orc_matrix_mul_point.argtypes = [PD, PD, PD]
    L.orc_matrix_mul_vector.argtypes = [PD, PD, PD]
    L.orc_view_transform.argtypes = [PD, PD, PD, PD]
    _lib = L
    return L


def _d3(v):
    return D3(*[float(x) for x in v])


def _d16(m):
    return D16(*[float(x) for x in np.asarray(m, dtype=np.float64).reshape(16)])


def _arr(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(PD)


# ---- matrix helpers (4x4 numpy row-major) -------------------------------------------------

def _mat_out(fn, *args):
    out = D16()
    fn(*args, out)
    return np.array(out[:], dtype=np.float64).reshape(4, 4)


def identity():
    return _mat_out(lib().orc_matrix_identity)


def translation(x, y, z):
    return _mat_out(lib().orc_matrix_translation, x, y, z)


def scaling(x, y, z):
    return _mat_out(lib().orc_matrix_scaling, x, y, z)


def rotation_x(r):
    return _mat_out(lib().orc_matrix_rotation_x, r)


def rotation_y(r):
    return _mat_out(lib().orc_matrix_rotation_y, r)


def rotation_z(r):
    return _mat_out(lib().orc_matrix_rotation_z, r)


def shearing(*a):
    return _mat_out(lib().orc_matrix_shearing, *a)


def matmul(a, b):
    return _mat_out(lib().orc_matrix_mul, _d16(a), _d16(b))


def determinant(a):
    return lib().orc_matrix_determinant(_d16(a))


def inverse(a):
    out = D16()
    rc = lib().orc_matrix_inverse(_d16(a), out)
    if rc != OK:
        raise ValueError("NonInvertibleMatrix")
    return np.array(out[:], dtype=np.float64).reshape(4, 4)


def mul_point(m, p):
    out = D3()
    lib().orc_matrix_mul_point(_d16(m), _d3(p), out)
    return np.array(out[:])


def mul_vector(m, v):
    out = D3()
    lib().orc_matrix_mul_vector(_d16(m), _d3(v), out)
    return np.array(out[:])


def view_transform(frm, to, up):
    return _mat_out(lib().orc_view_transform, _d3(frm), _d3(to), _d3(up))


def chain(*ms):
    """Matrix::identity().a().b()... fluent API (matrix.rs:238-266): each step LEFT-multiplies."""
    t = identity()
    for m in ms:
        t = matmul(m, t)
    return t


def default_material(**kw):
    """Material::default() (material.rs:83-96) with overrides."""
    m = Material()
    m.color = D3(*kw.pop("color", (1.0, 1.0, 1.0)))
    m.ambient = kw.pop("ambient", 0.1)
    m.diffuse = kw.pop("diffuse", 0.9)
    m.specular = kw.pop("specular", 0.9)
    m.shininess = kw.pop("shininess", 200.0)
    m.reflectivity = kw.pop("reflectivity", 0.0)
    m.transparency = kw.pop("transparency", 0.0)
    m.refractive_index = kw.pop("refractive_index", 1.0)
    m.casts_shadow = int(kw.pop("casts_shadow", True))
    m.pattern = kw.pop("pattern", -1)
    assert not kw, kw
    return m


class World:
    """Thin handle over orc_world."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_world_new())

    def __del__(self):
        try:
            if self.h:
                self.L.orc_world_free(self.h)
                self.h = None
        except Exception:
            pass

    # construction
    def material(self, **kw):
        m = kw.pop("_struct", None) or default_material(**kw)
        return self.L.orc_add_material(self.h, C.byref(m))

    def pattern(self, kind, a=(1, 1, 1), b=(0, 0, 0), sub_a=-1, sub_b=-1, transform=None):
        t = _d16(identity() if transform is None else transform)
        return self.L.orc_add_pattern(self.h, kind, _d3(a), _d3(b), sub_a, sub_b, t)

    def shape(self, kind, transform=None, material=None, parent=-1, minimum=-INF, maximum=INF, closed=False):
        if material is None:
            material = self.material()
        t = _d16(identity() if transform is None else transform)
        idx = self.L.orc_add_shape(self.h, kind, parent, t, material, minimum, maximum, int(closed))
        assert idx >= 0, idx
        return idx

    def triangle(self, p1, p2, p3, normals=None, material=None, parent=-1):
        if material is None:
            material = self.material()
        if normals is None:
            idx = self.L.orc_add_triangle(self.h, parent, _d3(p1), _d3(p2), _d3(p3), None, None, None, material)
        else:
            idx = self.L.orc_add_triangle(
                self.h, parent, _d3(p1), _d3(p2), _d3(p3), _d3(normals[0]), _d3(normals[1]), _d3(normals[2]),
                material)
        assert idx >= 0, idx
        return idx

    def csg(self, operation, parent=-1):
        """Csg node (shape.rs:2166-2184): add left, then right, with parent=<the returned index>"""
        idx = self.L.orc_add_csg(self.h, parent, operation)
        assert idx >= 0, idx
        return idx

    def csg_filter(self, csg, xs):
        """Csg::filter_intersections over [(t, shape), ...] -> indices of the records that pass"""
        n = len(xs)
        ts = (C.c_double * n)(*[float(x[0]) for x in xs])
        sh = (C.c_int * n)(*[int(x[1]) for x in xs])
        kept = (C.c_int * max(n, 1))()
        m = self.L.orc_csg_filter(self.h, csg, n, ts, sh, kept)
        assert m >= 0, m
        return list(kept[:m])

    def includes(self, shape, obj):
        return bool(self.L.orc_shape_includes(self.h, shape, obj))

    def set_material(self, shape, material):
        assert self.L.orc_shape_set_material(self.h, shape, material) == OK

    def point_light(self, position, intensity=(1, 1, 1)):
        return self.L.orc_add_point_light(self.h, _d3(position), _d3(intensity))

    def area_light(self, corner, full_u, usteps, full_v, vsteps, intensity=(1, 1, 1)):
        return self.L.orc_add_area_light(self.h, _d3(corner), _d3(full_u), usteps, _d3(full_v), vsteps,
                                         _d3(intensity))

    def camera(self, hsize, vsize, fov, view=None):
        self.hsize, self.vsize = hsize, vsize
        return self.L.orc_set_camera(self.h, hsize, vsize, fov, _d16(identity() if view is None else view))

    # hot path
    def render(self, threads=None, subset=None, depth=None):
        threads = threads or min(os.cpu_count() or 1, 16)
        self.L.orc_world_set_depth.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_world_set_depth.restype = None
        self.L.orc_world_set_depth(self.h, 5 if depth is None else depth)
        out = np.zeros((self.vsize, self.hsize, 3), dtype=np.float64)
        st = Stats()
        if subset is None:
            rc = self.L.orc_render(self.h, threads, out.ctypes.data_as(PD), C.byref(st))
        else:
            rb, re, rs, cs = subset
            rc = self.L.orc_render_subset(self.h, threads, rb, re, rs, cs, out.ctypes.data_as(PD), C.byref(st))
        if rc == ERR_NON_INVERTIBLE:
            raise ValueError("NonInvertibleMatrix")
        assert rc == OK, rc
        self.last_stats = dict(primary=st.primary_rays, shadow=st.shadow_rays, secondary=st.secondary_rays,
                               seconds=st.seconds, diverged=st.diverged_pixels)
        return out

    def render_pixels(self, xy, threads=None, depth=None):
        """colours of an explicit list of pixels [(x, y), ...] -> (n, 3) array (camera.rs:78-84 per pixel)"""
        threads = threads or min(os.cpu_count() or 1, 16)
        self.L.orc_world_set_depth.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_world_set_depth.restype = None
        self.L.orc_world_set_depth(self.h, 5 if depth is None else depth)
        xy = np.ascontiguousarray(np.asarray(xy, dtype=np.uint32).reshape(-1, 2))
        out = np.zeros((len(xy), 3), dtype=np.float64)
        st = Stats()
        rc = self.L.orc_render_pixels(self.h, threads, len(xy), xy.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      out.ctypes.data_as(PD), C.byref(st))
        if rc == ERR_NON_INVERTIBLE:
            raise ValueError("NonInvertibleMatrix")
        assert rc == OK, rc
        self.last_stats = dict(primary=st.primary_rays, shadow=st.shadow_rays, secondary=st.secondary_rays,
                               seconds=st.seconds, diverged=st.diverged_pixels)
        return out

    # unit-level
    def camera_params(self):
        o = D3()
        self.L.orc_camera_params(self.h, o)
        return tuple(o[:])

    def ray_for_pixel(self, px, py):
        o, d = D3(), D3()
        assert self.L.orc_ray_for_pixel(self.h, px, py, o, d) == OK
        return np.array(o[:]), np.array(d[:])

    def intersect(self, shape, origin, direction, cap=64):
        ts = (C.c_double * cap)()
        sh = (C.c_int * cap)()
        us = (C.c_double * cap)()
        vs = (C.c_double * cap)()
        n = self.L.orc_intersect(self.h, shape, _d3(origin), _d3(direction), cap, ts, sh, us, vs)
        assert n >= 0, n
        n = min(n, cap)
        return [(ts[i], sh[i], us[i], vs[i]) for i in range(n)]

    def local_intersect(self, shape, origin, direction, cap=16):
        ts = (C.c_double * cap)()
        n = self.L.orc_local_intersect(self.h, shape, _d3(origin), _d3(direction), cap, ts)
        return [ts[i] for i in range(min(n, cap))]

    def local_normal_at(self, shape, p, u=math.nan, v=math.nan):
        o = D3()
        self.L.orc_local_normal_at(self.h, shape, _d3(p), u, v, o)
        return np.array(o[:])

    def normal_at(self, shape, p, u=math.nan, v=math.nan):
        o = D3()
        assert self.L.orc_normal_at(self.h, shape, _d3(p), u, v, o) == OK
        return np.array(o[:])

    def world_to_object(self, shape, p):
        o = D3()
        self.L.orc_world_to_object(self.h, shape, _d3(p), o)
        return np.array(o[:])

    def normal_to_world(self, shape, n):
        o = D3()
        self.L.orc_normal_to_world(self.h, shape, _d3(n), o)
        return np.array(o[:])

    def color_at(self, origin, direction, remaining=5):
        o = D3()
        assert self.L.orc_color_at(self.h, _d3(origin), _d3(direction), remaining, o) == OK
        return np.array(o[:])

    def is_shadowed(self, point, light_position):
        r = C.c_int()
        assert self.L.orc_is_shadowed(self.h, _d3(point), _d3(light_position), C.byref(r)) == OK
        return bool(r.value)

    def intensity_at(self, light, point):
        r = C.c_double()
        assert self.L.orc_light_intensity_at(self.h, light, _d3(point), C.byref(r)) == OK
        return r.value

    def area_light_point(self, light, u, v):
        o = D3()
        assert self.L.orc_area_light_point(self.h, light, u, v, o) == OK
        return np.array(o[:])

    def light_info(self, light):
        p, u, v = D3(), D3(), D3()
        s = C.c_uint32()
        assert self.L.orc_light_info(self.h, light, p, u, v, C.byref(s)) == OK
        return np.array(p[:]), np.array(u[:]), np.array(v[:]), s.value

    def lighting(self, material, shape, light, point, eye, normal, intensity=1.0):
        o = D3()
        assert self.L.orc_lighting(self.h, material, shape, light, _d3(point), _d3(eye), _d3(normal), intensity,
                                   o) == OK
        return np.array(o[:])

    def pattern_at(self, pattern, p):
        o = D3()
        assert self.L.orc_pattern_at(self.h, pattern, _d3(p), o) == OK
        return np.array(o[:])

    def pattern_at_shape(self, pattern, shape, p):
        o = D3()
        assert self.L.orc_pattern_at_shape(self.h, pattern, shape, _d3(p), o) == OK
        return np.array(o[:])

    def prepare_computations(self, origin, direction, xs, hit_index):
        """xs: list of (t, shape) or (t, shape, u, v)."""
        n = len(xs)
        ts = (C.c_double * n)(*[float(x[0]) for x in xs])
        sh = (C.c_int * n)(*[int(x[1]) for x in xs])
        us = (C.c_double * n)(*[float(x[2]) if len(x) > 2 else math.nan for x in xs])
        vs = (C.c_double * n)(*[float(x[3]) if len(x) > 3 else math.nan for x in xs])
        out = (C.c_double * COMPS_LEN)()
        rc = self.L.orc_prepare_computations(self.h, _d3(origin), _d3(direction), n, ts, sh, us, vs, hit_index, out)
        assert rc == OK, rc
        return Comps(np.array(out[:]))

    def shade_hit(self, comps, remaining=5):
        o = D3()
        self.L.orc_shade_hit(self.h, comps.raw.ctypes.data_as(PD), remaining, o)
        return np.array(o[:])

    def reflected_color(self, comps, remaining=5):
        o = D3()
        self.L.orc_reflected_color(self.h, comps.raw.ctypes.data_as(PD), remaining, o)
        return np.array(o[:])

    def refracted_color(self, comps, remaining=5):
        o = D3()
        self.L.orc_refracted_color(self.h, comps.raw.ctypes.data_as(PD), remaining, o)
        return np.array(o[:])


class Comps:
    def __init__(self, raw):
        self.raw = np.ascontiguousarray(raw, dtype=np.float64)

    t = property(lambda s: s.raw[0])
    point = property(lambda s: s.raw[1:4])
    eye = property(lambda s: s.raw[4:7])
    normal = property(lambda s: s.raw[7:10])
    inside = property(lambda s: bool(s.raw[10]))
    over_point = property(lambda s: s.raw[11:14])
    under_point = property(lambda s: s.raw[14:17])
    reflect = property(lambda s: s.raw[17:20])
    n1 = property(lambda s: s.raw[20])
    n2 = property(lambda s: s.raw[21])
    shape = property(lambda s: int(s.raw[22]))

    def schlick(self):
        return lib().orc_schlick(self.raw.ctypes.data_as(PD))


def hit_index(ts):
    n = len(ts)
    a = (C.c_double * n)(*[float(t) for t in ts])
    return lib().orc_hit_index(n, a)


def scale_and_clamp(v):
    return int(lib().orc_scale_and_clamp(float(v)))


def canvas_to_ppm(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w = rgb.shape[0], rgb.shape[1]
    p = rgb.ctypes.data_as(PD)
    n = lib().orc_canvas_to_ppm(p, w, h, None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().orc_canvas_to_ppm(p, w, h, buf, n)
    return buf.raw[:n].decode("ascii")


def default_world(light_pos=(-10, 10, -10), m1=None, m2=None):
    """World::default() (world.rs:150-174). Returns (world, sphere1, sphere2, light)."""
    w = World()
    light = w.point_light(light_pos, (1, 1, 1))
    mat1 = w.material(**(m1 or dict(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)))
    s1 = w.shape(SPHERE, material=mat1)
    mat2 = w.material(**(m2 or {}))
    s2 = w.shape(SPHERE, transform=scaling(0.5, 0.5, 0.5), material=mat2)
    return w, s1, s2, light
