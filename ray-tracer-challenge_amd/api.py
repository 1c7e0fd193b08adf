"""Host-side mirror of the reference's public interface for the render path.

Names and argument meaning follow `ray_tracer::prelude` (ray_tracer/src/prelude.rs:1-10): Camera, World,
Material, PointLight/AreaLight, Matrix, the shapes, `compute_view_transformation`.  The objects here are
plain descriptions; `Camera.render(world)` (camera.rs:64) flattens them through the C ABI
(include/rtc.h) and runs the HIP kernels.  Nothing in this module computes pixels on the CPU.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N

INF = float("inf")


def _d3(v):
    return N.D3(float(v[0]), float(v[1]), float(v[2]))


def _d16(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(16))
    return N.D16(*a.tolist())


# ----------------------------------------------------------------------------- Matrix (matrix.rs)

class Matrix:
    """4x4 f64 matrix; arithmetic is done by librtc_amd's bit-faithful host math (matrix.rs)."""

    def __init__(self, rows=None):
        self.m = np.eye(4) if rows is None else np.array(rows, dtype=np.float64).reshape(4, 4)

    @staticmethod
    def _from(fn, *args):
        out = N.D16()
        fn(*args, out)
        return Matrix(np.array(out[:]).reshape(4, 4))

    @staticmethod
    def identity():
        return Matrix._from(N.lib().rtc_matrix_identity)

    @staticmethod
    def translation(x, y, z):
        return Matrix._from(N.lib().rtc_matrix_translation, x, y, z)

    @staticmethod
    def scaling(x, y, z):
        return Matrix._from(N.lib().rtc_matrix_scaling, x, y, z)

    @staticmethod
    def rotation_x(r):
        return Matrix._from(N.lib().rtc_matrix_rotation_x, r)

    @staticmethod
    def rotation_y(r):
        return Matrix._from(N.lib().rtc_matrix_rotation_y, r)

    @staticmethod
    def rotation_z(r):
        return Matrix._from(N.lib().rtc_matrix_rotation_z, r)

    @staticmethod
    def shearing(xy, xz, yx, yz, zx, zy):
        return Matrix._from(N.lib().rtc_matrix_shearing, xy, xz, yx, yz, zx, zy)

    def __matmul__(self, other):
        return Matrix._from(N.lib().rtc_matrix_multiply, _d16(self.m), _d16(other.m))

    __mul__ = __matmul__

    # fluent API, matrix.rs:238-266: each call LEFT-multiplies
    def translate(self, x, y, z):
        return Matrix.translation(x, y, z) @ self

    def scale(self, x, y, z):
        return Matrix.scaling(x, y, z) @ self

    def rotate_x(self, r):
        return Matrix.rotation_x(r) @ self

    def rotate_y(self, r):
        return Matrix.rotation_y(r) @ self

    def rotate_z(self, r):
        return Matrix.rotation_z(r) @ self

    def shear(self, *a):
        return Matrix.shearing(*a) @ self

    def inverse(self):
        out = N.D16()
        N.check(N.lib().rtc_matrix_inverse(_d16(self.m), out))
        return Matrix(np.array(out[:]).reshape(4, 4))

    def __eq__(self, other):  # matrix.rs:289-300
        return bool(np.all(np.abs(self.m - other.m) < 1e-5))

    def __repr__(self):
        return f"Matrix({self.m.tolist()})"


def compute_view_transformation(frm, to, up) -> Matrix:
    """transformation.rs:3-15"""
    out = N.D16()
    N.lib().rtc_view_transform(_d3(frm), _d3(to), _d3(up), out)
    return Matrix(np.array(out[:]).reshape(4, 4))


# ----------------------------------------------------------------------------- patterns (pattern.rs)

@dataclass
class Pattern:
    kind: int
    a: Sequence[float] = (1.0, 1.0, 1.0)
    b: Sequence[float] = (0.0, 0.0, 0.0)
    sub_a: Optional["Pattern"] = None
    sub_b: Optional["Pattern"] = None
    transformation: Matrix = field(default_factory=Matrix)


def SolidPattern(color=(1.0, 1.0, 1.0)):
    return Pattern(N.PAT_SOLID, a=color)


def StripePattern(a=None, b=None, transformation=None):
    return Pattern(N.PAT_STRIPE, sub_a=a or SolidPattern(), sub_b=b or SolidPattern((0, 0, 0)),
                   transformation=transformation or Matrix())


def CheckerPattern(a=None, b=None, transformation=None):
    return Pattern(N.PAT_CHECKER, sub_a=a or SolidPattern(), sub_b=b or SolidPattern((0, 0, 0)),
                   transformation=transformation or Matrix())


def BlendedPattern(a, b, transformation=None):
    return Pattern(N.PAT_BLENDED, sub_a=a, sub_b=b, transformation=transformation or Matrix())


def GradientPattern(a=(1, 1, 1), b=(0, 0, 0), transformation=None):
    return Pattern(N.PAT_GRADIENT, a=a, b=b, transformation=transformation or Matrix())


def RingPattern(a=(1, 1, 1), b=(0, 0, 0), transformation=None):
    return Pattern(N.PAT_RING, a=a, b=b, transformation=transformation or Matrix())


def RadialGradientPattern(a=(1, 1, 1), b=(0, 0, 0), transformation=None):
    return Pattern(N.PAT_RADIAL, a=a, b=b, transformation=transformation or Matrix())


def TestPattern(transformation=None):
    return Pattern(N.PAT_TEST, transformation=transformation or Matrix())


# ----------------------------------------------------------------------------- material.rs

@dataclass
class Material:
    """material.rs:4-15, defaults :83-96"""
    color: Sequence[float] = (1.0, 1.0, 1.0)
    ambient: float = 0.1
    diffuse: float = 0.9
    specular: float = 0.9
    shininess: float = 200.0
    pattern: Optional[Pattern] = None
    reflectivity: float = 0.0
    transparency: float = 0.0
    refractive_index: float = 1.0
    casts_shadow: bool = True


# ----------------------------------------------------------------------------- shapes (shape.rs)

@dataclass
class Shape:
    kind: int
    transformation: Matrix = field(default_factory=Matrix)
    material: Material = field(default_factory=Material)
    minimum: float = -INF
    maximum: float = INF
    closed: bool = False

    def with_transformation(self, t):
        self.transformation = t
        return self

    def with_material(self, m):
        self.material = m
        return self


def Sphere(transformation=None, material=None):
    return Shape(N.SPHERE, transformation or Matrix(), material or Material())


def glass_sphere():
    """Sphere::glass(), shape.rs:194-206"""
    return Sphere(material=Material(transparency=1.0, refractive_index=1.5))


def Plane(transformation=None, material=None):
    return Shape(N.PLANE, transformation or Matrix(), material or Material())


def Cube(transformation=None, material=None):
    return Shape(N.CUBE, transformation or Matrix(), material or Material())


def Cylinder(transformation=None, material=None, minimum=-INF, maximum=INF, closed=False):
    return Shape(N.CYLINDER, transformation or Matrix(), material or Material(), minimum, maximum, closed)


def Cone(transformation=None, material=None, minimum=-INF, maximum=INF, closed=False):
    return Shape(N.CONE, transformation or Matrix(), material or Material(), minimum, maximum, closed)


@dataclass
class Triangle:
    """Triangle (shape.rs:1446-1477) or, with normals, SmoothTriangle (shape.rs:1667-1710)."""
    point1: Sequence[float]
    point2: Sequence[float]
    point3: Sequence[float]
    normal1: Optional[Sequence[float]] = None
    normal2: Optional[Sequence[float]] = None
    normal3: Optional[Sequence[float]] = None
    material: Material = field(default_factory=Material)


def SmoothTriangle(p1, p2, p3, n1, n2, n3, material=None):
    return Triangle(p1, p2, p3, n1, n2, n3, material or Material())


@dataclass
class TriangleMesh:
    """A run of triangles added with one call (what obj_file produces for one `g` group)."""
    points: np.ndarray                   # (n, 9)
    normals: Optional[np.ndarray] = None  # (n, 9) or None
    material: Material = field(default_factory=Material)


@dataclass
class Group:
    """shape.rs:1891-1928.  `material` is the group's Option<Material>: Some(default) unless changed."""
    transformation: Matrix = field(default_factory=Matrix)
    children: List[object] = field(default_factory=list)
    material: Optional[Material] = field(default_factory=Material)

    def add_child(self, child):
        self.children.append(child)
        return self


@dataclass
class Csg:
    """shape.rs:2156-2184: Csg::new(operation, left, right); operands are shapes, groups or other Csg nodes"""
    operation: int
    left: object
    right: object


def CsgUnion(left, right):
    return Csg(N.CSG_UNION, left, right)


def CsgIntersection(left, right):
    return Csg(N.CSG_INTERSECTION, left, right)


def CsgDifference(left, right):
    return Csg(N.CSG_DIFFERENCE, left, right)


# ----------------------------------------------------------------------------- lights (light.rs)

@dataclass
class PointLight:
    position: Sequence[float] = (0.0, 0.0, 0.0)
    intensity: Sequence[float] = (1.0, 1.0, 1.0)


@dataclass
class AreaLight:
    corner: Sequence[float]
    full_uvec: Sequence[float]
    usteps: int
    full_vvec: Sequence[float]
    vsteps: int
    intensity: Sequence[float] = (1.0, 1.0, 1.0)


# ----------------------------------------------------------------------------- world.rs

@dataclass
class World:
    objects: List[object] = field(default_factory=list)
    lights: List[object] = field(default_factory=list)

    @staticmethod
    def default():
        """World::default(), world.rs:150-174"""
        s1 = Sphere(material=Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2))
        s2 = Sphere(transformation=Matrix.scaling(0.5, 0.5, 0.5))
        return World([s1, s2], [PointLight((-10, 10, -10), (1, 1, 1))])


# ----------------------------------------------------------------------------- flattening through the C ABI

class CompiledScene:
    """An rtc_scene handle: World + Camera pushed through the C ABI and committed to one GPU."""

    def __init__(self, world: World, camera: "Camera", device=0):
        """device: one HIP device index, or a list of them (rtc_scene_commit_devices: one process, several GPUs)"""
        self.L = N.lib()
        self.h = C.c_void_p(self.L.rtc_scene_create())
        if not self.h:
            raise MemoryError("rtc_scene_create failed")
        self.hsize, self.vsize = camera.hsize, camera.vsize
        self._mat_ids = {}
        self._pat_ids = {}
        self._keep = []
        L, h = self.L, self.h
        N.check(L.rtc_scene_set_camera(h, camera.hsize, camera.vsize, camera.field_of_view,
                                       _d16(camera.transformation.m)))
        for light in world.lights:
            if isinstance(light, AreaLight):
                N.check(L.rtc_scene_add_area_light(h, _d3(light.corner), _d3(light.full_uvec), light.usteps,
                                                   _d3(light.full_vvec), light.vsteps, _d3(light.intensity)))
            else:
                N.check(L.rtc_scene_add_point_light(h, _d3(light.position), _d3(light.intensity)))
        for obj in world.objects:
            self._add(obj)
        if isinstance(device, (list, tuple)):
            arr = (C.c_int * len(device))(*device)
            N.check(L.rtc_scene_commit_devices(h, arr, len(device)))
            device = device[0]
        else:
            N.check(L.rtc_scene_commit(h, device))
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.rtc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pattern(self, p: Pattern) -> int:
        key = id(p)
        if key in self._pat_ids:
            return self._pat_ids[key]
        sa = self._pattern(p.sub_a) if p.sub_a is not None else -1
        sb = self._pattern(p.sub_b) if p.sub_b is not None else -1
        out = C.c_uint32()
        N.check(self.L.rtc_scene_add_pattern(self.h, p.kind, _d3(p.a), _d3(p.b), sa, sb, _d16(p.transformation.m),
                                             C.byref(out)))
        self._pat_ids[key] = out.value
        self._keep.append(p)
        return out.value

    def _material(self, m: Optional[Material]) -> int:
        if m is None:
            return -1
        key = id(m)
        if key in self._mat_ids:
            return self._mat_ids[key]
        s = N.Material()
        s.color = N.D3(*[float(c) for c in m.color])
        s.ambient, s.diffuse, s.specular, s.shininess = m.ambient, m.diffuse, m.specular, m.shininess
        s.reflectivity, s.transparency, s.refractive_index = m.reflectivity, m.transparency, m.refractive_index
        s.casts_shadow = 1 if m.casts_shadow else 0
        s.pattern = self._pattern(m.pattern) if m.pattern is not None else -1
        out = C.c_uint32()
        N.check(self.L.rtc_scene_add_material(self.h, C.byref(s), C.byref(out)))
        self._mat_ids[key] = out.value
        self._keep.append(m)
        return out.value

    def _add(self, obj):
        L, h = self.L, self.h
        if isinstance(obj, Group):
            N.check(L.rtc_scene_begin_group(h, _d16(obj.transformation.m), self._material(obj.material)))
            for c in obj.children:
                self._add(c)
            N.check(L.rtc_scene_end_group(h))
        elif isinstance(obj, Csg):
            N.check(L.rtc_scene_begin_csg(h, obj.operation))
            self._add(obj.left)
            self._add(obj.right)
            N.check(L.rtc_scene_end_csg(h))
        elif isinstance(obj, Shape):
            N.check(L.rtc_scene_add_shape(h, obj.kind, _d16(obj.transformation.m), self._material(obj.material),
                                          obj.minimum, obj.maximum, int(obj.closed)))
        elif isinstance(obj, Triangle):
            pts = np.array([*obj.point1, *obj.point2, *obj.point3], dtype=np.float64)
            nrm = None
            if obj.normal1 is not None:
                nrm = np.array([*obj.normal1, *obj.normal2, *obj.normal3], dtype=np.float64)
            N.check(L.rtc_scene_add_triangles(h, 1, pts.ctypes.data_as(N.PD),
                                              nrm.ctypes.data_as(N.PD) if nrm is not None else None,
                                              self._material(obj.material)))
        elif isinstance(obj, TriangleMesh):
            pts = np.ascontiguousarray(obj.points, dtype=np.float64).reshape(-1, 9)
            nrm = None
            if obj.normals is not None:
                nrm = np.ascontiguousarray(obj.normals, dtype=np.float64).reshape(-1, 9)
            N.check(L.rtc_scene_add_triangles(h, pts.shape[0], pts.ctypes.data_as(N.PD),
                                              nrm.ctypes.data_as(N.PD) if nrm is not None else None,
                                              self._material(obj.material)))
        else:
            raise TypeError(f"unsupported object {type(obj)}")

    # ---- rendering ----
    def _opts(self, depth=5, tile_rows=0, tile_index=0, tile_count=1, use_bvh=0, collect_stats=False, schedule=0,
              exchange=0, max_rays_per_pixel=0, fp_mode=0, tail_polls=0):
        o = N.RenderOpts()
        self.L.rtc_render_opts_default(C.byref(o))
        o.depth, o.tile_rows, o.tile_index, o.tile_count = depth, tile_rows, tile_index, tile_count
        o.use_bvh, o.collect_stats = use_bvh, 1 if collect_stats else 0
        o.schedule, o.exchange, o.max_rays_per_pixel, o.fp_mode = schedule, exchange, max_rays_per_pixel, fp_mode
        o.tail_polls = tail_polls
        return o

    def render(self, out=None, **kw):
        """rtc_render: full canvas as a (vsize, hsize, 3) float64 numpy array + stats dict.  RTC_ERR_OVERFLOW raises
        RtcError; the canvas written so far is kept in `last_canvas`."""
        o = self._opts(**kw)
        if out is None:
            out = np.zeros((self.vsize, self.hsize, 3), dtype=np.float64)
        st = N.Stats()
        self.last_canvas = out
        rc = self.L.rtc_render(self.h, C.byref(o), out.ctypes.data_as(N.PD), C.byref(st))
        self.last_stats = st.as_dict()
        N.check(rc)
        return out

    def check(self):
        """rtc_scene_check: synchronise and report the most recent launch (raises RtcError on overflow / fault)"""
        N.check(self.L.rtc_scene_check(self.h))

    def local_rows(self, tile_rows=0, tile_index=0, tile_count=1):
        return int(self.L.rtc_local_rows(self.vsize, tile_rows, tile_index, tile_count))

    def local_row_map(self, tile_rows=0, tile_index=0, tile_count=1):
        n = self.local_rows(tile_rows, tile_index, tile_count)
        buf = (C.c_uint32 * max(n, 1))()
        self.L.rtc_local_row_map(self.vsize, tile_rows, tile_index, tile_count, buf, n)
        return np.array(buf[:n], dtype=np.int64)

    def render_device(self, device_ptr: int, stream: int = 0, **kw):
        """rtc_render_device: asynchronous, into caller-owned device memory (e.g. a torch tensor)."""
        o = self._opts(**kw)
        N.check(self.L.rtc_render_device(self.h, C.byref(o), C.c_void_p(device_ptr), C.c_void_p(stream)))

    def render_device_static(self, device_ptr: int, stream: int = 0, **kw):
        """rtc_render_device_static: the same launch with the library's static device array as its work buffer."""
        o = self._opts(**kw)
        N.check(self.L.rtc_render_device_static(self.h, C.byref(o), C.c_void_p(device_ptr), C.c_void_p(stream)))

    def stats(self):
        st = N.Stats()
        N.check(self.L.rtc_scene_last_stats(self.h, C.byref(st)))
        return st.as_dict()

    def info(self):
        i = N.SceneInfo()
        N.check(self.L.rtc_scene_get_info(self.h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}


# ----------------------------------------------------------------------------- camera.rs / canvas.rs

class NativeScene(CompiledScene):
    """An rtc_scene built by the C++ loaders (rtc_scene_load_yaml*, rtc_scene_add_obj) instead of api.py objects,
    plus the builder-tree getters of include/rtc.h."""

    def __init__(self, handle):
        self.L = N.lib()
        self.h = C.c_void_p(handle)
        self.device = None
        hs, vs = C.c_uint32(), C.c_uint32()
        if self.L.rtc_scene_get_camera(self.h, C.byref(hs), C.byref(vs), None, None) == N.OK:
            self.hsize, self.vsize = hs.value, vs.value

    @staticmethod
    def from_yaml(text: str, base_dir: Optional[str] = None, width: int = 0, height: int = 0) -> "NativeScene":
        h = C.c_void_p()
        N.check(N.lib().rtc_scene_load_yaml(text.encode(), base_dir.encode() if base_dir else None, width or 0,
                                            height or 0, C.byref(h)))
        return NativeScene(h.value)

    @staticmethod
    def from_yaml_file(path: str, width: int = 0, height: int = 0) -> "NativeScene":
        h = C.c_void_p()
        N.check(N.lib().rtc_scene_load_yaml_file(path.encode(), width or 0, height or 0, C.byref(h)))
        return NativeScene(h.value)

    @staticmethod
    def from_obj(text: str, transformation: Optional[Matrix] = None, material_id: int = -1):
        """parse_obj_string into an otherwise empty scene -> (scene, ObjInfo dict)"""
        L = N.lib()
        sc = NativeScene(L.rtc_scene_create())
        info = N.ObjInfo()
        t = _d16(transformation.m) if transformation is not None else None
        N.check(L.rtc_scene_add_obj(sc.h, text.encode(), t, material_id, C.byref(info)))
        return sc, {n: getattr(info, n) for n, _ in info._fields_}

    def commit(self, device=0):
        if isinstance(device, (list, tuple)):
            arr = (C.c_int * len(device))(*device)
            N.check(self.L.rtc_scene_commit_devices(self.h, arr, len(device)))
            device = device[0]
        else:
            N.check(self.L.rtc_scene_commit(self.h, device))
        self.device = device
        return self

    # ---- builder tree ----
    def item(self, idx: int) -> dict:
        it = N.ItemInfo()
        N.check(self.L.rtc_scene_get_item(self.h, idx, C.byref(it)))
        d = {n: getattr(it, n) for n, _ in it._fields_ if n != "transform"}
        d["transform"] = np.array(it.transform[:]).reshape(4, 4)
        return d

    def children(self, parent: int = -1) -> List[int]:
        out = []
        for k in range(self.L.rtc_scene_child_count(self.h, parent)):
            v = C.c_uint32()
            N.check(self.L.rtc_scene_get_child(self.h, parent, k, C.byref(v)))
            out.append(v.value)
        return out

    def material(self, mid: int) -> Material:
        m = N.Material()
        N.check(self.L.rtc_scene_get_material(self.h, mid, C.byref(m)))
        return Material(color=tuple(m.color), ambient=m.ambient, diffuse=m.diffuse, specular=m.specular,
                        shininess=m.shininess, reflectivity=m.reflectivity, transparency=m.transparency,
                        refractive_index=m.refractive_index, casts_shadow=bool(m.casts_shadow))

    def triangles(self, idx: int):
        it = self.item(idx)
        n = it["n_triangles"]
        pts = np.zeros((n, 9))
        nrm = np.zeros((n, 9)) if it["smooth"] else None
        N.check(self.L.rtc_scene_get_triangles(self.h, idx, 0, n, pts.ctypes.data_as(N.PD),
                                               nrm.ctypes.data_as(N.PD) if nrm is not None else None))
        return pts, nrm

    def lights(self) -> List[dict]:
        out = []
        k = 0
        while True:
            li = N.LightInfo()
            if self.L.rtc_scene_get_light(self.h, k, C.byref(li)) != N.OK:
                break
            out.append({n: (list(getattr(li, n)) if hasattr(getattr(li, n), "__len__") else getattr(li, n))
                        for n, _ in li._fields_})
            k += 1
        return out

    def camera(self) -> dict:
        hs, vs, fov, view = C.c_uint32(), C.c_uint32(), C.c_double(), N.D16()
        N.check(self.L.rtc_scene_get_camera(self.h, C.byref(hs), C.byref(vs), C.byref(fov), view))
        return dict(hsize=hs.value, vsize=vs.value, field_of_view=fov.value, view=np.array(view[:]).reshape(4, 4))


def canvas_to_ppm(pixels: np.ndarray) -> str:
    """Canvas::to_ppm (canvas.rs:44-82) through rtc_canvas_to_ppm"""
    px = np.ascontiguousarray(pixels, dtype=np.float64)
    h, w = px.shape[0], px.shape[1]
    need = C.c_size_t()
    L = N.lib()
    N.check(L.rtc_canvas_to_ppm(px.ctypes.data_as(N.PD), w, h, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value + 1)
    N.check(L.rtc_canvas_to_ppm(px.ctypes.data_as(N.PD), w, h, buf, need.value + 1, None))
    return buf.raw[: need.value].decode("ascii")


class Canvas:
    """canvas.rs:7-41 over a (height, width, 3) float64 array."""

    def __init__(self, pixels: np.ndarray):
        self.pixels = pixels
        self.height, self.width = pixels.shape[0], pixels.shape[1]

    def pixel_at(self, x, y):
        return self.pixels[y, x]

    def write_pixel(self, x, y, color):
        self.pixels[y, x] = color

    def to_ppm(self) -> str:
        return canvas_to_ppm(self.pixels)


@dataclass
class Camera:
    """camera.rs:7-37"""
    hsize: int
    vsize: int
    field_of_view: float
    transformation: Matrix = field(default_factory=Matrix)

    def compile(self, world: World, device: int = 0) -> CompiledScene:
        return CompiledScene(world, self, device)

    def render(self, world: World, device: int = 0, **opts) -> Canvas:
        """Camera::render, camera.rs:64-93 -- on the GPU."""
        scene = self.compile(world, device)
        try:
            return Canvas(scene.render(**opts))
        finally:
            scene.close()
