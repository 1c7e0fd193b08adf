"""ray-tracer-challenge_amd: MI355X-native implementation of the per-pixel render path of
anna-hope/ray-tracer-challenge (Camera::render -> World::color_at -> ...), behind the reference's own
interface names.  The compute lives in librtc_amd.so (HIP kernels for gfx950 + host flattener/BVH, C ABI in
include/rtc.h); this package is the thin host-side mirror.  There is no CPU render path here.
"""
from . import _native, parallel, stress_scene
from ._native import NonInvertibleMatrix, RtcError
from .api import (AreaLight, BlendedPattern, Camera, Canvas, CheckerPattern, CompiledScene, Cone, Csg, CsgDifference,
                  CsgIntersection, CsgUnion, Cube, Cylinder,
                  GradientPattern, Group, Material, Matrix, NativeScene, Pattern, Plane, PointLight, RadialGradientPattern,
                  RingPattern, Shape, SmoothTriangle, SolidPattern, Sphere, StripePattern, TestPattern, Triangle,
                  TriangleMesh, World, canvas_to_ppm, compute_view_transformation, glass_sphere)
from .scene_yaml import (ParsedObj, ParseError, Scene, parse_obj_file, parse_obj_string, parse_scene,
                         parse_scene_file)

__all__ = [n for n in dir() if not n.startswith("_")]
