"""Seeded random scenes for the fuzz parity tests (tests/test_gpu_fuzz.py on the GPU, tests/test_oracle_guard.py on
the CPU): every shape kind, nested groups, Csg trees, patterns, glass and mirrors, point and area lights, random
transforms (rotations, non-uniform scales, shears) and cameras.  Deterministic per seed (numpy Generator, PCG64)."""
import math

import numpy as np

from scene_feed import rtc

M = rtc.Matrix


def _colour(rng, lo=0.05, hi=1.0):
    return tuple(float(v) for v in rng.uniform(lo, hi, 3))


def _transform(rng, spread=3.0, lift=True):
    t = M.scaling(*[float(v) for v in rng.uniform(0.35, 1.3, 3)])
    if rng.random() < 0.25:
        t = t.shear(*[float(v) for v in rng.uniform(-0.4, 0.4, 6)])
    t = t.rotate_x(float(rng.uniform(-math.pi, math.pi))).rotate_y(float(rng.uniform(-math.pi, math.pi)))
    if rng.random() < 0.5:
        t = t.rotate_z(float(rng.uniform(-math.pi, math.pi)))
    y = float(rng.uniform(0.2, 2.6)) if lift else float(rng.uniform(-1.0, 1.0))
    return t.translate(float(rng.uniform(-spread, spread)), y, float(rng.uniform(-2.0, 3.5)))


def _pattern(rng, level=0):
    kind = int(rng.integers(0, 8))
    t = M.scaling(*[float(v) for v in rng.uniform(0.15, 1.2, 3)]).rotate_y(float(rng.uniform(0, math.pi)))
    if rng.random() < 0.5:
        t = t.rotate_z(float(rng.uniform(0, math.pi))).translate(*[float(v) for v in rng.uniform(-1, 1, 3)])
    sub = (lambda: _pattern(rng, level + 1)) if level < 2 and rng.random() < 0.35 else (lambda: rtc.SolidPattern(_colour(rng)))
    if kind == 0:
        return rtc.StripePattern(sub(), sub(), t)
    if kind == 1:
        return rtc.CheckerPattern(sub(), sub(), t)
    if kind == 2:
        return rtc.BlendedPattern(sub(), sub(), t)
    if kind == 3:
        return rtc.GradientPattern(_colour(rng), _colour(rng), t)
    if kind == 4:
        return rtc.RingPattern(_colour(rng), _colour(rng), t)
    if kind == 5:
        return rtc.RadialGradientPattern(_colour(rng), _colour(rng), t)
    if kind == 6:
        return rtc.TestPattern(t)
    return rtc.SolidPattern(_colour(rng))


def _material(rng, glass_ok=True):
    m = rtc.Material(color=_colour(rng), ambient=float(rng.uniform(0.0, 0.3)), diffuse=float(rng.uniform(0.2, 1.0)),
                     specular=float(rng.uniform(0.0, 1.0)), shininess=float(rng.uniform(5.0, 300.0)))
    r = rng.random()
    if r < 0.25:
        m.reflectivity = float(rng.uniform(0.1, 1.0))
    elif r < 0.5 and glass_ok:
        m.transparency = float(rng.uniform(0.3, 1.0))
        m.refractive_index = float(rng.uniform(1.0, 2.5))
        m.reflectivity = float(rng.uniform(0.0, 1.0)) if rng.random() < 0.6 else 0.0
        m.diffuse = float(rng.uniform(0.05, 0.4))
    if rng.random() < 0.3:
        m.pattern = _pattern(rng)
    if rng.random() < 0.15:
        m.casts_shadow = False
    return m


def _leaf(rng, spread=3.0, lift=True):
    kind = int(rng.integers(0, 5))
    t, m = _transform(rng, spread, lift), _material(rng)
    if kind == 0:
        return rtc.Sphere(t, m)
    if kind == 1:
        return rtc.Cube(t, m)
    if kind in (2, 3):
        ctor = rtc.Cylinder if kind == 2 else rtc.Cone
        lo = float(rng.uniform(-1.5, 0.2))
        hi = lo + float(rng.uniform(0.3, 2.0))
        if rng.random() < 0.12:
            return ctor(t, m)  # infinite, open
        return ctor(t, m, lo, hi, bool(rng.random() < 0.6))
    # a small fan of triangles under a group that decides their material (shape.rs:1520-1532)
    g = rtc.Group(transformation=t, material=m if rng.random() < 0.7 else rtc.Material())
    for _ in range(int(rng.integers(1, 5))):
        p = rng.uniform(-1.0, 1.0, (3, 3))
        n = rng.uniform(-1.0, 1.0, (3, 3))
        # glass fans are flat: the interpolated normal of a smooth triangle may point to the other side of its face than
        # the eye vector says, under_point (intersection.rs:88) then lies on the eye's side, the refracted ray -- which
        # keeps its bounce budget, world.rs:142 -- meets the same triangle again, and again: the reference overflows its
        # stack on such a scene (and so does the oracle); this library reports RTC_ERR_OVERFLOW
        if rng.random() < 0.5 or g.material.transparency > 0.0:
            g.add_child(rtc.Triangle(tuple(p[0]), tuple(p[1]), tuple(p[2])))
        else:
            n /= np.linalg.norm(n, axis=1, keepdims=True)
            g.add_child(rtc.SmoothTriangle(tuple(p[0]), tuple(p[1]), tuple(p[2]), tuple(n[0]), tuple(n[1]), tuple(n[2])))
    return g


def _csg_operand(rng, level):
    if level < 2 and rng.random() < 0.3:
        return _csg(rng, level + 1)
    kind = int(rng.integers(0, 4))
    t = M.scaling(*[float(v) for v in rng.uniform(0.5, 1.1, 3)]).rotate_y(float(rng.uniform(0, math.pi)))
    t = t.translate(*[float(v) for v in rng.uniform(-0.6, 0.6, 3)])
    m = _material(rng)
    if kind == 0:
        return rtc.Sphere(t, m)
    if kind == 1:
        return rtc.Cube(t, m)
    if kind == 2:
        return rtc.Cylinder(t, m, -1.0, 1.0, True)
    return rtc.Group(transformation=t, children=[rtc.Sphere(M.translation(-0.4, 0, 0), m), rtc.Cube(M.scaling(0.5, 0.5, 0.5).translate(0.5, 0, 0), m)])


def _csg(rng, level=0):
    op = [rtc.CsgUnion, rtc.CsgIntersection, rtc.CsgDifference][int(rng.integers(0, 3))]
    return op(_csg_operand(rng, level), _csg_operand(rng, level))


def _group(rng, level=0):
    g = rtc.Group(transformation=_transform(rng, 2.0 if level == 0 else 1.0, lift=(level == 0)))
    for _ in range(int(rng.integers(2, 6))):
        if level < 2 and rng.random() < 0.3:
            g.add_child(_group(rng, level + 1))
        else:
            g.add_child(_leaf(rng, 1.2, lift=False))
    return g


def random_scene(seed, w=120, h=80, n_objects=None):
    """(world, camera, depth) of seed; n_objects overrides the 3..13 top-level objects (tools/fuzz_hunt.py: larger scenes)"""
    rng = np.random.default_rng(seed)
    objs = []
    if rng.random() < 0.8:
        objs.append(rtc.Plane(material=_material(rng, glass_ok=False)))
    if rng.random() < 0.4:
        objs.append(rtc.Plane(M.rotation_x(math.pi / 2).translate(0, 0, float(rng.uniform(4, 8))), _material(rng, glass_ok=False)))
    for _ in range(int(rng.integers(3, 14)) if n_objects is None else n_objects):
        r = rng.random()
        if r < 0.6:
            objs.append(_leaf(rng))
        elif r < 0.85:
            objs.append(_group(rng))
        else:
            # the Csg node has no transformation of its own: wrap it in a group to place it
            objs.append(rtc.Group(transformation=_transform(rng, 2.5), children=[_csg(rng)]))
    n_lights = int(rng.integers(1, 4))
    if n_objects is not None:
        # the reference traces the secondary rays of a hit once per light and refraction keeps the bounce budget: a chain
        # of k glass surfaces costs 2^k with two lights -- in a crowd of glass the oracle would not finish
        n_lights = 1
    lights = []
    for k in range(n_lights):
        pos = (float(rng.uniform(-8, 8)), float(rng.uniform(4, 10)), float(rng.uniform(-9, -2)))
        if rng.random() < 0.2:
            lights.append(rtc.AreaLight(pos, (float(rng.uniform(0.5, 2)), 0, 0), int(rng.integers(1, 4)),
                                        (0, float(rng.uniform(0.5, 2)), 0), int(rng.integers(1, 4)), _colour(rng, 0.2, 0.9)))
        else:
            lights.append(rtc.PointLight(pos, _colour(rng, 0.2, 0.9)))
    # the reference re-traces the secondary rays once per light (world.rs:45-68): keep the oracle's tree in bounds
    depth = int(rng.integers(0, 6)) if n_lights == 1 else int(rng.integers(0, 4))
    if rng.random() < 0.25:
        w, h = h, w  # portrait
    frm = (float(rng.uniform(-4, 4)), float(rng.uniform(0.5, 5)), float(rng.uniform(-9, -5)))
    to = (float(rng.uniform(-1, 1)), float(rng.uniform(0.3, 1.8)), float(rng.uniform(-0.5, 1.0)))
    cam = rtc.Camera(w, h, float(rng.uniform(0.6, 1.5)), rtc.compute_view_transformation(frm, to, (0, 1, 0)))
    return rtc.World(objs, lights), cam, depth


def crowd_scene(seed, n=36_000, w=480, h=270):
    """(world, camera): `n` small shapes of every kind in a slab in front of the camera -- enough for the hierarchy to
    be stored quantised (more than 1 MB of f32 nodes, rtc_scene.cpp build_bvh) with leaves that are NOT all compact
    spheres: rotated and sheared spheres, cubes, capped cylinders and cones, triangles in groups, nested groups; one in
    five glass or mirror (spheres and cubes only: closed convex shapes let every refracted ray out again)"""
    rng = np.random.default_rng(seed)
    objs = [rtc.Plane(material=rtc.Material(color=(0.8, 0.8, 0.75), reflectivity=0.2, specular=0.1,
                                            pattern=rtc.CheckerPattern(rtc.SolidPattern((0.9, 0.9, 0.9)), rtc.SolidPattern((0.3, 0.3, 0.35)))))]
    mats = []
    for _ in range(64):
        m = rtc.Material(color=_colour(rng), ambient=float(rng.uniform(0.05, 0.2)), diffuse=float(rng.uniform(0.4, 0.9)),
                         specular=float(rng.uniform(0.0, 0.9)), shininess=float(rng.uniform(10.0, 250.0)))
        if rng.random() < 0.2:
            m.pattern = _pattern(rng, level=2)
        mats.append(m)
    glass = [rtc.Material(color=_colour(rng), diffuse=0.1, ambient=0.05, transparency=float(rng.uniform(0.6, 1.0)),
                          refractive_index=float(rng.uniform(1.1, 2.0)), reflectivity=float(rng.uniform(0.0, 0.9)))
             for _ in range(8)]
    mirror = [rtc.Material(color=_colour(rng), diffuse=0.4, reflectivity=float(rng.uniform(0.3, 1.0))) for _ in range(8)]

    def place(scale_lo=0.04, scale_hi=0.16, axis_aligned=False):
        t = M.scaling(*[float(v) for v in rng.uniform(scale_lo, scale_hi, 3)])
        if not axis_aligned:
            if rng.random() < 0.15:
                t = t.shear(*[float(v) for v in rng.uniform(-0.3, 0.3, 6)])
            t = t.rotate_x(float(rng.uniform(-math.pi, math.pi))).rotate_z(float(rng.uniform(-math.pi, math.pi)))
        return t.translate(float(rng.uniform(-9, 9)), float(rng.uniform(0.1, 5.0)), float(rng.uniform(-3, 14)))

    def leaf():
        r = rng.random()
        kind = int(rng.integers(0, 6))
        if r < 0.1 and kind < 2:
            m = glass[int(rng.integers(0, 8))]
        elif r < 0.2:
            m = mirror[int(rng.integers(0, 8))]
        else:
            m = mats[int(rng.integers(0, 64))]
        if kind == 0:
            # half of the spheres keep a diagonal inverse: candidates for the compact 64-byte leaf records
            return rtc.Sphere(place(axis_aligned=bool(rng.random() < 0.5)), m)
        if kind == 1:
            return rtc.Cube(place(), m)
        if kind == 2:
            return rtc.Cylinder(place(), m, float(rng.uniform(-1.0, 0.0)), float(rng.uniform(0.2, 1.0)), bool(rng.random() < 0.7))
        if kind == 3:
            lo = float(rng.uniform(-1.0, 0.3))
            # open cones cannot be bounded (an unclipped hit anywhere along the double cone when |a| < EPSILON,
            # shape.rs:1219-1225): a few of them only, they go to the list every ray tests in full
            return rtc.Cone(place(), m, lo, lo + float(rng.uniform(0.3, 1.0)), bool(rng.random() < 0.995))
        pts = rng.uniform(-1.0, 1.0, (int(rng.integers(1, 4)), 9))
        if kind == 4:
            return rtc.Group(transformation=place(0.08, 0.3), children=[rtc.TriangleMesh(pts)], material=m)
        nrm = rng.uniform(-1.0, 1.0, pts.shape).reshape(-1, 3, 3)
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        return rtc.Group(transformation=place(0.08, 0.3), children=[rtc.TriangleMesh(pts, nrm.reshape(-1, 9))], material=m)

    count = 0
    while count < n:
        if rng.random() < 0.02:
            # a cluster: a group (sometimes nested) with a transformation of its own
            g = rtc.Group(transformation=M.rotation_y(float(rng.uniform(0, math.pi))).scale(*[float(v) for v in rng.uniform(0.6, 1.4, 3)]))
            inner = rtc.Group(transformation=M.translation(float(rng.uniform(-1, 1)), 0.0, float(rng.uniform(-1, 1))))
            for _ in range(int(rng.integers(4, 12))):
                (inner if rng.random() < 0.4 else g).add_child(leaf())
                count += 1
            g.add_child(inner)
            objs.append(g)
        else:
            objs.append(leaf())
            count += 1
    lights = [rtc.PointLight((-12, 14, -10), (0.9, 0.9, 0.9))]
    cam = rtc.Camera(w, h, 1.0, rtc.compute_view_transformation((0.5, 3.2, -9), (0, 1.8, 3), (0, 1, 0)))
    return rtc.World(objs, lights), cam
