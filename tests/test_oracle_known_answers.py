"""Pins the CPU oracle against the reference's own known-answer unit tests.

Each test re-states one (or one table of) `#[test]` function(s) of the reference; the citation is
`path:line` relative to the reference root.  Point/Vector/Color `assert_eq!` in the reference go
through `equal()` (|d| < 1e-5, lib.rs:26-29); plain f64 `assert_eq!` is exact.
"""
import math

import numpy as np
import pytest

import orc
from orc import (CONE, CUBE, CYLINDER, GROUP, PLANE, SMOOTH_TRIANGLE, SPHERE, TEST_SHAPE, TRIANGLE, World,
                 default_world)

PI = math.pi
SQ2 = math.sqrt(2.0)
EPS = 1e-5
NAN = math.nan


def close(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return bool(np.all(np.abs(a - b) < EPS))


def norm(v):
    v = np.asarray(v, dtype=float)
    # tuple.rs:101-108
    m = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return v / m


# ------------------------------------------------------------------ matrix.rs

def test_matrix_determinants():
    # matrix.rs:565-578
    a = [[-2, -8, 3, 5], [-3, 1, 7, 3], [1, 2, -9, 6], [-6, 7, 7, -9]]
    assert orc.determinant(a) == -4071.0
    # matrix.rs:580-590
    assert orc.determinant([[6, 4, 4, 4], [5, 5, 7, 6], [4, -9, 3, -7], [9, 1, 7, -6]]) == -2120.0
    # matrix.rs:592-603
    sing = [[-4, 2, -2, -3], [9, 6, 2, 6], [0, -5, 1, -5], [0, 0, 0, 0]]
    assert orc.determinant(sing) == 0.0
    with pytest.raises(ValueError):
        orc.inverse(sing)


def test_matrix_inverses():
    # matrix.rs:605-627
    a = [[-5, 2, 6, -8], [1, -5, 1, 8], [7, 7, -6, -7], [1, -3, 7, 4]]
    b = orc.inverse(a)
    assert orc.determinant(a) == 532.0
    assert abs(b[3][2] - (-160.0 / 532.0)) < EPS and abs(b[2][3] - 105.0 / 532.0) < EPS
    assert close(b, [[0.21805, 0.45113, 0.24060, -0.04511], [-0.80827, -1.45677, -0.44361, 0.52068],
                     [-0.07895, -0.22368, -0.05263, 0.19737], [-0.52256, -0.81391, -0.30075, 0.30639]])
    # matrix.rs:629-645
    a = [[8, -5, 9, 2], [7, 5, 6, 1], [-6, 0, 9, 6], [-3, 0, -9, -4]]
    assert close(orc.inverse(a), [[-0.15385, -0.15385, -0.28205, -0.53846], [-0.07692, 0.12308, 0.02564, 0.03077],
                                  [0.35897, 0.35897, 0.43590, 0.92308], [-0.69231, -0.69231, -0.76923, -1.92308]])
    # matrix.rs:647-662
    a = [[9, 3, 0, 9], [-5, -2, -6, -3], [-4, 9, 6, 4], [-7, 6, 6, 2]]
    assert close(orc.inverse(a), [[-0.04074, -0.07778, 0.14444, -0.22222], [-0.07778, 0.03333, 0.36667, -0.33333],
                                  [-0.02901, -0.14630, -0.10926, 0.12963], [0.17778, 0.06667, -0.26667, 0.33333]])
    # matrix.rs:664-680
    a = np.array([[3, -9, 7, 3], [3, -8, 2, -9], [-4, 4, 4, 1], [-6, 5, -1, 1]], dtype=float)
    b = np.array([[8, 2, 2, 2], [3, -1, 7, 0], [7, 0, 5, 4], [6, -2, 0, 5]], dtype=float)
    c = orc.matmul(a, b)
    assert close(orc.matmul(c, orc.inverse(b)), a)


def test_matrix_transform_builders():
    # matrix.rs:682-702
    t = orc.translation(5, -3, 2)
    assert close(orc.mul_point(t, (-3, 4, 5)), (2, 1, 7))
    assert close(orc.mul_point(orc.inverse(t), (-3, 4, 5)), (-8, 7, 3))
    assert close(orc.mul_vector(t, (-3, 4, 5)), (-3, 4, 5))
    # matrix.rs:704-731
    s = orc.scaling(2, 3, 4)
    assert close(orc.mul_point(s, (-4, 6, 8)), (-8, 18, 32))
    assert close(orc.mul_vector(orc.inverse(s), (-4, 6, 8)), (-2, 2, 2))
    # matrix.rs:733-777
    v = SQ2 / 2
    assert close(orc.mul_point(orc.rotation_x(PI / 4), (0, 1, 0)), (0, v, v))
    assert close(orc.mul_point(orc.rotation_x(PI / 2), (0, 1, 0)), (0, 0, 1))
    assert close(orc.mul_point(orc.inverse(orc.rotation_x(PI / 4)), (0, 1, 0)), (0, v, -v))
    assert close(orc.mul_point(orc.rotation_y(PI / 4), (0, 0, 1)), (v, 0, v))
    assert close(orc.mul_point(orc.rotation_y(PI / 2), (0, 0, 1)), (1, 0, 0))
    assert close(orc.mul_point(orc.rotation_z(PI / 4), (0, 1, 0)), (-v, v, 0))
    assert close(orc.mul_point(orc.rotation_z(PI / 2), (0, 1, 0)), (-1, 0, 0))
    # matrix.rs:779-826
    for args, exp in [((1, 0, 0, 0, 0, 0), (5, 3, 4)), ((0, 1, 0, 0, 0, 0), (6, 3, 4)),
                      ((0, 0, 1, 0, 0, 0), (2, 5, 4)), ((0, 0, 0, 1, 0, 0), (2, 7, 4)),
                      ((0, 0, 0, 0, 1, 0), (2, 3, 6)), ((0, 0, 0, 0, 0, 1), (2, 3, 7))]:
        assert close(orc.mul_point(orc.shearing(*args), (2, 3, 4)), exp)
    # matrix.rs:828-866
    a, b, c = orc.rotation_x(PI / 2), orc.scaling(5, 5, 5), orc.translation(10, 5, 7)
    assert close(orc.mul_point(orc.matmul(orc.matmul(c, b), a), (1, 0, 1)), (15, 0, 7))
    assert close(orc.mul_point(orc.chain(a, b, c), (1, 0, 1)), (15, 0, 7))


def test_view_transformation():
    # transformation.rs:22-77
    assert close(orc.view_transform((0, 0, 0), (0, 0, -1), (0, 1, 0)), orc.identity())
    assert close(orc.view_transform((0, 0, 0), (0, 0, 1), (0, 1, 0)), orc.scaling(-1, 1, -1))
    assert close(orc.view_transform((0, 0, 8), (0, 0, 0), (0, 1, 0)), orc.translation(0, 0, -8))
    assert close(orc.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0)),
                 [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
                  [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]])


# ------------------------------------------------------------------ camera.rs

def test_camera_pixel_size():
    w = World()
    w.camera(200, 125, PI / 2)
    assert abs(w.camera_params()[2] - 0.01) < EPS  # camera.rs:113-117
    w.camera(125, 200, PI / 2)
    assert abs(w.camera_params()[2] - 0.01) < EPS  # camera.rs:119-123


def test_camera_rays():
    w = World()
    w.camera(201, 101, PI / 2)
    o, d = w.ray_for_pixel(100, 50)  # camera.rs:125-131
    assert close(o, (0, 0, 0)) and close(d, (0, 0, -1))
    o, d = w.ray_for_pixel(0, 0)  # camera.rs:133-139
    assert close(o, (0, 0, 0)) and close(d, (0.66519, 0.33259, -0.66851))
    w.camera(201, 101, PI / 2, orc.chain(orc.translation(0, -2, 5), orc.rotation_y(PI / 4)))
    o, d = w.ray_for_pixel(100, 50)  # camera.rs:141-154
    assert close(o, (0, 2, -5)) and close(d, (SQ2 / 2, 0, -SQ2 / 2))


def test_render_default_world():
    # camera.rs:156-165
    w, *_ = default_world()
    w.camera(11, 11, PI / 2, orc.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    img = w.render(threads=2)
    assert close(img[5, 5], (0.38066, 0.47583, 0.2855))
    # Q1: last row / column never rendered (camera.rs:72,78)
    assert np.all(img[10, :, :] == 0.0) and np.all(img[:, 10, :] == 0.0)


# ------------------------------------------------------------------ intersection.rs

def test_ray_position_and_transform():
    # intersection.rs:241-248, 312-328 via a sphere carrying the transform
    w = World()
    s = w.shape(SPHERE, transform=orc.inverse(orc.translation(3, 4, 5)))
    # Ray::transform(matrix) == object-space ray of a shape whose inverse is `matrix`
    p = orc.mul_point(orc.translation(3, 4, 5), (1, 2, 3))
    assert close(p, (4, 6, 8))
    assert close(orc.mul_vector(orc.scaling(2, 3, 4), (0, 1, 0)), (0, 3, 0))
    assert s == 0


def test_hit_selection():
    # intersection.rs:270-310
    assert orc.hit_index([1, 2]) == 0
    assert orc.hit_index([-1, 1]) == 1
    assert orc.hit_index([-2, -1]) == -1
    assert orc.hit_index([5, 7, -3, 2]) == 3


def test_prepare_computations_basics():
    w = World()
    s = w.shape(SPHERE)
    c = w.prepare_computations((0, 0, -5), (0, 0, 1), [(4, s)], 0)  # intersection.rs:330-355
    assert c.t == 4 and c.shape == s
    assert close(c.point, (0, 0, -1)) and close(c.eye, (0, 0, -1)) and close(c.normal, (0, 0, -1))
    assert not c.inside
    c = w.prepare_computations((0, 0, 0), (0, 0, 1), [(1, s)], 0)  # intersection.rs:357-369
    assert close(c.point, (0, 0, 1)) and close(c.eye, (0, 0, -1)) and c.inside and close(c.normal, (0, 0, -1))
    s2 = w.shape(SPHERE, transform=orc.translation(0, 0, 1))
    c = w.prepare_computations((0, 0, -5), (0, 0, 1), [(5, s2)], 0)  # intersection.rs:371-382
    assert c.over_point[2] < -EPS / 2 and c.point[2] > c.over_point[2]
    p = w.shape(PLANE)
    v = 2.0 / 2.0  # sic: the reference uses 2.0_f64 / 2. here (intersection.rs:387)
    c = w.prepare_computations((0, 1, -1), (0, -v, v), [(SQ2, p)], 0)  # intersection.rs:384-394
    assert close(c.reflect, (0, v, v))
    g = w.shape(SPHERE, transform=orc.translation(0, 0, 1),
                material=w.material(transparency=1.0, refractive_index=1.5))
    c = w.prepare_computations((0, 0, -5), (0, 0, 1), [(5, g)], 0)  # intersection.rs:445-454
    assert c.under_point[2] > EPS / 2 and c.point[2] < c.under_point[2]


def test_n1_n2_containers():
    # intersection.rs:396-443
    w = World()
    a = w.shape(SPHERE, transform=orc.scaling(2, 2, 2), material=w.material(refractive_index=1.5))
    b = w.shape(SPHERE, transform=orc.translation(0, 0, -0.25), material=w.material(refractive_index=2.0))
    c = w.shape(SPHERE, transform=orc.translation(0, 0, 0.25), material=w.material(refractive_index=2.5))
    xs = [(2, a), (2.75, b), (3.25, c), (4.75, b), (5.25, c), (6, a)]
    n1s = [1.0, 1.5, 2.0, 2.5, 2.5, 1.5]
    n2s = [1.5, 2.0, 2.5, 2.5, 1.5, 1.0]
    for i in range(6):
        comps = w.prepare_computations((0, 0, -4), (0, 0, 1), xs, i)
        assert comps.n1 == n1s[i] and comps.n2 == n2s[i]


def test_schlick():
    w = World()
    glass = w.shape(SPHERE, material=w.material(transparency=1.0, refractive_index=1.5))
    v = 2.0 / 2.0  # sic (intersection.rs:459)
    c = w.prepare_computations((0, 0, v), (0, 1, 0), [(-v, glass), (v, glass)], 1)
    assert c.schlick() == 1.0  # intersection.rs:456-468
    c = w.prepare_computations((0, 0, 0), (0, 1, 0), [(-1, glass), (1, glass)], 1)
    assert abs(c.schlick() - 0.04) < EPS  # intersection.rs:470-481
    c = w.prepare_computations((0, 0.99, -2), (0, 0, 1), [(1.8589, glass)], 0)
    assert abs(c.schlick() - 0.48873) < EPS  # intersection.rs:483-491


# ------------------------------------------------------------------ shape.rs

def test_sphere_intersections():
    w = World()
    s = w.shape(SPHERE)
    ts = lambda o, d: [x[0] for x in w.intersect(s, o, d)]
    assert ts((0, 0, -5), (0, 0, 1)) == [4.0, 6.0]  # shape.rs:314-322
    assert ts((0, 1, -5), (0, 0, 1)) == [5.0, 5.0]  # shape.rs:324-332
    assert ts((0, 2, -5), (0, 0, 1)) == []  # shape.rs:334-340
    assert ts((0, 0, 0), (0, 0, 1)) == [-1.0, 1.0]  # shape.rs:342-350
    assert ts((0, 0, 5), (0, 0, 1)) == [-6.0, -4.0]  # shape.rs:352-360
    s2 = w.shape(SPHERE, transform=orc.scaling(2, 2, 2))
    assert [x[0] for x in w.intersect(s2, (0, 0, -5), (0, 0, 1))] == [3.0, 7.0]  # shape.rs:385-393
    s3 = w.shape(SPHERE, transform=orc.translation(5, 0, 0))
    assert w.intersect(s3, (0, 0, -5), (0, 0, 1)) == []  # shape.rs:395-401


def test_sphere_normals():
    w = World()
    s = w.shape(SPHERE)
    assert close(w.normal_at(s, (1, 0, 0)), (1, 0, 0))  # shape.rs:403-422
    assert close(w.normal_at(s, (0, 1, 0)), (0, 1, 0))
    assert close(w.normal_at(s, (0, 0, 1)), (0, 0, 1))
    v = math.sqrt(3) / 3
    n = w.normal_at(s, (v, v, v))  # shape.rs:424-438
    assert close(n, (v, v, v)) and close(n, norm(n))
    s2 = w.shape(SPHERE, transform=orc.translation(0, 1, 0))
    f = math.sqrt(0.5)
    assert close(w.normal_at(s2, (0, 1.70711, -f)), (0, f, -f))  # shape.rs:440-447
    s3 = w.shape(SPHERE, transform=orc.chain(orc.rotation_z(PI / 5), orc.scaling(1, 0.5, 1)))
    assert close(w.normal_at(s3, (0, SQ2 / 2, -SQ2 / 2)), (0, 0.97014, -0.24254))  # shape.rs:449-456


def test_plane():
    w = World()
    p = w.shape(PLANE)
    for pt in [(0, 0, 0), (10, 0, -10), (-5, 0, 150)]:  # shape.rs:588-597
        assert close(w.local_normal_at(p, pt), (0, 1, 0))
    assert w.local_intersect(p, (0, 10, 0), (0, 0, 1)) == []  # shape.rs:599-605
    assert w.local_intersect(p, (0, 0, 0), (0, 0, 1)) == []  # shape.rs:607-613
    assert w.local_intersect(p, (0, 1, 0), (0, -1, 0)) == [1.0]  # shape.rs:615-624
    assert w.local_intersect(p, (0, -1, 0), (0, 1, 0)) == [1.0]  # shape.rs:626-635


def test_cube():
    w = World()
    c = w.shape(CUBE)
    hits = [((5, 0.5, 0), (-1, 0, 0), 4, 6), ((-5, 0.5, 0), (1, 0, 0), 4, 6), ((0.5, 5, 0), (0, -1, 0), 4, 6),
            ((0.5, -5, 0), (0, 1, 0), 4, 6), ((0.5, 0, 5), (0, 0, -1), 4, 6), ((0.5, 0, -5), (0, 0, 1), 4, 6),
            ((0, 0.5, 0), (0, 0, 1), -1, 1)]
    for o, d, t1, t2 in hits:  # shape.rs:791-811
        assert w.local_intersect(c, o, d) == [float(t1), float(t2)]
    misses = [((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8018, 0.2673, 0.5345)),
              ((0, 0, -2), (0.5345, 0.8018, 0.2673)), ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)),
              ((2, 2, 0), (-1, 0, 0))]
    for o, d in misses:  # shape.rs:813-830
        assert w.local_intersect(c, o, d) == []
    normals = [((1, 0.5, -0.8), (1, 0, 0)), ((-1, -0.2, 0.9), (-1, 0, 0)), ((0.4, 1, -0.1), (0, 1, 0)),
               ((0.3, -1, 0.7), (0, -1, 0)), ((-0.6, 0.3, 1), (0, 0, 1)), ((0.4, 0.4, -1), (0, 0, -1)),
               ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))]
    for p, n in normals:  # shape.rs:832-850
        assert close(w.local_normal_at(c, p), n)


def test_cylinder():
    w = World()
    cyl = w.shape(CYLINDER)
    for o, d in [((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, -5), (1, 1, 1))]:  # shape.rs:1039-1053
        assert w.local_intersect(cyl, o, norm(d)) == []
    for o, d, t0, t1 in [((1, 0, -5), (0, 0, 1), 5, 5), ((0, 0, -5), (0, 0, 1), 4, 6),
                         ((0.5, 0, -5), (0.1, 1, 1), 6.80798, 7.08872)]:  # shape.rs:1055-1078
        ts = w.local_intersect(cyl, o, norm(d))
        assert len(ts) == 2 and abs(ts[0] - t0) < EPS and abs(ts[1] - t1) < EPS
    for p, n in [((1, 0, 0), (1, 0, 0)), ((0, 5, -1), (0, 0, -1)), ((0, -2, 1), (0, 0, 1)),
                 ((-1, 1, 0), (-1, 0, 0))]:  # shape.rs:1080-1094
        assert close(w.local_normal_at(cyl, p), n)
    trunc = w.shape(CYLINDER, minimum=1, maximum=2)
    for o, d, cnt in [((0, 1.5, 0), (0.1, 1, 0), 0), ((0, 3, -5), (0, 0, 1), 0), ((0, 0, -5), (0, 0, 1), 0),
                      ((0, 2, -5), (0, 0, 1), 0), ((0, 1, -5), (0, 0, 1), 0),
                      ((0, 1.5, -2), (0, 0, 1), 2)]:  # shape.rs:1103-1122
        assert len(w.local_intersect(trunc, o, norm(d))) == cnt
    closed = w.shape(CYLINDER, minimum=1, maximum=2, closed=True)
    for o, d, cnt in [((0, 3, 0), (0, -1, 0), 2), ((0, 3, -2), (0, -1, 2), 2), ((0, 4, -2), (0, -1, 1), 2),
                      ((0, 0, -2), (0, 1, 2), 2), ((0, -1, -2), (0, 1, 1), 2)]:  # shape.rs:1130-1148
        assert len(w.local_intersect(closed, o, norm(d))) == cnt
    for p, n in [((0, 1, 0), (0, -1, 0)), ((0.5, 1, 0), (0, -1, 0)), ((0, 1, 0.5), (0, -1, 0)),
                 ((0, 2, 0), (0, 1, 0)), ((0.5, 2, 0), (0, 1, 0)), ((0, 2, 0.5), (0, 1, 0))]:  # shape.rs:1150-1167
        assert close(w.local_normal_at(closed, p), n)


def test_cone():
    w = World()
    cone = w.shape(CONE)
    for o, d, t0, t1 in [((0, 0, -5), (0, 0, 1), 5, 5), ((0, 0, -5), (1, 1, 1), 8.66025, 8.66025),
                         ((1, 1, -5), (-0.5, -1, 1), 4.55006, 49.44994)]:  # shape.rs:1356-1384
        ts = w.local_intersect(cone, o, norm(d))
        assert len(ts) == 2 and abs(ts[0] - t0) < EPS and abs(ts[1] - t1) < EPS
    ts = w.local_intersect(cone, (0, 0, -1), norm((0, 1, 1)))  # shape.rs:1386-1394
    assert len(ts) == 1 and abs(ts[0] - 0.35355) < EPS
    capped = w.shape(CONE, minimum=-0.5, maximum=0.5, closed=True)
    for o, d, cnt in [((0, 0, -5), (0, 1, 0), 0), ((0, 0, -0.25), (0, 1, 1), 2),
                      ((0, 0, -0.25), (0, 1, 0), 4)]:  # shape.rs:1396-1418
        assert len(w.local_intersect(capped, o, norm(d))) == cnt
    for p, n in [((0, 0, 0), (0, 0, 0)), ((1, 1, 1), (1, -SQ2, 1)), ((-1, -1, 0), (-1, 1, 0))]:  # shape.rs:1420-1436
        assert close(w.local_normal_at(cone, p), n)


TRI = ((0, 1, 0), (-1, 0, 0), (1, 0, 0))
TRI_N = ((0, 1, 0), (-1, 0, 0), (1, 0, 0))


def test_triangle():
    w = World()
    t = w.triangle(*TRI)
    for p in [(0, 0.5, 0), (-0.5, 0.75, 0), (0.5, 0.25, 0)]:  # shape.rs:1563-1590 (normal (0,0,-1))
        assert close(w.local_normal_at(t, p), (0, 0, -1))
    assert w.local_intersect(t, (0, -1, -2), (0, 1, 0)) == []  # shape.rs:1592-1603
    assert w.local_intersect(t, (1, 1, -2), (0, 0, 1)) == []  # shape.rs:1605-1616
    assert w.local_intersect(t, (-1, 1, -2), (0, 0, 1)) == []  # shape.rs:1618-1630
    assert w.local_intersect(t, (0, -1, -2), (0, 0, 1)) == []  # shape.rs:1632-1644
    assert w.local_intersect(t, (0, 0.5, -2), (0, 0, 1)) == [2.0]  # shape.rs:1646-1658


def test_smooth_triangle():
    w = World()
    t = w.triangle(*TRI, normals=TRI_N)
    xs = w.intersect(t, (-0.2, 0.3, -2), (0, 0, 1))  # shape.rs:1822-1839
    assert abs(xs[0][2] - 0.45) < EPS and abs(xs[0][3] - 0.25) < EPS
    assert close(w.normal_at(t, (0, 0, 0), 0.45, 0.25), (-0.5547, 0.83205, 0))  # shape.rs:1841-1861
    c = w.prepare_computations((-0.2, 0.3, -2), (0, 0, 1), [(1, t, 0.45, 0.25)], 0)  # shape.rs:1863-1881
    assert close(c.normal, (-0.5547, 0.83205, 0))


def test_groups():
    w = World()
    g0 = w.shape(GROUP, material=w.material())
    assert w.intersect(g0, (0, 0, 0), (0, 0, 1)) == []  # shape.rs:2092-2098
    g = w.shape(GROUP, material=w.material())
    s1 = w.shape(SPHERE, parent=g)
    s2 = w.shape(SPHERE, parent=g, transform=orc.translation(0, 0, -3))
    w.shape(SPHERE, parent=g, transform=orc.translation(5, 0, 0))
    xs = w.intersect(g, (0, 0, -5), (0, 0, 1))  # shape.rs:2100-2126
    assert [x[1] for x in xs] == [s2, s2, s1, s1]
    g2 = w.shape(GROUP, transform=orc.scaling(2, 2, 2), material=w.material())
    w.shape(SPHERE, parent=g2, transform=orc.translation(5, 0, 0))
    assert len(w.intersect(g2, (10, 0, -10), (0, 0, 1))) == 2  # shape.rs:2128-2142


def _nested(w, inner_scale):
    g1 = w.shape(GROUP, transform=orc.rotation_y(PI / 2), material=w.material())
    g2 = w.shape(GROUP, parent=g1, transform=orc.scaling(*inner_scale), material=w.material())
    s = w.shape(SPHERE, parent=g2, transform=orc.translation(5, 0, 0))
    return s


def test_parent_chain_conversions():
    w = World()
    s = _nested(w, (2, 2, 2))
    assert close(w.world_to_object(s, (-2, 0, -10)), (0, 0, -1))  # shape.rs:2555-2578
    s = _nested(w, (1, 2, 3))
    v = math.sqrt(3) / 3
    assert close(w.normal_to_world(s, (v, v, v)), (0.28571, 0.42857, -0.85714))  # shape.rs:2580-2606
    assert close(w.normal_at(s, (1.7321, 1.1547, -5.5774)), (0.2857, 0.42854, -0.85716))  # shape.rs:2608-2631


def test_test_shape_normals():
    w = World()
    f = math.sqrt(0.5)
    s = w.shape(TEST_SHAPE, transform=orc.translation(0, 1, 0))
    assert close(w.normal_at(s, (0, 1.70711, -f)), (0, f, -f))  # shape.rs:2531-2538
    s = w.shape(TEST_SHAPE, transform=orc.chain(orc.rotation_z(PI / 2), orc.scaling(1, 0.5, 1)))
    assert close(w.normal_at(s, (0, SQ2 / 2, -SQ2 / 2)), (0, 0.97014, -0.24254))  # shape.rs:2540-2547


# ------------------------------------------------------------------ material.rs

def _lighting_case(eye, light_pos, intensity=1.0):
    w = World()
    m = w.material()
    s = w.shape(SPHERE, material=m)
    l = w.point_light(light_pos, (1, 1, 1))
    return w.lighting(m, s, l, (0, 0, 0), eye, (0, 0, -1), intensity)


def test_phong_lighting():
    v = SQ2 / 2
    assert close(_lighting_case((0, 0, -1), (0, 0, -10)), (1.9, 1.9, 1.9))  # material.rs:118-130
    assert close(_lighting_case((0, v, -v), (0, 0, -10)), (1.0, 1.0, 1.0))  # material.rs:132-145
    assert close(_lighting_case((0, 0, -1), (0, 10, -10)), (0.7364,) * 3)  # material.rs:147-160
    assert close(_lighting_case((0, -v, -v), (0, 10, -10)), (1.6364,) * 3)  # material.rs:162-176
    assert close(_lighting_case((0, 0, -1), (0, 0, 10)), (0.1,) * 3)  # material.rs:178-190
    assert close(_lighting_case((0, 0, -1), (0, 0, -10), 0.0), (0.1,) * 3)  # material.rs:192-204


def test_lighting_with_pattern():
    # material.rs:206-244
    w = World()
    white = w.pattern(orc.PAT_SOLID, a=(1, 1, 1))
    black = w.pattern(orc.PAT_SOLID, a=(0, 0, 0))
    stripe = w.pattern(orc.PAT_STRIPE, sub_a=white, sub_b=black)
    m = w.material(ambient=1.0, diffuse=0.0, specular=0.0, pattern=stripe)
    s = w.shape(SPHERE, material=m)
    l = w.point_light((0, 0, -10))
    assert close(w.lighting(m, s, l, (0.9, 0, 0), (0, 0, -1), (0, 0, -1)), (1, 1, 1))
    assert close(w.lighting(m, s, l, (1.1, 0, 0), (0, 0, -1), (0, 0, -1)), (0, 0, 0))


def test_lighting_intensity_attenuation():
    # material.rs:259-301
    w = World()
    l = w.point_light((0, 0, -10))
    m = w.material(ambient=0.1, diffuse=0.9, specular=0.0, color=(1, 1, 1))
    s = w.shape(SPHERE, material=m)
    for inten, exp in [(1.0, 1.0), (0.5, 0.55), (0.0, 0.1)]:
        assert close(w.lighting(m, s, l, (0, 0, -1), (0, 0, -1), (0, 0, -1), inten), (exp,) * 3)


# ------------------------------------------------------------------ light.rs

def test_point_light_intensity_at():
    # light.rs:149-168
    w, _, _, light = default_world()
    for p, exp in [((0, 1.0001, 0), 1.0), ((-1.0001, 0, 0), 1.0), ((0, 0, -1.0001), 1.0), ((0, 0, 1.0001), 0.0),
                   ((1.0001, 0, 0), 0.0), ((0, -1.0001, 0), 0.0), ((0, 0, 0), 0.0)]:
        assert abs(w.intensity_at(light, p) - exp) < EPS


def test_area_light():
    w = World()
    l = w.area_light((0, 0, 0), (2, 0, 0), 4, (0, 0, 1), 2)  # light.rs:170-184
    pos, uvec, vvec, samples = w.light_info(l)
    assert close(uvec, (0.5, 0, 0)) and close(vvec, (0, 0, 0.5)) and samples == 8 and close(pos, (1, 0, 0.5))
    for u, v, exp in [(0, 0, (0.25, 0, 0.25)), (1, 0, (0.75, 0, 0.25)), (0, 1, (0.25, 0, 0.75)),
                      (2, 0, (1.25, 0, 0.25)), (3, 1, (1.75, 0, 0.75))]:  # light.rs:186-205
        assert close(w.area_light_point(l, u, v), exp)
    w2, *_ = default_world()
    al = w2.area_light((-0.5, -0.5, -5), (1, 0, 0), 2, (0, 1, 0), 2)  # light.rs:207-227
    for p, exp in [((0, 0, 2), 0.0), ((1, -1, 2), 0.25), ((1.5, 0, 2), 0.5), ((1.25, 1.25, 3), 0.75),
                   ((0, 0, -2), 1.0)]:
        assert w2.intensity_at(al, p) == exp


# ------------------------------------------------------------------ world.rs

def test_intersect_world():
    w, *_ = default_world()
    assert [x[0] for x in w.intersect(-1, (0, 0, -5), (0, 0, 1))] == [4.0, 4.5, 5.5, 6.0]  # world.rs:237-247


def test_shade_hit():
    w, s1, s2, _ = default_world()
    c = w.prepare_computations((0, 0, -5), (0, 0, 1), [(4, s1)], 0)
    assert close(w.shade_hit(c), (0.38066, 0.47583, 0.2855))  # world.rs:249-260
    w, s1, s2, _ = default_world(light_pos=(0, 0.25, 0))
    c = w.prepare_computations((0, 0, 0), (0, 0, 1), [(0.5, s2)], 0)
    assert close(w.shade_hit(c), (0.90498,) * 3)  # world.rs:262-282
    w = World()
    w.point_light((0, 0, -10))
    w.shape(SPHERE)
    sb = w.shape(SPHERE, transform=orc.translation(0, 0, 10))
    c = w.prepare_computations((0, 0, 5), (0, 0, 1), [(4, sb)], 0)
    assert close(w.shade_hit(c), (0.1, 0.1, 0.1))  # world.rs:284-302


def test_color_at():
    w, *_ = default_world()
    assert close(w.color_at((0, 0, -5), (0, 1, 0)), (0, 0, 0))  # world.rs:304-311
    assert close(w.color_at((0, 0, -5), (0, 0, 1)), (0.38066, 0.47583, 0.2855))  # world.rs:313-320
    w, *_ = default_world(m1=dict(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0),
                          m2=dict(ambient=1.0))
    assert close(w.color_at((0, 0, 0.75), (0, 0, -1)), (1, 1, 1))  # world.rs:322-342


def _world_with_reflective_plane(reflectivity=0.5, **m1):
    w, s1, s2, _ = default_world(**m1)
    pm = w.material(reflectivity=reflectivity)
    plane = w.shape(PLANE, transform=orc.translation(0, -1, 0), material=pm)
    return w, s1, s2, plane


def test_reflection():
    w, s1, s2, _ = default_world(m1=dict(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0))
    c = w.prepare_computations((0, 0, 0), (0, 0, 1), [(1, s2)], 0)
    assert close(w.reflected_color(c), (0, 0, 0))  # world.rs:344-378
    v = SQ2 / 2
    w, s1, s2, plane = _world_with_reflective_plane()
    c = w.prepare_computations((0, 0, -3), (0, -v, v), [(SQ2, plane)], 0)
    assert close(w.reflected_color(c), (0.19033, 0.23792, 0.14274))  # world.rs:380-421
    assert close(w.shade_hit(c), (0.87675, 0.92434, 0.82917))  # world.rs:423-461
    assert close(w.reflected_color(c, remaining=0), (0, 0, 0))  # world.rs:507-529


def test_mutual_mirrors_terminate():
    # world.rs:463-505
    w = World()
    w.point_light((0, 0, 0))
    w.shape(SPHERE, material=w.material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2))
    w.shape(SPHERE, transform=orc.scaling(0.5, 0.5, 0.5))
    w.shape(PLANE, transform=orc.translation(0, -1, 0), material=w.material(reflectivity=1.0))
    w.shape(PLANE, transform=orc.translation(0, 1, 0), material=w.material(reflectivity=1.0))
    c = w.color_at((0, 0, 0), (0, 1, 0))
    assert np.all(np.isfinite(c))


def test_refraction():
    w, s1, s2, _ = default_world()
    c = w.prepare_computations((0, 0, -5), (0, 0, 1), [(4, s1), (6, s1)], 0)
    assert close(w.refracted_color(c), (0, 0, 0))  # world.rs:531-543
    glass1 = dict(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, transparency=1.0, refractive_index=1.5)
    w, s1, s2, _ = default_world(m1=glass1)
    c = w.prepare_computations((0, 0, -5), (0, 0, 1), [(4, s1), (6, s1)], 0)
    assert close(w.refracted_color(c, remaining=0), (0, 0, 0))  # world.rs:545-564
    v = SQ2 / 2
    c = w.prepare_computations((0, 0, v), (0, 1, 0), [(-v, s1), (v, s1)], 1)
    assert close(w.refracted_color(c), (0, 0, 0))  # world.rs:566-588 (total internal reflection)


def test_refracted_color_with_test_pattern():
    # world.rs:590-628
    w = World()
    w.point_light((-10, 10, -10))
    tp = w.pattern(orc.PAT_TEST)
    a = w.shape(SPHERE, material=w.material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0,
                                            pattern=tp))
    b = w.shape(SPHERE, transform=orc.scaling(0.5, 0.5, 0.5),
                material=w.material(transparency=1.0, refractive_index=1.5))
    xs = [(-0.9899, a), (-0.4899, b), (0.4899, b), (0.9899, a)]
    c = w.prepare_computations((0, 0, 0.1), (0, 1, 0), xs, 2)
    assert close(w.refracted_color(c), (0, 0.99887, 0.04722))


def _floor_world(**floor_kw):
    w, s1, s2, _ = default_world()
    floor = w.shape(PLANE, transform=orc.translation(0, -1, 0), material=w.material(**floor_kw))
    w.shape(SPHERE, transform=orc.translation(0, -3.5, -0.5), material=w.material(color=(1, 0, 0), ambient=0.5))
    return w, floor


def test_shade_hit_transparent_and_schlick():
    v = SQ2 / 2
    w, floor = _floor_world(transparency=0.5, refractive_index=1.5)
    c = w.prepare_computations((0, 0, -3), (0, -v, v), [(SQ2, floor)], 0)
    assert close(w.shade_hit(c), (0.93642, 0.68642, 0.68642))  # world.rs:630-660
    w, floor = _floor_world(reflectivity=0.5, transparency=0.5, refractive_index=1.5)
    c = w.prepare_computations((0, 0, -3), (0, -v, v), [(SQ2, floor)], 0)
    assert close(w.shade_hit(c), (0.93391, 0.69643, 0.69243))  # world.rs:662-695


def test_is_shadowed():
    # world.rs:697-712
    w, *_ = default_world()
    lp = (-10, -10, -10)
    for p, exp in [((-10, -10, 10), False), ((10, 10, 10), True), ((-20, -20, -20), False),
                   ((-5, -5, -5), False)]:
        assert w.is_shadowed(p, lp) == exp


# ------------------------------------------------------------------ pattern.rs

def test_patterns():
    w = World()
    white = w.pattern(orc.PAT_SOLID, a=(1, 1, 1))
    black = w.pattern(orc.PAT_SOLID, a=(0, 0, 0))
    stripe = w.pattern(orc.PAT_STRIPE, sub_a=white, sub_b=black)
    W, B = (1, 1, 1), (0, 0, 0)
    for p, exp in [((0, 1, 0), W), ((0, 2, 0), W), ((0, 0, 1), W), ((0, 0, 2), W), ((0, 0, 0), W),
                   ((0.9, 0, 0), W), ((1, 0, 0), B), ((-0.1, 0, 0), B), ((-1, 0, 0), B),
                   ((-1.1, 0, 0), W)]:  # pattern.rs:135-162
        assert close(w.pattern_at(stripe, p), exp)
    s2 = w.shape(SPHERE, transform=orc.scaling(2, 2, 2))
    s1 = w.shape(SPHERE)
    assert close(w.pattern_at_shape(stripe, s2, (1.5, 0, 0)), W)  # pattern.rs:164-172
    stripe2 = w.pattern(orc.PAT_STRIPE, sub_a=white, sub_b=black, transform=orc.scaling(2, 2, 2))
    assert close(w.pattern_at_shape(stripe2, s1, (1.5, 0, 0)), W)  # pattern.rs:174-186
    stripe3 = w.pattern(orc.PAT_STRIPE, sub_a=white, sub_b=black, transform=orc.translation(0.5, 0, 0))
    assert close(w.pattern_at_shape(stripe3, s2, (2.5, 0, 0)), W)  # pattern.rs:188-198
    grad = w.pattern(orc.PAT_GRADIENT, a=W, b=B)
    for x, e in [(0, 1.0), (0.25, 0.75), (0.5, 0.5), (0.75, 0.25)]:  # pattern.rs:256-272
        assert close(w.pattern_at(grad, (x, 0, 0)), (e,) * 3)
    ring = w.pattern(orc.PAT_RING, a=W, b=B)
    for p, exp in [((0, 0, 0), W), ((1, 0, 0), B), ((0, 0, 1), B), ((0.708, 0, 0.708), B)]:  # pattern.rs:328-339
        assert close(w.pattern_at(ring, p), exp)
    chk = w.pattern(orc.PAT_CHECKER, sub_a=white, sub_b=black)
    for p, exp in [((0, 0, 0), W), ((0.99, 0, 0), W), ((1.01, 0, 0), B), ((0, 0.99, 0), W), ((0, 1.01, 0), B),
                   ((0, 0, 0.99), W), ((0, 0, 1.01), B)]:  # pattern.rs:401-423
        assert close(w.pattern_at(chk, p), exp)
    rad = w.pattern(orc.PAT_RADIAL, a=W, b=B)
    for x, e in [(0, 1.0), (0.25, 0.75), (0.5, 0.5)]:  # pattern.rs:483-496
        assert close(w.pattern_at(rad, (x, 0, 0)), (e,) * 3)
    assert close(w.pattern_at(white, (1, 0, 0)), W)  # pattern.rs:547-554
    tp = w.pattern(orc.PAT_TEST)
    assert close(w.pattern_at_shape(tp, s2, (2, 3, 4)), (1, 1.5, 2))  # pattern.rs:703-711
    tp2 = w.pattern(orc.PAT_TEST, transform=orc.scaling(2, 2, 2))
    assert close(w.pattern_at_shape(tp2, s1, (2, 3, 4)), (1, 1.5, 2))  # pattern.rs:713-722
    tp3 = w.pattern(orc.PAT_TEST, transform=orc.translation(0.5, 1, 1.5))
    assert close(w.pattern_at_shape(tp3, s2, (2.5, 3, 3.5)), (0.75, 0.5, 0.25))  # pattern.rs:724-733


# ------------------------------------------------------------------ canvas.rs

def test_ppm():
    img = np.zeros((3, 5, 3))
    ppm = orc.canvas_to_ppm(img)
    lines = ppm.split("\n")
    assert lines[:3] == ["P3", "5 3", "255"]  # canvas.rs:115-123
    assert ppm.endswith("\n")  # canvas.rs:173-179
    img[0, 0] = (1.5, 0, 0)
    img[1, 2] = (0, 0.5, 0)
    img[2, 4] = (-0.5, 0, 1.0)
    lines = orc.canvas_to_ppm(img).split("\n")  # canvas.rs:125-140
    assert lines[3] == "255 0 0 0 0 0 0 0 0 0 0 0 0 0 0"
    assert lines[4] == "0 0 0 0 0 0 0 128 0 0 0 0 0 0 0"
    assert lines[5] == "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"
    img = np.zeros((2, 10, 3))
    img[:, :] = (1.0, 0.8, 0.6)
    lines = orc.canvas_to_ppm(img).split("\n")  # canvas.rs:142-171
    assert lines[3] == "255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204"
    assert lines[4] == "153 255 204 153 255 204 153 255 204 153 255 204 153"
    assert lines[5] == lines[3] and lines[6] == lines[4]


def test_scale_and_clamp():
    # canvas.rs:86-89 + Rust `as u8` saturation
    assert orc.scale_and_clamp(1.5) == 255 and orc.scale_and_clamp(-0.5) == 0 and orc.scale_and_clamp(0.5) == 128
    assert orc.scale_and_clamp(float("nan")) == 0


# ----------------------------------------------------------------------------- shape.rs:2146-2411 (csg)

def test_csg_operation_rule():
    # shape.rs:2322-2355
    U, I, D = orc.CSG_UNION, orc.CSG_INTERSECTION, orc.CSG_DIFFERENCE
    examples = [
        (U, 1, 1, 1, 0), (U, 1, 1, 0, 1), (U, 1, 0, 1, 0), (U, 1, 0, 0, 1),
        (U, 0, 1, 1, 0), (U, 0, 1, 0, 0), (U, 0, 0, 1, 1), (U, 0, 0, 0, 1),
        (I, 1, 1, 1, 1), (I, 1, 1, 0, 0), (I, 1, 0, 1, 1), (I, 1, 0, 0, 0),
        (I, 0, 1, 1, 1), (I, 0, 1, 0, 1), (I, 0, 0, 1, 0), (I, 0, 0, 0, 0),
        (D, 1, 1, 1, 0), (D, 1, 1, 0, 1), (D, 1, 0, 1, 0), (D, 1, 0, 0, 1),
        (D, 0, 1, 1, 1), (D, 0, 1, 0, 1), (D, 0, 0, 1, 0), (D, 0, 0, 0, 0),
    ]
    for op, lhit, inl, inr, expected in examples:
        assert bool(orc.lib().orc_csg_intersection_allowed(op, lhit, inl, inr)) == bool(expected)


def test_csg_filter_list_of_intersections():
    # shape.rs:2357-2382: sphere (left) and cube (right), records at t = 1 (s1), 2 (s2), 3 (s1), 4 (s2)
    for op, x0, x1 in ((orc.CSG_UNION, 0, 3), (orc.CSG_INTERSECTION, 1, 2), (orc.CSG_DIFFERENCE, 0, 1)):
        w = orc.World()
        c = w.csg(op)
        s1 = w.shape(orc.SPHERE, parent=c)
        s2 = w.shape(orc.CUBE, parent=c)
        assert w.csg_filter(c, [(1, s1), (2, s2), (3, s1), (4, s2)]) == [x0, x1]


def test_csg_create_and_includes():
    # shape.rs:2297-2320 (left / right / parents) and the includes() rules :136-138, :2023-2030, :2278-2280
    w = orc.World()
    c = w.csg(orc.CSG_UNION)
    s1 = w.shape(orc.SPHERE, parent=c)
    g = w.shape(orc.GROUP, parent=c)
    s2 = w.shape(orc.CUBE, parent=g)
    inner = w.csg(orc.CSG_DIFFERENCE, parent=g)
    s3 = w.shape(orc.SPHERE, parent=inner)
    s4 = w.shape(orc.CONE, parent=inner)
    other = w.shape(orc.SPHERE)
    assert w.includes(s1, s1) and not w.includes(s1, s2)
    assert w.includes(g, s2) and w.includes(g, s4) and not w.includes(g, s1)
    assert w.includes(inner, s3) and w.includes(inner, s4) and not w.includes(inner, s2)
    assert all(w.includes(c, s) for s in (s1, s2, s3, s4)) and not w.includes(c, other)
    # a Csg has exactly a left and a right child
    assert orc.lib().orc_add_shape(w.h, orc.SPHERE, c, orc._d16(orc.identity()), w.material(), -orc.INF, orc.INF, 0) < 0


def test_csg_ray_misses_and_hits():
    # shape.rs:2384-2392
    w = orc.World()
    c = w.csg(orc.CSG_UNION)
    w.shape(orc.SPHERE, parent=c)
    w.shape(orc.CUBE, parent=c)
    assert w.local_intersect(c, (0, 2, -5), (0, 0, 1)) == []
    # shape.rs:2394-2409: two overlapping spheres, union -> t = 4 (sphere1), 6.5 (sphere2)
    w = orc.World()
    c = w.csg(orc.CSG_UNION)
    s1 = w.shape(orc.SPHERE, parent=c)
    s2 = w.shape(orc.SPHERE, transform=orc.translation(0, 0, 0.5), parent=c)
    xs = w.intersect(c, (0, 0, -5), (0, 0, 1))
    assert [(x[0], x[1]) for x in xs] == [(4.0, s1), (6.5, s2)]
