"""Host side of the boundary without a GPU: flattening, BVH, tiling, error behaviour (rtc_scene_commit with
device = -1 does the host work only)."""
import ctypes as C
import math

import numpy as np
import pytest

import scene_feed as sf
from scene_feed import rtc

N = rtc._native


def _info(name, w=64, h=32):
    world, cam = sf.load(name, w, h)
    sc = rtc.CompiledScene(world, cam, device=-1)
    try:
        return sc.info()
    finally:
        sc.close()


def test_sample_scenes_flatten():
    # cover = 19 objects / 2 lights, reflective_world = 6 / 1, ice_cream = 2 / 1
    i = _info("cover")
    assert (i["n_prims"], i["n_lights"], i["n_unbounded"]) == (19, 2, 1)
    i = _info("reflective_world")
    assert (i["n_prims"], i["n_lights"]) == (6, 1)
    assert i["n_unbounded"] >= 2  # the two planes (+ unbounded cylinders / open cones)
    i = _info("ice_cream")
    assert (i["n_prims"], i["n_lights"]) == (2, 1)


def test_teapot_flattens_to_a_mesh_with_a_bvh():
    i = _info("teapot")
    # obj/teapot.obj: 3120 quads + 80 triangles, fan-triangulated (obj_file/src/lib.rs:150-211); 2 planes
    assert i["n_triangles"] == 3120 * 2 + 80
    assert i["n_prims"] == i["n_triangles"] + 2 and i["n_unbounded"] == 2
    assert i["n_bvh_nodes"] >= i["n_triangles"] // 4 - 1
    assert i["bvh_max_leaf"] <= 4 and i["bvh_max_depth"] <= 30
    assert i["n_lights"] == 1 and i["committed"] == 1


def test_render_without_device_fails_loudly():
    world, cam = sf.load("ice_cream", 16, 8)
    sc = rtc.CompiledScene(world, cam, device=-1)
    try:
        with pytest.raises(rtc.RtcError) as e:
            sc.render()
        assert e.value.code == N.ERR_NO_DEVICE
    finally:
        sc.close()


def test_non_invertible_transform_is_reported_at_commit():
    # error.rs:3-7 NonInvertibleMatrix (matrix.rs:153-161): det == 0
    world = rtc.World([rtc.Sphere(transformation=rtc.Matrix.scaling(1.0, 0.0, 1.0))],
                      [rtc.PointLight((-10, 10, -10), (1, 1, 1))])
    cam = rtc.Camera(8, 8, math.pi / 2)
    with pytest.raises(rtc.NonInvertibleMatrix):
        rtc.CompiledScene(world, cam, device=-1)
    with pytest.raises(rtc.NonInvertibleMatrix):
        rtc.Matrix([[1, 2, 3, 4], [2, 4, 6, 8], [0, 1, 0, 1], [1, 0, 0, 0]]).inverse()


def test_call_order_errors():
    L = N.lib()
    h = L.rtc_scene_create()
    try:
        assert L.rtc_scene_end_group(h) == N.ERR_STATE
        assert L.rtc_scene_commit(h, -1) == N.ERR_INVALID  # no camera
        assert b"camera" in L.rtc_last_error()
        assert L.rtc_scene_add_shape(h, 99, None, 0, 0.0, 0.0, 0) == N.ERR_INVALID
    finally:
        L.rtc_scene_destroy(h)
    assert L.rtc_scene_commit(None, -1) == N.ERR_INVALID


@pytest.mark.parametrize("vsize,tile_rows,count", [(2160, 8, 8), (1080, 8, 3), (50, 4, 3), (7, 16, 2), (33, 1, 5)])
def test_row_tiles_partition_the_canvas(vsize, tile_rows, count):
    L = N.lib()
    seen = []
    for r in range(count):
        n = L.rtc_local_rows(vsize, tile_rows, r, count)
        rows = rtc.parallel.tile_row_map(vsize, tile_rows, r, count)
        assert n == len(rows)
        buf = (N.C.c_uint32 * max(n, 1))()
        assert L.rtc_local_row_map(vsize, tile_rows, r, count, buf, n) == n
        assert list(buf[:n]) == rows.tolist()
        seen.extend(rows.tolist())
    assert sorted(seen) == list(range(vsize))


def test_gather_index_round_trip():
    vsize, tile_rows, world_size = 37, 4, 3
    padded = rtc.parallel.max_local_rows(vsize, tile_rows, world_size)
    idx = rtc.parallel.gather_index(vsize, tile_rows, world_size, padded)
    canvas = np.zeros(vsize + 1)
    for r in range(world_size):
        rows = rtc.parallel.tile_row_map(vsize, tile_rows, r, world_size)
        local = np.full(padded, -1.0)
        local[: len(rows)] = rows
        canvas[idx[r * padded:(r + 1) * padded]] = local
    assert canvas[:vsize].tolist() == list(map(float, range(vsize)))


def test_host_matrix_known_answers():
    # matrix.rs:605-680 and transformation.rs:60-77 through the C ABI (the oracle has its own copy of these tests)
    a = rtc.Matrix([[-5, 2, 6, -8], [1, -5, 1, 8], [7, 7, -6, -7], [1, -3, 7, 4]])
    exp = [[0.21805, 0.45113, 0.24060, -0.04511], [-0.80827, -1.45677, -0.44361, 0.52068],
           [-0.07895, -0.22368, -0.05263, 0.19737], [-0.52256, -0.81391, -0.30075, 0.30639]]
    assert np.all(np.abs(a.inverse().m - np.array(exp)) < 1e-5)
    b = rtc.Matrix([[8, 2, 2, 2], [3, -1, 7, 0], [7, 0, 5, 4], [6, -2, 0, 5]])
    assert np.all(np.abs(((a @ b) @ b.inverse()).m - a.m) < 1e-5)
    v = rtc.compute_view_transformation((1, 3, 2), (4, -2, 8), (1, 1, 0))
    expv = [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
            [-0.35857, 0.59761, -0.71714, 0.0], [0.0, 0.0, 0.0, 1.0]]
    assert np.all(np.abs(v.m - np.array(expv)) < 1e-5)
    # fluent builders left-multiply in call order (matrix.rs:238-266)
    t = rtc.Matrix.identity().rotate_x(math.pi / 2).scale(5, 5, 5).translate(10, 5, 7)
    p = t.m @ np.array([1.0, 0.0, 1.0, 1.0])
    assert np.all(np.abs(p[:3] - np.array([15.0, 0.0, 7.0])) < 1e-5)


def test_csg_call_order_and_flattening():
    # Csg::new takes exactly a left and a right shape (shape.rs:2166); the flattener turns the tree into one entry
    # of the intersection structures
    N = rtc._native
    L = N.lib()
    h = C.c_void_p(L.rtc_scene_create())
    m = N.Material()
    L.rtc_material_default(C.byref(m))
    mid = C.c_uint32()
    assert L.rtc_scene_add_material(h, C.byref(m), C.byref(mid)) == N.OK
    assert L.rtc_scene_begin_csg(h, 7) == N.ERR_INVALID
    assert L.rtc_scene_begin_csg(h, N.CSG_DIFFERENCE) == N.OK
    assert L.rtc_scene_end_csg(h) == N.ERR_STATE                      # no operands yet
    assert L.rtc_scene_add_shape(h, N.CUBE, None, mid.value, 0.0, 0.0, 0) == N.OK
    pts = (C.c_double * 9)(0, 0, 0, 1, 0, 0, 0, 1, 0)
    assert L.rtc_scene_add_triangles(h, 1, pts, None, mid.value) == N.ERR_INVALID   # shape.rs:1521-1523
    assert L.rtc_scene_end_group(h) == N.ERR_STATE                    # the open node is a Csg
    assert L.rtc_scene_begin_csg(h, N.CSG_UNION) == N.OK              # right operand: another Csg
    assert L.rtc_scene_add_shape(h, N.SPHERE, None, mid.value, 0.0, 0.0, 0) == N.OK
    assert L.rtc_scene_add_shape(h, N.SPHERE, None, mid.value, 0.0, 0.0, 0) == N.OK
    assert L.rtc_scene_add_shape(h, N.SPHERE, None, mid.value, 0.0, 0.0, 0) == N.ERR_STATE  # a third operand
    assert L.rtc_scene_end_csg(h) == N.OK
    assert L.rtc_scene_add_shape(h, N.SPHERE, None, mid.value, 0.0, 0.0, 0) == N.ERR_STATE
    assert L.rtc_scene_end_csg(h) == N.OK
    assert L.rtc_scene_set_camera(h, 16, 16, 1.0, None) == N.OK
    assert L.rtc_scene_commit(h, -1) == N.OK
    info = N.SceneInfo()
    assert L.rtc_scene_get_info(h, C.byref(info)) == N.OK
    assert info.n_prims == 3 + 1          # three leaves + the tree's pseudo-primitive
    it = N.ItemInfo()
    assert L.rtc_scene_get_item(h, 0, C.byref(it)) == N.OK and it.kind == 3 and it.shape_type == N.CSG_DIFFERENCE
    L.rtc_scene_destroy(h)


def test_heavy_tiles_are_the_ones_that_show_reflective_or_transparent_objects():
    # the static schedule's tile order: tiles touched by the projected bounds of something
    # reflective or transparent are "heavy" -- rendered first and, where the device has room, in parts
    tiles = lambda w, h: ((w + 7) // 8) * ((h + 7) // 8)
    i = _info("cover", 320, 180)          # a glass sphere among 17 slightly reflective cubes
    assert 0 < i["n_heavy_tiles"] < tiles(320, 180)
    j = _info("cover", 640, 360)
    assert 3 * i["n_heavy_tiles"] < j["n_heavy_tiles"] < 5 * i["n_heavy_tiles"]  # an area: ~4x at twice the resolution
    assert _info("ice_cream", 320, 180)["n_heavy_tiles"] == 0   # opaque cone + sphere
    assert _info("teapot", 320, 160)["n_heavy_tiles"] == 0      # opaque mesh, matte planes (unbounded: never weighed)
