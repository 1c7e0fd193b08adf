"""Scene loaders (YAML scene files, Wavefront OBJ): the reference's own loader tests re-stated
(parse_scene/src/lib.rs:790-1047, obj_file/src/lib.rs:248-448)."""
import os

import numpy as np
import pytest

from scene_feed import rtc, SCENES

CAMERA = """
- add: camera
  width: 1000
  height: 500
  field-of-view: 1.0471975512
  from: [0, 5, -7]
  to: [0, 2, 0]
  up: [0, 1, 0]
- add: light
  at: [-10, 10, -10]
  intensity: [1, 1, 1]
"""
I = rtc.Matrix.identity


def _mat(m, **kw):
    d = rtc.Material()
    for k, v in kw.items():
        setattr(d, k, v)
    for f in ("ambient", "diffuse", "specular", "shininess", "reflectivity", "transparency", "refractive_index"):
        assert getattr(m, f) == getattr(d, f), f
    assert tuple(m.color) == tuple(d.color)


def test_parse_simple_scene():  # parse_scene/src/lib.rs:790-839
    sc = rtc.parse_scene(CAMERA + """
- add: light
  at: [10, 10, -10]
  intensity: [1, 1, 1]
- add: sphere
  material:
    color: [1, 0.2, 1]
    diffuse: 0.7
    specular: 0.3
  transform:
    - [ scale, 1, 0.5, 1 ]
    - [translate, 0, 1, 0]
- add: plane
  material:
    color: [1, 0.9, 0.9]
    specular: 0
  transform:
    - [rotate-x, 1.5708]
    - [ translate, 0, 0, 5]
""")
    assert len(sc.lights) == 2
    assert tuple(sc.lights[0].position) == (-10, 10, -10) and tuple(sc.lights[0].intensity) == (1, 1, 1)
    assert tuple(sc.lights[1].position) == (10, 10, -10)
    assert len(sc.objects) == 2
    _mat(sc.objects[0].material, color=(1, 0.2, 1), diffuse=0.7, specular=0.3)
    assert sc.objects[0].transformation == I().scale(1, 0.5, 1).translate(0, 1, 0)
    assert (sc.camera.hsize, sc.camera.vsize, sc.camera.field_of_view) == (1000, 500, 1.0471975512)


def test_parse_scene_with_definitions():  # parse_scene/src/lib.rs:841-901
    sc = rtc.parse_scene(CAMERA + """
- define: white-material
  value:
    color: [1, 1, 1]
    diffuse: 0.7
    ambient: 0.1
    shininess: 250
    specular: 0.0
    reflective: 0.1
- define: blue-material
  extend: white-material
  value:
    color: [0.537, 0.831, 0.914]
- define: standard-transform
  value:
    - [ translate, 1, -1, 1 ]
    - [ scale, 0.5, 0.5, 0.5 ]
- define: large-object
  value:
    - standard-transform
    - [ scale, 3.5, 3.5, 3.5 ]
- add: cube
  material: blue-material
  transform:
    - large-object
    - [ translate, 0.5, 1.5, -0.5 ]
""")
    assert len(sc.objects) == 1
    _mat(sc.objects[0].material, color=(0.537, 0.831, 0.914), diffuse=0.7, ambient=0.1, specular=0.0, shininess=250.0,
         reflectivity=0.1)
    assert sc.objects[0].transformation == (
        I().translate(1, -1, 1).scale(0.5, 0.5, 0.5).scale(3.5, 3.5, 3.5).translate(0.5, 1.5, -0.5))


GROUP = CAMERA + """
- add: group
  transform:
    - [ translate, 0, 1, 0 ]
  children:
"""


def test_parse_scene_with_group():  # parse_scene/src/lib.rs:903-970
    sc = rtc.parse_scene(GROUP + """
    - add: sphere
      material:
        color: [1, 0.2, 1]
        diffuse: 0.7
        specular: 0.3
      transform:
        - [ scale, 1, 0.5, 1 ]
    - add: plane
      material:
        color: [1, 0.9, 0.9]
        specular: 0
      transform:
        - [rotate-x, 1.5708]
        - [ translate, 0, 0, 5]
""")
    assert len(sc.objects) == 1
    g = sc.objects[0]
    assert isinstance(g, rtc.Group) and g.transformation == I().translate(0, 1, 0)
    sphere, plane = g.children
    _mat(sphere.material, color=(1, 0.2, 1), diffuse=0.7, specular=0.3)
    assert sphere.transformation == rtc.Matrix.scaling(1, 0.5, 1)
    _mat(plane.material, color=(1, 0.9, 0.9), specular=0.0)
    assert plane.transformation == I().rotate_x(1.5708).translate(0, 0, 5)


def test_parse_scene_with_nested_groups():  # parse_scene/src/lib.rs:972-1047
    sc = rtc.parse_scene(GROUP + """
    - add: group
      transform:
        - [ scale, 1, 0.5, 1 ]
      children:
        - add: sphere
          material:
            color: [1, 0.2, 1]
            diffuse: 0.7
            specular: 0.3
        - add: cube
          material: {}
    - add: plane
      material:
        color: [1, 0.9, 0.9]
        specular: 0
      transform:
        - [rotate-x, 1.5708]
        - [ translate, 0, 0, 5]
""")
    g = sc.objects[0]
    assert g.transformation == I().translate(0, 1, 0)
    child_group, plane = g.children
    sphere, cube = child_group.children
    _mat(sphere.material, color=(1, 0.2, 1), diffuse=0.7, specular=0.3)
    _mat(cube.material)
    _mat(plane.material, color=(1, 0.9, 0.9), specular=0.0)
    assert plane.transformation == I().rotate_x(1.5708).translate(0, 0, 5)


def test_parse_errors():
    with pytest.raises(rtc.ParseError):
        rtc.parse_scene("- add: light\n  at: [0, 0, 0]\n  intensity: [1, 1, 1]\n")  # MissingCamera
    with pytest.raises(rtc.ParseError):
        rtc.parse_scene(CAMERA + "- add: sphere\n  material: nope\n")  # undefined definition
    with pytest.raises(rtc.ParseError):
        rtc.parse_scene(CAMERA + "- add: torus\n")


# ---------------------------------------------------------------------------------------------- obj_file

def test_ignore_unrecognized_statements():  # obj_file/src/lib.rs:248-258
    text = ("There was a young lady named Bright\nWho traveled much faster than light.\nShe set out one day\n"
            "In a relative way,\nAnd came back the previous night.")
    assert rtc.parse_obj_string(text).ignored_lines == 5


def test_process_vertex_data():  # :260-280
    p = rtc.parse_obj_string("\n#\n# some random gibberish here\n#\n\nv  -1 1 0\nv -1.0000 0.5000 0.0000\nv 1 0 0\nv 1 1 0\n")
    assert p.vertices == [(-1, 1, 0), (-1, 0.5, 0), (1, 0, 0), (1, 1, 0)]


def _tris(group):
    out = []
    for c in group.children:
        if isinstance(c, rtc.TriangleMesh):
            pts = np.asarray(c.points).reshape(-1, 3, 3)
            nrm = None if c.normals is None else np.asarray(c.normals).reshape(-1, 3, 3)
            for i in range(pts.shape[0]):
                out.append((pts[i], None if nrm is None else nrm[i]))
    return out


def test_parse_triangle_faces():  # :282-310
    p = rtc.parse_obj_string("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n\nf 1 2 3\nf 1 3 4\n")
    t = _tris(p.default_group)
    v = np.array(p.vertices)
    assert np.array_equal(t[0][0], v[[0, 1, 2]]) and np.array_equal(t[1][0], v[[0, 2, 3]])


def test_triangulate_polygons():  # :312-347
    p = rtc.parse_obj_string("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\n\nf 1 2 3 4 5\n")
    t = _tris(p.default_group)
    v = np.array(p.vertices)
    assert len(t) == 3
    assert np.array_equal(t[0][0], v[[0, 1, 2]]) and np.array_equal(t[1][0], v[[0, 2, 3]])
    assert np.array_equal(t[2][0], v[[0, 3, 4]])


def test_triangles_in_groups_and_file_to_group():  # :349-388 (fixture: obj_file/triangles.obj)
    p = rtc.parse_obj_file(os.path.join(SCENES, "obj", "triangles.obj"))
    g1, g2 = p.groups["FirstGroup"], p.groups["SecondGroup"]
    v = np.array(p.vertices)
    assert np.array_equal(_tris(g1)[0][0], v[[0, 1, 2]]) and np.array_equal(_tris(g2)[0][0], v[[0, 2, 3]])
    named = [c for c in p.default_group.children if isinstance(c, rtc.Group)]
    assert named[0] is g1 and named[1] is g2


def test_vertex_normal_records():  # :390-401
    p = rtc.parse_obj_string("vn 0 0 1\nvn 0.707 0 -0.707\nvn 1 2 3\n")
    assert p.normals == [(0, 0, 1), (0.707, 0, -0.707), (1, 2, 3)]


def test_faces_with_normals():  # :403-448
    p = rtc.parse_obj_string("v 0 1 0\nv -1 0 0\nv 1 0 0\n\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\n\n"
                             "f 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2\n")
    t = _tris(p.default_group)
    v, n = np.array(p.vertices), np.array(p.normals)
    for pts, nrm in t:
        assert np.array_equal(pts, v[[0, 1, 2]]) and np.array_equal(nrm, n[[2, 0, 1]])


def test_polygon_fan_keeps_the_first_three_normals():
    # Q11 (obj_file/src/lib.rs:160-171): triangle k of a fan uses vertices (0, k, k+1) but ALWAYS normals (0, 1, 2)
    p = rtc.parse_obj_string("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 1 0 0\nvn 0 1 0\nvn 0 0 1\nvn 1 1 1\n"
                             "f 1//1 2//2 3//3 4//4\n")
    t = _tris(p.default_group)
    n = np.array(p.normals)
    assert len(t) == 2
    assert np.array_equal(t[0][1], n[[0, 1, 2]]) and np.array_equal(t[1][1], n[[0, 1, 2]])
