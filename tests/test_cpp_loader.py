"""The C++ loaders behind the C ABI (rtc_scene_load_yaml, rtc_scene_add_obj, rtc_canvas_to_ppm).

1. the reference's own loader tests (parse_scene/src/lib.rs:790-1047, obj_file/src/lib.rs:248-448,
   canvas.rs:115-179) re-stated against the builder tree the C++ loader produces;
2. the C++ loader and the Python mirror (scene_yaml.py, itself checked in test_loaders.py) must build the SAME tree
   -- transforms, materials and triangle coordinates compared bit for bit -- for every sample scene.
"""
import os

import numpy as np
import pytest

import test_loaders as TL
from scene_feed import rtc, SCENES

N = rtc._native
I = rtc.Matrix.identity
GROUP, SHAPE, TRIS = 0, 1, 2


def _mat_eq(m, **kw):
    d = rtc.Material(**kw)
    for f in ("ambient", "diffuse", "specular", "shininess", "reflectivity", "transparency", "refractive_index"):
        assert getattr(m, f) == getattr(d, f), f
    assert tuple(m.color) == tuple(map(float, d.color))


def test_parse_simple_scene():  # parse_scene/src/lib.rs:790-839
    sc = rtc.NativeScene.from_yaml(TL.CAMERA + """
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
    lights = sc.lights()
    assert len(lights) == 2
    assert lights[0]["position"] == [-10, 10, -10] and lights[0]["intensity"] == [1, 1, 1]
    assert lights[1]["position"] == [10, 10, -10]
    objs = sc.children(-1)
    assert len(objs) == 2
    s = sc.item(objs[0])
    assert (s["kind"], s["shape_type"]) == (SHAPE, N.SPHERE)
    _mat_eq(sc.material(s["material_id"]), color=(1, 0.2, 1), diffuse=0.7, specular=0.3)
    assert np.array_equal(s["transform"], I().scale(1, 0.5, 1).translate(0, 1, 0).m)
    cam = sc.camera()
    assert (cam["hsize"], cam["vsize"], cam["field_of_view"]) == (1000, 500, 1.0471975512)
    assert np.array_equal(cam["view"], rtc.compute_view_transformation((0, 5, -7), (0, 2, 0), (0, 1, 0)).m)
    sc.close()


def test_parse_scene_with_definitions():  # parse_scene/src/lib.rs:841-901
    sc = rtc.NativeScene.from_yaml(TL.CAMERA + """
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
    (o,) = sc.children(-1)
    it = sc.item(o)
    assert it["shape_type"] == N.CUBE
    _mat_eq(sc.material(it["material_id"]), color=(0.537, 0.831, 0.914), diffuse=0.7, ambient=0.1, specular=0.0,
            shininess=250.0, reflectivity=0.1)
    exp = I().translate(1, -1, 1).scale(0.5, 0.5, 0.5).scale(3.5, 3.5, 3.5).translate(0.5, 1.5, -0.5)
    assert np.array_equal(it["transform"], exp.m)
    sc.close()


def test_parse_scene_with_nested_groups():  # parse_scene/src/lib.rs:903-1047
    sc = rtc.NativeScene.from_yaml(TL.GROUP + """
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
    (g,) = sc.children(-1)
    gi = sc.item(g)
    assert gi["kind"] == GROUP and np.array_equal(gi["transform"], I().translate(0, 1, 0).m)
    child_group, plane = sc.children(g)
    assert sc.item(child_group)["kind"] == GROUP
    assert np.array_equal(sc.item(child_group)["transform"], rtc.Matrix.scaling(1, 0.5, 1).m)
    sphere, cube = sc.children(child_group)
    _mat_eq(sc.material(sc.item(sphere)["material_id"]), color=(1, 0.2, 1), diffuse=0.7, specular=0.3)
    _mat_eq(sc.material(sc.item(cube)["material_id"]))
    pi = sc.item(plane)
    assert pi["shape_type"] == N.PLANE
    _mat_eq(sc.material(pi["material_id"]), color=(1, 0.9, 0.9), specular=0.0)
    assert np.array_equal(pi["transform"], I().rotate_x(1.5708).translate(0, 0, 5).m)
    sc.close()


@pytest.mark.parametrize("text", [
    "- add: light\n  at: [0, 0, 0]\n  intensity: [1, 1, 1]\n",          # MissingCamera
    TL.CAMERA + "- add: sphere\n  material: nope\n",                       # unknown definition
    TL.CAMERA + "- add: torus\n",                                          # unknown item
    TL.CAMERA + "- add: sphere\n  transform:\n    - [ wobble, 1 ]\n",      # unknown transformation
    TL.CAMERA + "- add: group\n",                                          # missing children
    "add: camera\n",                                                        # not a sequence
])
def test_parse_errors(text):
    with pytest.raises(rtc.RtcError) as e:
        rtc.NativeScene.from_yaml(text)
    assert e.value.code == N.ERR_PARSE


# ---------------------------------------------------------------------------------------------- obj_file

def _all_tris(sc, group):
    out = []
    for c in sc.children(group):
        it = sc.item(c)
        if it["kind"] == TRIS:
            pts, nrm = sc.triangles(c)
            for k in range(pts.shape[0]):
                out.append((pts[k].reshape(3, 3), None if nrm is None else nrm[k].reshape(3, 3)))
    return out


def test_obj_gibberish_and_vertices():  # obj_file/src/lib.rs:248-280
    text = ("There was a young lady named Bright\nWho traveled much faster than light.\nShe set out one day\n"
            "In a relative way,\nAnd came back the previous night.")
    sc, info = rtc.NativeScene.from_obj(text)
    assert info["ignored_lines"] == 5
    sc.close()
    sc, info = rtc.NativeScene.from_obj("\n#\n# some random gibberish here\n#\n\nv  -1 1 0\nv -1.0000 0.5000 0.0000\nv 1 0 0\nv 1 1 0\n")
    assert info["vertices"] == 4 and info["ignored_lines"] == 3
    sc.close()


def test_obj_faces_and_fans():  # :282-347
    sc, info = rtc.NativeScene.from_obj("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\n\nf 1 2 3\nf 1 3 4\nf 1 2 3 4 5\n")
    (g,) = sc.children(-1)
    t = _all_tris(sc, g)
    v = np.array([[-1, 1, 0], [-1, 0, 0], [1, 0, 0], [1, 1, 0], [0, 2, 0]], dtype=float)
    assert info["triangles"] == 5 and len(t) == 5
    for tri, idx in zip(t, ([0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 3], [0, 3, 4])):
        assert np.array_equal(tri[0], v[idx]) and tri[1] is None
    sc.close()


def test_obj_named_groups():  # :349-388 (fixture: obj_file/triangles.obj)
    text = open(os.path.join(SCENES, "obj", "triangles.obj")).read()
    sc, info = rtc.NativeScene.from_obj(text)
    (g,) = sc.children(-1)
    kids = sc.children(g)
    assert info["groups"] == 2 and [sc.item(k)["kind"] for k in kids] == [GROUP, GROUP]
    v = np.array([[-1, 1, 0], [-1, 0, 0], [1, 0, 0], [1, 1, 0]], dtype=float)
    assert np.array_equal(_all_tris(sc, kids[0])[0][0], v[[0, 1, 2]])
    assert np.array_equal(_all_tris(sc, kids[1])[0][0], v[[0, 2, 3]])
    sc.close()


def test_obj_normals_and_fan_quirk():  # :390-448 and Q11 (obj_file/src/lib.rs:160-171)
    sc, info = rtc.NativeScene.from_obj("v 0 1 0\nv -1 0 0\nv 1 0 0\n\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\n\n"
                                        "f 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2\n")
    assert info["normals"] == 3
    (g,) = sc.children(-1)
    v = np.array([[0, 1, 0], [-1, 0, 0], [1, 0, 0]], dtype=float)
    n = np.array([[-1, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=float)
    for pts, nrm in _all_tris(sc, g):
        assert np.array_equal(pts, v) and np.array_equal(nrm, n[[2, 0, 1]])
    sc.close()
    sc, _ = rtc.NativeScene.from_obj("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 1 0 0\nvn 0 1 0\nvn 0 0 1\nvn 1 1 1\n"
                                     "f 1//1 2//2 3//3 4//4\n")
    (g,) = sc.children(-1)
    t = _all_tris(sc, g)
    nn = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    assert len(t) == 2 and np.array_equal(t[0][1], nn) and np.array_equal(t[1][1], nn)
    sc.close()


def test_obj_errors():
    for text in ("v 1 2 x\n", "v 0 0 0\nf 1 2 3\n", "v 0 0 0\nf a b c\n", "g\n"):
        with pytest.raises(rtc.RtcError) as e:
            rtc.NativeScene.from_obj(text)
        assert e.value.code == N.ERR_PARSE


# ---------------------------------------------------------------------------------------------- canvas.rs

def test_ppm_header_and_pixels():  # canvas.rs:115-146
    px = np.zeros((3, 5, 3))
    px[0, 0] = (1.5, 0, 0)
    px[1, 2] = (0, 0.5, 0)
    px[2, 4] = (-0.5, 0, 1)
    lines = rtc.canvas_to_ppm(px).split("\n")
    assert lines[:3] == ["P3", "5 3", "255"]
    assert lines[3] == "255 0 0 0 0 0 0 0 0 0 0 0 0 0 0"
    assert lines[4] == "0 0 0 0 0 0 0 128 0 0 0 0 0 0 0"
    assert lines[5] == "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"


def test_ppm_line_wrapping_and_trailing_newline():  # canvas.rs:148-179
    px = np.empty((2, 10, 3))
    px[:] = (1, 0.8, 0.6)
    ppm = rtc.Canvas(px).to_ppm()
    lines = ppm.split("\n")
    assert lines[3] == "255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204"
    assert lines[4] == "153 255 204 153 255 204 153 255 204 153 255 204 153"
    assert lines[5] == lines[3] and lines[6] == lines[4]
    assert ppm.endswith("\n") and not ppm.endswith("\n\n")


def test_ppm_matches_oracle():
    import orc
    rng = np.random.default_rng(7)
    px = rng.uniform(-0.2, 1.3, size=(7, 33, 3))
    px[3, 3, 1] = np.nan
    assert rtc.canvas_to_ppm(px) == orc.canvas_to_ppm(px)


# ---------------------------------------------------------------------------------------------- C++ == Python mirror

def _walk_native(sc, parent, out):
    for c in sc.children(parent):
        it = sc.item(c)
        if it["kind"] == GROUP:
            out.append(("group", it["transform"].tobytes()))
            _walk_native(sc, c, out)
            out.append(("end",))
        elif it["kind"] == SHAPE:
            m = sc.material(it["material_id"])
            out.append(("shape", it["shape_type"], it["transform"].tobytes(), it["minimum"], it["maximum"], it["closed"],
                        tuple(m.color), m.ambient, m.diffuse, m.specular, m.shininess, m.reflectivity, m.transparency,
                        m.refractive_index, m.casts_shadow))
        else:
            pts, nrm = sc.triangles(c)
            out.append(("tris", pts.tobytes(), None if nrm is None else nrm.tobytes()))


def _walk_python(objs, out):
    for o in objs:
        if isinstance(o, rtc.Group):
            out.append(("group", np.asarray(o.transformation.m, dtype=np.float64).tobytes()))
            _walk_python(o.children, out)
            out.append(("end",))
        elif isinstance(o, rtc.Shape):
            m = o.material
            out.append(("shape", o.kind, np.asarray(o.transformation.m, dtype=np.float64).tobytes(), o.minimum, o.maximum,
                        1 if o.closed else 0, tuple(map(float, m.color)), m.ambient, m.diffuse, m.specular, m.shininess,
                        m.reflectivity, m.transparency, m.refractive_index, bool(m.casts_shadow)))
        elif isinstance(o, rtc.TriangleMesh):
            pts = np.ascontiguousarray(o.points, dtype=np.float64).reshape(-1, 9)
            nrm = None if o.normals is None else np.ascontiguousarray(o.normals, dtype=np.float64).reshape(-1, 9)
            out.append(("tris", pts.tobytes(), None if nrm is None else nrm.tobytes()))
        else:
            raise TypeError(type(o))


@pytest.mark.parametrize("name", ["cover", "reflective_world", "ice_cream", "teapot"])
def test_cpp_loader_builds_the_same_tree_as_the_python_mirror(name):
    path = os.path.join(SCENES, name + ".yaml")
    nat = rtc.NativeScene.from_yaml_file(path, 320, 200)
    py = rtc.parse_scene_file(path, 320, 200)
    a, b = [], []
    _walk_native(nat, -1, a)
    _walk_python(py.objects, b)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x == y
    cam = nat.camera()
    assert (cam["hsize"], cam["vsize"], cam["field_of_view"]) == (320, 200, py.camera.field_of_view)
    assert np.array_equal(cam["view"], py.camera.transformation.m)
    nl = nat.lights()
    assert len(nl) == len(py.lights)
    for x, y in zip(nl, py.lights):
        assert x["position"] == list(map(float, y.position)) and x["intensity"] == list(map(float, y.intensity))
    # and the flattened tables agree too
    nat.commit(-1)
    ref = rtc.CompiledScene(rtc.World(py.objects, py.lights), py.camera, device=-1)
    ia, ib = nat.info(), ref.info()
    for k in ("n_prims", "n_unbounded", "n_bvh_nodes", "n_xforms", "n_lights", "n_triangles", "bvh_max_depth"):
        assert ia[k] == ib[k], k
    nat.close(), ref.close()
