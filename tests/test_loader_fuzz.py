"""CPU: seeded random YAML scenes through BOTH loaders -- the C++ one behind the C ABI (rtc_scene_load_yaml,
csrc/rtc_loader.cpp: an own parser for the YAML subset the reference's scenes use) and the Python mirror
(scene_yaml.py over PyYAML with serde_yaml's core-schema rules).  Both restate parse_scene/src/lib.rs:118-784; here they
must build the same tree -- transforms, materials, cylinder / cone limits, group order (nested groups first,
lib.rs:677-712), `define` / `extend` (Q16), first definition of a name wins (lib.rs find()), lights, camera -- bit for
bit, whatever the number spelling (integers, decimals, exponents, signs), flow-sequence spacing, comments and blank
lines the generator throws in."""
import numpy as np
import pytest

import test_cpp_loader as CL
from scene_feed import rtc, SCENES


def _num(rng, lo, hi, integers=0.3):
    v = float(rng.uniform(lo, hi))
    r = rng.random()
    if r < integers:
        return str(int(round(v)))
    if r < 0.75:
        return repr(round(v, int(rng.integers(1, 7))))
    if r < 0.85:
        return "%.*e" % (int(rng.integers(1, 6)), v)
    if r < 0.92:
        s = repr(round(v, 3))
        return "+" + s if v >= 0 else s
    return "%.3f" % v


def _vec(rng, lo, hi, n=3):
    sep = [", ", ",", " , ", ",  "][int(rng.integers(0, 4))]
    pad = ["", " "][int(rng.integers(0, 2))]
    return "[" + pad + sep.join(_num(rng, lo, hi) for _ in range(n)) + pad + "]"


def _transform_item(rng, names):
    r = rng.random()
    if names and r < 0.25:
        return names[int(rng.integers(0, len(names)))]
    sep = [", ", ","][int(rng.integers(0, 2))]
    if r < 0.5:
        return "[" + sep.join(["scale"] + [_num(rng, 0.2, 3.0) for _ in range(3)]) + "]"
    if r < 0.75:
        return "[ " + sep.join(["translate"] + [_num(rng, -5.0, 5.0) for _ in range(3)]) + " ]"
    return "[" + sep.join(["rotate-" + "xyz"[int(rng.integers(0, 3))], _num(rng, -3.2, 3.2, integers=0.1)]) + "]"


def _material_lines(rng, indent):
    keys = [("color", lambda: _vec(rng, 0.0, 1.0)), ("ambient", lambda: _num(rng, 0, 0.5)),
            ("diffuse", lambda: _num(rng, 0, 1)), ("specular", lambda: _num(rng, 0, 1)),
            ("shininess", lambda: _num(rng, 1, 400)), ("reflective", lambda: _num(rng, 0, 1)),
            ("transparency", lambda: _num(rng, 0, 1)), ("refractive-index", lambda: _num(rng, 1, 2.5)),
            ("casts-shadow", lambda: ["true", "false"][int(rng.integers(0, 2))])]
    out = []
    for k, f in keys:
        if rng.random() < 0.45:
            out.append(" " * indent + f"{k}: {f()}")
    if not out:
        out.append(" " * indent + "diffuse: " + _num(rng, 0, 1))
    return out


def _object_lines(rng, indent, mats, trs, level=0):
    pad = " " * indent
    kinds = ["sphere", "plane", "cube", "cylinder", "cone"] + (["group"] * 2 if level < 3 else [])
    kind = kinds[int(rng.integers(0, len(kinds)))]
    out = [pad + "- add: " + kind]
    body = " " * (indent + 2)
    if rng.random() < 0.2:
        out.append(body + "# " + "a comment with: colons, [brackets] and - dashes")
    if rng.random() < 0.7:
        if mats and rng.random() < 0.4:
            out.append(body + "material: " + mats[int(rng.integers(0, len(mats)))])
        else:
            out.append(body + "material:")
            out += _material_lines(rng, indent + 4)
    if rng.random() < 0.8:
        out.append(body + "transform:")
        for _ in range(int(rng.integers(1, 5))):
            out.append(body + "  - " + _transform_item(rng, trs))
    if kind in ("cylinder", "cone"):
        if rng.random() < 0.7:
            out.append(body + ["min", "minimum"][1] + ": " + _num(rng, -2, 0))
        if rng.random() < 0.7:
            out.append(body + "maximum: " + _num(rng, 0, 2))
        if rng.random() < 0.6:
            out.append(body + "closed: " + ["true", "false"][int(rng.integers(0, 2))])
    if kind == "group":
        out.append(body + "children:")
        for _ in range(int(rng.integers(1, 5))):
            out += _object_lines(rng, indent + 4, mats, trs, level + 1)
    if rng.random() < 0.15:
        out.append("")
    return out


def random_yaml(seed):
    rng = np.random.default_rng(seed)
    lines = ["# scene %d" % seed, "- add: camera", "  width: %d" % rng.integers(16, 400),
             "  height: %d" % rng.integers(16, 300), "  field-of-view: " + _num(rng, 0.3, 1.6, integers=0.0),
             "  from: " + _vec(rng, -10, 10), "  to: " + _vec(rng, -2, 2), "  up: [0, 1, 0]", ""]
    for _ in range(int(rng.integers(1, 4))):
        if rng.random() < 0.3:
            lines += ["- add: area-light", "  corner: " + _vec(rng, -5, 5), "  full_uvec: " + _vec(rng, 0, 2),
                      "  usteps: %d" % rng.integers(1, 5), "  full_vvec: " + _vec(rng, 0, 2),
                      "  vsteps: %d" % rng.integers(1, 5), "  intensity: " + _vec(rng, 0.1, 1.5)]
        else:
            lines += ["- add: light", "  at: " + _vec(rng, -20, 20), "  intensity: " + _vec(rng, 0.1, 1.0)]
    mats, trs = [], []
    for k in range(int(rng.integers(0, 6))):
        if rng.random() < 0.5:
            name = "mat-%d" % (k if rng.random() < 0.85 or not mats else 0)  # (sometimes a name that is taken: the first wins)
            lines.append("- define: " + name)
            if mats and rng.random() < 0.5:
                lines.append("  extend: " + mats[int(rng.integers(0, len(mats)))])
            lines.append("  value:")
            lines += _material_lines(rng, 4)
            if name not in mats:
                mats.append(name)
        else:
            name = "tr-%d" % k
            lines += ["- define: " + name, "  value:"]
            for _ in range(int(rng.integers(1, 4))):
                lines.append("    - " + _transform_item(rng, trs))
            trs.append(name)
        if rng.random() < 0.3:
            lines.append("")
    for _ in range(int(rng.integers(1, 9))):
        lines += _object_lines(rng, 0, mats, trs)
    if rng.random() < 0.3:
        lines += ["- add: obj", "  file: obj/triangles.obj", "  transform:", "    - " + _transform_item(rng, trs)]
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", list(range(200)))
def test_both_loaders_build_the_same_tree(seed):
    text = random_yaml(seed)
    nat = rtc.NativeScene.from_yaml(text, base_dir=SCENES)
    try:
        py = rtc.parse_scene(text, base_dir=SCENES)
        a, b = [], []
        CL._walk_native(nat, -1, a)
        CL._walk_python(py.objects, b)
        assert len(a) == len(b), text
        for x, y in zip(a, b):
            assert x == y, text
        cam = nat.camera()
        assert (cam["hsize"], cam["vsize"], cam["field_of_view"]) == (py.camera.hsize, py.camera.vsize, py.camera.field_of_view)
        assert np.array_equal(cam["view"], py.camera.transformation.m)
        nl = nat.lights()
        assert len(nl) == len(py.lights)
        for x, y in zip(nl, py.lights):
            if isinstance(y, rtc.AreaLight):
                assert x["is_area"] == 1 and x["usteps"] == y.usteps and x["vsteps"] == y.vsteps, text
                assert x["intensity"] == list(map(float, y.intensity)) and x["corner"] == list(map(float, y.corner)), text
                # AreaLight::new, light.rs:80-104: uvec = full_uvec / usteps
                assert x["uvec"] == [float(c) / y.usteps for c in y.full_uvec], text
                assert x["vvec"] == [float(c) / y.vsteps for c in y.full_vvec], text
            else:
                assert x["position"] == list(map(float, y.position)) and x["intensity"] == list(map(float, y.intensity))
    finally:
        nat.close()


def _mutate(text, seed):
    """one damaged line: deleted, its ':' / '[' / ']' / ',' removed, indented one more or two less"""
    rng = np.random.default_rng(seed + 9999)
    lines = text.split("\n")
    k = int(rng.integers(1, len(lines) - 1))
    mode = int(rng.integers(0, 7))
    if mode == 0:
        del lines[k]
    elif mode == 1:
        lines[k] = lines[k].replace(":", "", 1)
    elif mode == 2:
        lines[k] = lines[k].replace("[", "", 1)
    elif mode == 3:
        lines[k] = " " + lines[k]
    elif mode == 4:
        lines[k] = lines[k].replace("]", "", 1)
    elif mode == 5:
        lines[k] = lines[k].replace(",", " ", 1)
    elif lines[k].startswith("  "):
        lines[k] = lines[k][2:]
    return "\n".join(lines)


@pytest.mark.parametrize("block", range(10))
def test_both_loaders_agree_on_damaged_scenes(block):
    # 60 damaged scenes per block: both loaders refuse, or both accept and build the same tree (about half of each)
    accepted = refused = 0
    for seed in range(60 * block, 60 * block + 60):
        text = _mutate(random_yaml(seed % 200), seed)
        nat = py = None
        try:
            nat = rtc.NativeScene.from_yaml(text, base_dir=SCENES)
        except rtc.RtcError:
            pass
        try:
            py = rtc.parse_scene(text, base_dir=SCENES)
        except rtc.ParseError:
            pass
        try:
            assert (nat is None) == (py is None), text
            if nat is not None:
                a, b = [], []
                CL._walk_native(nat, -1, a)
                CL._walk_python(py.objects, b)
                assert a == b, text
                accepted += 1
            else:
                refused += 1
        finally:
            if nat is not None:
                nat.close()
    assert accepted > 5 and refused > 5


DUPLICATE = """
- add: camera
  width: 10
  height: 10
  field-of-view: 1
  from: [0, 0, -5]
  to: [0, 0, 0]
  up: [0, 1, 0]
- add: sphere
  material:
    color: [1, 0, 0]
  transform:
    - [scale, 2, 2, 2]
  material:
    color: [0, 1, 0]
"""


def test_a_key_named_twice_is_refused_by_both_loaders():
    # serde_yaml 0.9 (parse_scene/Cargo.toml:9 pins 0.9.21) fails on a mapping that names a key twice ("duplicate entry
    # with key"), so parse_scene returns ParseError::Yaml; found by the damaged-scene fuzz above, where the C++ parser
    # kept the first `material:` and PyYAML the last
    with pytest.raises(rtc.RtcError) as e:
        rtc.NativeScene.from_yaml(DUPLICATE)
    assert "duplicate entry with key" in str(e.value)
    with pytest.raises(rtc.ParseError) as e:
        rtc.parse_scene(DUPLICATE)
    assert "duplicate entry with key" in str(e.value)
    with pytest.raises(rtc.RtcError):
        rtc.NativeScene.from_yaml(DUPLICATE.replace("  material:\n    color: [0, 1, 0]\n", "  material: {color: [0, 1, 0], color: [1, 1, 1]}\n")
                                  .replace("  material:\n    color: [1, 0, 0]\n", ""))


# ----------------------------------------------------------------------------------------------- obj_file

_NUMS = ["1", "-1", "0.5", "-2.25", "1e2", "1.5E-1", "+3", ".5", "5.", "1_0", "0x10", "inf", "-inf", "nan", "NaN",
         "Infinity", "1e", "--1", "1.2.3", "1,5", "١", "1e+2", "-.5e1", "1d2", "0x1p3", "1f"]


def random_obj(seed):
    """vertices, normals, faces in every index form (v, v//n, v/t/n, v/t), polygons (fans, Q11), named groups, ignored
    records; every second file also holds records that must be refused (bad numbers, indices out of range, too few
    vertices, an unnamed group)"""
    rng = np.random.default_rng(seed)
    exotic = rng.random() < 0.5

    def num():
        if exotic and rng.random() < 0.08:
            return _NUMS[int(rng.integers(0, len(_NUMS)))]
        return _NUMS[int(rng.integers(0, 8))] if rng.random() < 0.5 else repr(round(float(rng.uniform(-3, 3)), 3))

    lines = []
    nv, nn = int(rng.integers(3, 12)), int(rng.integers(0, 6))
    for _ in range(nv):
        lines.append("v " + " ".join(num() for _ in range(3 if rng.random() < 0.9 else int(rng.integers(1, 5)))))
    for _ in range(nn):
        lines.append("vn " + " ".join(num() for _ in range(3)))
    for _ in range(int(rng.integers(1, 10))):
        r = rng.random()
        if r < 0.1:
            lines.append("g " + ["First", "Second", "First", ""][int(rng.integers(0, 4))])
        elif r < 0.15:
            lines.append(["# comment", "vt 0 1", "s off", "usemtl x", "", "   "][int(rng.integers(0, 6))])
        else:
            toks, mode = [], int(rng.integers(0, 4))
            for _ in range(int(rng.integers(1 if exotic else 3, 7))):
                vi = int(rng.integers(0 if exotic else 1, nv + (2 if exotic else 1)))
                ni = int(rng.integers(0 if exotic else 1, max(nn, 1) + (2 if exotic else 1)))
                if mode == 0 or (nn == 0 and not exotic):
                    toks.append(str(vi))
                elif mode == 1:
                    toks.append(f"{vi}//{ni}")
                elif mode == 2:
                    toks.append(f"{vi}/7/{ni}")
                else:
                    toks.append(f"{vi}/{ni}" if exotic else str(vi))
            if exotic and rng.random() < 0.1:
                toks[0] = ["+1", "-1", "1.0", "a", "1/", "/1", "1//"][int(rng.integers(0, 7))]
            lines.append("f " + " ".join(toks))
    return "\n".join(lines) + "\n"


def _obj_native(sc, parent, out):
    for c in sc.children(parent):
        if sc.item(c)["kind"] == CL.GROUP:
            out.append("group")
            _obj_native(sc, c, out)
            out.append("end")
        else:
            pts, nrm = sc.triangles(c)
            out.append((pts.tobytes(), None if nrm is None else nrm.tobytes()))


def _obj_python(objs, out):
    for o in objs:
        if isinstance(o, rtc.Group):
            out.append("group")
            _obj_python(o.children, out)
            out.append("end")
        else:
            pts = np.ascontiguousarray(o.points, dtype=np.float64).reshape(-1, 9)
            nrm = None if o.normals is None else np.ascontiguousarray(o.normals, dtype=np.float64).reshape(-1, 9)
            out.append((pts.tobytes(), None if nrm is None else nrm.tobytes()))


@pytest.mark.parametrize("block", range(8))
def test_both_obj_loaders_agree(block):
    accepted = refused = 0
    for seed in range(100 * block, 100 * block + 100):
        text = random_obj(seed)
        sc = po = None
        try:
            sc, _ = rtc.NativeScene.from_obj(text)
        except rtc.RtcError:
            pass
        try:
            po = rtc.parse_obj_string(text)
        except rtc.ParseError:
            pass
        try:
            assert (sc is None) == (po is None), text
            if sc is not None:
                a, b = [], []
                _obj_native(sc, -1, a)
                _obj_python([po.default_group], b)
                assert a == b, text
                accepted += 1
            else:
                refused += 1
        finally:
            if sc is not None:
                sc.close()
    assert accepted > 10 and refused > 10


@pytest.mark.parametrize("tok,ok", [("1", True), ("+3", True), (".5", True), ("5.", True), ("1.5E-1", True), ("-.5e1", True),
                                    ("inf", True), ("-Infinity", True), ("NaN", True), ("1_0", False), ("0x10", False),
                                    ("0x1p3", False), ("١", False), ("1e", False), ("1d2", False), ("1,5", False),
                                    ("nan(1)", False), ("infinit", False), (".", False), ("+", False)])
def test_vertex_numbers_follow_rust_f64_from_str(tok, ok):
    # obj_file/src/lib.rs:82-96 parses coordinates with str::parse::<f64>(): ASCII decimal syntax, or inf / infinity /
    # nan in any case.  (strtod would also take hexadecimal floats, Python's float() "1_0" and non-ASCII digits: found by
    # the fuzz above.)
    text = f"v {tok} 0 0\n"
    for load, err in ((lambda: rtc.NativeScene.from_obj(text)[0].close(), rtc.RtcError), (lambda: rtc.parse_obj_string(text), rtc.ParseError)):
        if ok:
            load()
        else:
            with pytest.raises(err):
                load()


# ----------------------------------------------------------------------------------------------- scalar typing

_CAM = """
- add: camera
  width: 10
  height: 10
  field-of-view: 1
  from: [0, 0, -5]
  to: [0, 0, 0]
  up: [0, 1, 0]
"""

# spelling -> the number serde_yaml 0.9 reads (None: it reads a STRING, and parse_scene's get_f64 refuses the scene)
_SCALARS = [("3", 3.0), ("-3", -3.0), ("+3", 3.0), ("0", 0.0), ("-0", 0.0), ("0x10", 16.0), ("-0x10", -16.0), ("+0x1F", 31.0),
            ("0o17", 15.0), ("0b101", 5.0), ("007", None), ("-012", None), ("00", None), ("0.5", 0.5), ("00.5", 0.5),
            (".5", 0.5), ("5.", 5.0), ("+.5", 0.5), ("-.5e1", -5.0), ("1e3", 1000.0), ("1E+3", 1000.0), ("1e-2", 0.01),
            ("1_000", None), ("0x1G", None), ("0b102", None), ("++1", None), ("+-1", None), ("1e", None), ("1e999", None),
            ("inf", None), ("nan", None), ("Infinity", None), (".inf", float("inf")), ("+.inf", float("inf")),
            ("-.INF", float("-inf")), ("18446744073709551615", 18446744073709551615.0),
            ("99999999999999999999", 1e20), ("-9223372036854775808", -9223372036854775808.0), ("true", None), ("~", None),
            ("'3'", None), ('"3"', None), ("three", None)]


@pytest.mark.parametrize("spelling,value", _SCALARS)
def test_plain_scalars_are_typed_like_serde_yaml_0_9(spelling, value):
    # serde_yaml 0.9 de.rs (visit_untagged_scalar, parse_unsigned_int / parse_negative_int, digits_but_not_number,
    # parse_f64), restated from its published source -- the crate is not vendored (parse_scene/Cargo.toml:9 pins
    # 0.9.21): parity unpinned.  The number arrives as the x of a translation.
    text = _CAM + f"- add: sphere\n  transform:\n    - [translate, {spelling}, 0, 0]\n"
    if value is None:
        with pytest.raises(rtc.RtcError):
            rtc.NativeScene.from_yaml(text)
        with pytest.raises(rtc.ParseError):
            rtc.parse_scene(text)
        return
    nat = rtc.NativeScene.from_yaml(text)
    try:
        (obj,) = nat.children(-1)
        a = nat.item(obj)["transform"]
    finally:
        nat.close()
    b = np.asarray(rtc.parse_scene(text).objects[0].transformation.m)
    assert a[0, 3] == value and b[0, 3] == value and np.array_equal(a, b, equal_nan=True)  # (inf * 0 in the product)
