"""YAML scene and Wavefront OBJ loaders -- host-side mirror of the reference's `parse_scene` and `obj_file`
crates (parse_scene/src/lib.rs:735-784, obj_file/src/lib.rs:40-242).

These are loaders only: they build the description objects of api.py; pixels are computed by the HIP
kernels behind `Camera.render`.  Quirks reproduced on purpose: Q11 (OBJ polygon fans reuse normals 0,1,2),
Q14 (a `material:` on a YAML group is parsed, then dropped), Q16 (`extend` overrides only present keys;
transform lists left-multiply in list order).
"""
import os
import re
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import yaml

from .api import (AreaLight, Camera, Cone, Cube, Cylinder, Group, Material, Matrix, Plane, PointLight, Sphere,
                  TriangleMesh, compute_view_transformation)

INF = float("inf")


class ParseError(ValueError):
    """parse_scene::ParseError / obj_file::ParserError"""


# --- a loader with serde_yaml 0.9's scalar typing (below): only true/false are booleans, `1e3` is a float, `0x10` is 16
class _CoreLoader(yaml.SafeLoader):
    def construct_mapping(self, node, deep=False):
        # serde_yaml 0.9 (the reference pins 0.9.21, parse_scene/Cargo.toml:9) refuses a mapping that names a key twice
        # ("duplicate entry with key ..."); PyYAML would keep the last one
        seen = set()
        for key_node, _ in node.value:
            key = self.construct_object(key_node, deep=True)
            try:
                dup = key in seen
                seen.add(key)
            except TypeError:
                dup = False
            if dup:
                raise yaml.constructor.ConstructorError(None, None, f"duplicate entry with key {key!r}", key_node.start_mark)
        return super().construct_mapping(node, deep=deep)


# Plain scalars are typed the way serde_yaml 0.9 types them (de.rs, visit_untagged_scalar; the reference pins 0.9.21,
# parse_scene/Cargo.toml:9 -- the crate is not vendored, the rules below restate its published source: parity unpinned):
#   null: "", ~, null, Null, NULL;  bool: true / false in the three spellings;
#   integers: decimal with an optional sign, 0x / 0o / 0b with an optional sign before the prefix; an integer spelt with
#   leading zeros ("007") is a STRING (digits_but_not_number); integers beyond 64 bits fall through to the float rule;
#   floats: .inf / -.inf / .nan in the three spellings, else whatever Rust's f64::from_str accepts after an optional
#   "+", provided the value is FINITE ("inf", "nan", "1e999" stay strings);  everything else: a string.
_CoreLoader.yaml_implicit_resolvers = {}
_DEC = re.compile(r"[0-9]+", re.ASCII)
_RADIX = {"0x": (16, re.compile(r"[0-9a-fA-F]+", re.ASCII)), "0o": (8, re.compile(r"[0-7]+", re.ASCII)),
          "0b": (2, re.compile(r"[01]+", re.ASCII))}
_RUST_F64_PLAIN = re.compile(r"[+-]?(?:(?i:inf|infinity|nan)|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)", re.ASCII)


def _digits_but_not_number(v):
    t = v[1:] if v[:1] in ("-", "+") else v
    return len(t) > 1 and t[0] == "0" and t[1:].isascii() and t[1:].isdigit()


def _serde_int(v):
    neg = v.startswith("-")
    body = v[1:] if v[:1] in ("-", "+") else v
    if body[:1] in ("-", "+"):
        return None
    for prefix, (radix, rx) in _RADIX.items():
        if body.startswith(prefix) and rx.fullmatch(body[2:]):
            return -int(body[2:], radix) if neg else int(body[2:], radix)
    if _digits_but_not_number(v) or not _DEC.fullmatch(body):
        return None
    return -int(body) if neg else int(body)


def _serde_scalar(v):
    if v in ("", "~", "null", "Null", "NULL"):
        return None
    if v in ("true", "True", "TRUE"):
        return True
    if v in ("false", "False", "FALSE"):
        return False
    i = _serde_int(v)
    if i is not None and -(1 << 63) <= i < (1 << 64):
        return i
    if not _digits_but_not_number(v):
        unpositive = v[1:] if v.startswith("+") else v
        if v.startswith("+") and unpositive[:1] in ("-", "+"):
            return v
        if unpositive in (".inf", ".Inf", ".INF"):
            return INF
        if v in ("-.inf", "-.Inf", "-.INF"):
            return -INF
        if v in (".nan", ".NaN", ".NAN"):
            return float("nan")
        if _RUST_F64_PLAIN.fullmatch(unpositive):
            f = float(unpositive)
            if f == f and f not in (INF, -INF):
                return f
    return v


def _construct_plain(loader, node):
    v = loader.construct_scalar(node)
    return _serde_scalar(v) if node.style is None else v


_CoreLoader.add_constructor("tag:yaml.org,2002:str", _construct_plain)


def _as_f64(v):
    """serde_yaml Value::as_f64: any number (integers included), nothing else."""
    if isinstance(v, bool):
        return None
    if isinstance(v, (int, float)):
        return float(v)
    return None


def _get_f64(v, what="value"):
    r = _as_f64(v)
    if r is None:
        raise ParseError(f"Invalid value: {v!r} ({what})")
    return r


def _get_f64_vec(v, what="value"):
    if not isinstance(v, list):
        raise ParseError(f"Invalid value: {v!r} ({what})")
    return [_get_f64(x, what) for x in v]


def _as_u64(v):
    if isinstance(v, bool) or not isinstance(v, int) or v < 0:
        return None
    return v


def _f64_or(mapping, key, default):
    if not isinstance(mapping, dict):
        return default
    r = _as_f64(mapping.get(key))
    return default if r is None else r


@dataclass
class Scene:
    """parse_scene::Scene, parse_scene/src/lib.rs:86-91"""
    camera: Camera
    lights: List[object] = field(default_factory=list)
    objects: List[object] = field(default_factory=list)


# ----------------------------------------------------------------------------- parse_scene

def _parse_transformation(item):
    """parse_scene/src/lib.rs:118-192 -> (name, params)"""
    if not isinstance(item, list):
        raise ParseError(f"Invalid value: {item!r}")
    if not item:
        raise ParseError("Invalid parameters for transformation: Missing transformation name")
    name = item[0]
    if not isinstance(name, str):
        raise ParseError(f"Invalid value: {item!r}")
    if name in ("scale", "translate"):
        p = [_get_f64(x) for x in item[1:]]
        if len(p) < 3:
            raise ParseError(f"Invalid parameters for transformation: {name} needs 3 parameters")
        return (name, p[0], p[1], p[2])
    if name in ("rotate-x", "rotate-y", "rotate-z"):
        if len(item) < 2:
            raise ParseError("Invalid parameters for transformation: Missing transformation parameters")
        return (name, _get_f64(item[1]))
    raise ParseError(f"Invalid parameters for transformation: Unknown transformation name: {name}")


def _apply(t: Matrix, tr) -> Matrix:
    """construct_object's match arms, parse_scene/src/lib.rs:578-621 (Matrix fluent API: left-multiply)"""
    name = tr[0]
    if name == "scale":
        return t.scale(tr[1], tr[2], tr[3])
    if name == "translate":
        return t.translate(tr[1], tr[2], tr[3])
    if name == "rotate-x":
        return t.rotate_x(tr[1])
    if name == "rotate-y":
        return t.rotate_y(tr[1])
    return t.rotate_z(tr[1])


def _parse_material(desc) -> Material:
    """parse_scene/src/lib.rs:488-520"""
    d = desc if isinstance(desc, dict) else {}
    color = (1.0, 1.0, 1.0)
    if "color" in d:
        try:
            cv = _get_f64_vec(d["color"])
        except ParseError:
            cv = [1.0, 1.0, 1.0]  # .and_then(|x| x.get_f64_vec().ok()).unwrap_or_else(|| vec![1., 1., 1.])
        if len(cv) < 3:
            raise ParseError("Invalid value: color needs 3 components")
        color = (cv[0], cv[1], cv[2])
    cs = d.get("casts-shadow")
    return Material(
        color=color,
        ambient=_f64_or(d, "ambient", 0.1),
        diffuse=_f64_or(d, "diffuse", 0.9),
        specular=_f64_or(d, "specular", 0.9),
        shininess=_f64_or(d, "shininess", 200.0),
        reflectivity=_f64_or(d, "reflective", 0.0),
        transparency=_f64_or(d, "transparency", 0.0),
        refractive_index=_f64_or(d, "refractive-index", 1.0),
        casts_shadow=cs if isinstance(cs, bool) else True,
    )


def _copy_material(m: Material) -> Material:
    return Material(m.color, m.ambient, m.diffuse, m.specular, m.shininess, m.pattern, m.reflectivity,
                    m.transparency, m.refractive_index, m.casts_shadow)


def _parse_definition(desc, definitions):
    """parse_scene/src/lib.rs:194-342 -> (name, ('material', Material) | ('transforms', [..]))"""
    name = desc.get("define")
    if not isinstance(name, str):
        raise ParseError(f"Invalid value: {desc!r}")
    if "value" not in desc:
        raise ParseError("Invalid parameters for a constant definition: Missing 'value' field")
    value = desc["value"]
    if isinstance(value, dict):
        if "extend" in desc:
            ext = desc["extend"]
            if not isinstance(ext, str):
                raise ParseError(f"Invalid value: {ext!r}")
            if ext not in definitions:
                raise ParseError(f"Invalid parameters for a constant definition: Definition {ext} not found")
            kind, original = definitions[ext]
            if kind != "material":
                raise ParseError("Cannot extend a definition that does not have a material")
            m = _copy_material(original)
            new = _parse_material(value)
            for key, attr in (("color", "color"), ("ambient", "ambient"), ("diffuse", "diffuse"),
                              ("specular", "specular"), ("shininess", "shininess"),
                              ("reflective", "reflectivity"), ("refractive-index", "refractive_index"),
                              ("transparency", "transparency"), ("casts-shadow", "casts_shadow")):
                if key in value:  # Q16: only keys present in `value` override
                    setattr(m, attr, getattr(new, attr))
            return name, ("material", m)
        return name, ("material", _parse_material(value))
    if isinstance(value, list):
        trs = []
        for item in value:
            if isinstance(item, str):
                if item not in definitions:
                    raise ParseError(f"Invalid parameters for a constant definition: Definition {item} not found")
                kind, original = definitions[item]
                if kind != "transforms":
                    raise ParseError("Cannot extend a definition that does not have transformations")
                trs.extend(original)
            else:
                trs.append(_parse_transformation(item))
        return name, ("transforms", trs)
    raise ParseError("Definition 'value' must be a mapping or a sequence")


def _construct_camera(d, width=None, height=None) -> Camera:
    """parse_scene/src/lib.rs:344-397; width/height overrides are this build's addition"""
    w = _as_u64(d.get("width"))
    h = _as_u64(d.get("height"))
    if w is None:
        raise ParseError("Invalid parameters for camera: Missing camera width / must be an integer")
    if h is None:
        raise ParseError("Invalid parameters for camera: Missing camera height / must be an integer")
    for key in ("field-of-view", "from", "to", "up"):
        if key not in d:
            raise ParseError(f"Invalid parameters for camera: Missing '{key}'")
    fov = _get_f64(d["field-of-view"])
    frm, to, up = _get_f64_vec(d["from"]), _get_f64_vec(d["to"]), _get_f64_vec(d["up"])
    view = compute_view_transformation(frm[:3], to[:3], up[:3])
    return Camera(width or w, height or h, fov, view)


def _construct_object(d, definitions, base_dir):
    """parse_scene/src/lib.rs:522-733"""
    kind = d.get("add")
    if not isinstance(kind, str):
        raise ParseError("Invalid value: Object's 'add' must be a string")
    if "material" in d:
        md = d["material"]
        if isinstance(md, str):
            if md not in definitions:
                raise ParseError(f"Invalid value: No such defintion: {md}")
            k, m = definitions[md]
            if k != "material":
                raise ParseError(f"Invalid value: Definition '{md}' does not refer to a material")
            material = _copy_material(m)
        else:
            material = _parse_material(md)
    else:
        material = Material()

    t = Matrix.identity()
    trs = d.get("transform")
    if isinstance(trs, list):
        for value in trs:
            if isinstance(value, str):
                if value not in definitions:
                    raise ParseError(f"Invalid value: No such defintion: {value}")
                k, lst = definitions[value]
                if k != "transforms":
                    raise ParseError(f"Invalid value: Definition '{value}' does not refer to a set of transformations")
                for tr in lst:
                    t = _apply(t, tr)
            else:
                t = _apply(t, _parse_transformation(value))

    if kind == "sphere":
        return Sphere(t, material)
    if kind == "plane":
        return Plane(t, material)
    if kind == "cube":
        return Cube(t, material)
    if kind in ("cylinder", "cone"):
        minimum = _f64_or(d, "minimum", -INF)
        maximum = _f64_or(d, "maximum", INF)
        closed = d.get("closed") if isinstance(d.get("closed"), bool) else False
        ctor = Cylinder if kind == "cylinder" else Cone
        return ctor(t, material, minimum, maximum, closed)
    if kind == "group":
        # Q14: the parsed `material` is dropped; groups carry Some(Material::default()) (shape.rs:1909)
        g = Group(transformation=t)
        if "children" not in d:
            raise ParseError("Invalid parameters for group: Missing 'children' in group definition")
        children = d["children"]
        if not isinstance(children, list):
            raise ParseError("Invalid parameters for group: Group 'children' must be a sequence")
        nested, others = [], []
        for c in children:
            if not isinstance(c, dict):
                raise ParseError("Invalid parameters for group: Each value in a group's 'children' must be a mapping")
            child = _construct_object(c, definitions, base_dir)
            # A nested YAML group is added to its parent as soon as it is constructed (lib.rs:677-679); every
            # other child only after the whole `children` list has been built (lib.rs:707-712): nested groups
            # therefore come first in Group::children order.
            (nested if c.get("add") == "group" else others).append(child)
        g.children = nested + others
        return g
    if kind == "obj":
        fn = d.get("file")
        if fn is None:
            raise ParseError("Invalid parameters for obj: Missing 'file' field")
        if not isinstance(fn, str):
            raise ParseError("Invalid value: obj 'file' must be a string")
        return parse_obj_file(_resolve(fn, base_dir), material, t).default_group
    raise ParseError(f"unimplemented object type: {kind}")  # parse_scene/src/lib.rs:729 unimplemented!()


def _resolve(filename, base_dir):
    """The reference opens the path relative to the CWD (samples/teapot.yaml:34 expects the repo root).  This
    build also looks next to the YAML file and one directory above it."""
    if os.path.isabs(filename) or os.path.exists(filename):
        return filename
    for root in ([base_dir, os.path.dirname(base_dir)] if base_dir else []):
        cand = os.path.join(root, filename)
        if os.path.exists(cand):
            return cand
    return filename


def parse_scene(text: str, base_dir: Optional[str] = None, width: Optional[int] = None,
                height: Optional[int] = None) -> Scene:
    """parse_scene/src/lib.rs:735-784"""
    try:
        seq = yaml.load(text, Loader=_CoreLoader)
    except yaml.YAMLError as e:
        raise ParseError(f"YAML deserialization error: {e}") from e
    if not isinstance(seq, list):
        raise ParseError("YAML deserialization error: expected a sequence")
    camera = None
    lights, objects = [], []
    definitions = {}
    order = []
    for item in seq:
        if not isinstance(item, dict):
            raise ParseError(f"Invalid YAML object: Expected mapping, found {item!r}")
        add = item.get("add")
        if isinstance(add, str):
            if add == "camera":
                camera = _construct_camera(item, width, height)
            elif add == "light":
                if "at" not in item:
                    raise ParseError("Invalid parameters for light: Missing 'at'")
                if "intensity" not in item:
                    raise ParseError("Invalid parameters for light: Missing 'intensity'")
                lights.append(PointLight(tuple(_get_f64_vec(item["at"])[:3]), tuple(_get_f64_vec(item["intensity"])[:3])))
            elif add == "area-light":
                for key in ("corner", "full_uvec", "usteps", "full_vvec", "vsteps", "intensity"):
                    if key not in item:
                        raise ParseError(f"Invalid parameters for light: Missing '{key}'")
                us, vs = _as_u64(item["usteps"]), _as_u64(item["vsteps"])
                if us is None or vs is None:
                    raise ParseError("Invalid parameters for light: Invalid value for usteps/vsteps")
                lights.append(AreaLight(tuple(_get_f64_vec(item["corner"])[:3]),
                                        tuple(_get_f64_vec(item["full_uvec"])[:3]), us,
                                        tuple(_get_f64_vec(item["full_vvec"])[:3]), vs,
                                        tuple(_get_f64_vec(item["intensity"])[:3])))
            elif add in ("sphere", "plane", "cube", "cylinder", "cone", "group", "obj"):
                objects.append(_construct_object(item, definitions, base_dir))
            else:
                raise ParseError(f"Encountered unknown item: {add}")
        elif isinstance(item.get("define"), str):
            name, value = _parse_definition(item, definitions)
            # definitions.iter().find(..) returns the FIRST definition with that name
            if name not in definitions:
                definitions[name] = value
                order.append(name)
    if camera is None:
        raise ParseError("The scene is missing camera defintion")
    return Scene(camera, lights, objects)


def parse_scene_file(path: str, width: Optional[int] = None, height: Optional[int] = None) -> Scene:
    with open(path, "r") as f:
        return parse_scene(f.read(), os.path.dirname(os.path.abspath(path)), width, height)


# ----------------------------------------------------------------------------- obj_file

@dataclass
class ParsedObj:
    """obj_file::ParsedObj, obj_file/src/lib.rs:25-32"""
    ignored_lines: int
    default_group: Group
    vertices: List[tuple]
    normals: List[tuple]
    groups: dict


class _MeshBuilder:
    """Collects consecutive triangles of one group into TriangleMesh runs (flat / smooth kept apart)."""

    def __init__(self, group: Group, material: Material):
        self.group, self.material = group, material
        self.pts, self.nrm, self.smooth = [], [], None

    def flush(self):
        if self.pts:
            pts = np.array(self.pts, dtype=np.float64).reshape(-1, 9)
            nrm = np.array(self.nrm, dtype=np.float64).reshape(-1, 9) if self.smooth else None
            # Triangle::new(.., Material::default()) -- the parent group's material wins (shape.rs:1520-1532)
            self.group.children.append(TriangleMesh(pts, nrm, Material()))
        self.pts, self.nrm, self.smooth = [], [], None

    def add(self, p1, p2, p3, normals):
        smooth = normals is not None
        if self.smooth is not None and smooth != self.smooth:
            self.flush()
        self.smooth = smooth
        self.pts.append([*p1, *p2, *p3])
        if smooth:
            self.nrm.append([*normals[0], *normals[1], *normals[2]])


# the strings Rust's f64::from_str accepts (Python's float() also takes "1_0" and non-ASCII digits)
_RUST_F64 = re.compile(r"[+-]?(?:(?i:inf|infinity|nan)|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)", re.ASCII)


def _parse_f64(tok, line_no):
    if not _RUST_F64.fullmatch(tok):
        raise ParseError(f"invalid input: Invalid vertex record on line {line_no}: {tok}")
    return float(tok)


def _parse_usize(tok, what, line_no):
    if not re.fullmatch(r"\+?[0-9]+", tok):
        raise ParseError(f"invalid input: Invalid {what} index for face record on line {line_no}: {tok}")
    return int(tok)


def parse_obj_string(text: str, material: Optional[Material] = None,
                     transformation: Optional[Matrix] = None) -> ParsedObj:
    """obj_file/src/lib.rs:40-230"""
    material = material if material is not None else Material()
    default_group = Group(transformation=transformation or Matrix.identity(), material=material)
    ignored = 0
    vertices, normals = [], []
    groups = {}
    builders = {id(default_group): _MeshBuilder(default_group, material)}
    current = None
    for n, line in enumerate(text.splitlines()):
        tokens = line.split()
        if not tokens:
            continue
        line_no = n + 1
        head = tokens[0]
        if head in ("v", "vn"):
            if len(tokens) >= 4:
                x, y, z = (_parse_f64(t, line_no) for t in tokens[1:4])
                (vertices if head == "v" else normals).append((x, y, z))
            # else: "Malformed vertex record" is printed and the line skipped (obj_file/src/lib.rs:122-124)
        elif head == "f":
            face, nidx = [], []
            for tok in tokens[1:]:
                sub = tok.split("/")
                vi = _parse_usize(sub[0], "vertex", n)
                if vi < 1 or vi > len(vertices):
                    raise ParseError(f"missing vertex at index {vi}")
                face.append(vertices[vi - 1])
                if len(sub) > 2:
                    ni = _parse_usize(sub[2], "normal", n)
                    if ni < 1:
                        raise ParseError(f"invalid input: normal index 0 on line {line_no}")
                    nidx.append(ni - 1)
            if len(face) < 2:
                raise ParseError(f"invalid input: face with {len(face)} vertices on line {line_no}")
            target = current if current is not None else default_group
            b = builders[id(target)]
            for i in range(1, len(face) - 1):
                if not nidx:
                    b.add(face[0], face[i], face[i + 1], None)
                else:
                    # Q11: every fan triangle takes the face's normals 0, 1, 2 (obj_file/src/lib.rs:169-171)
                    if len(nidx) < 3 or max(nidx[:3]) >= len(normals):
                        raise ParseError(f"invalid input: missing normal on line {line_no}")
                    b.add(face[0], face[i], face[i + 1], (normals[nidx[0]], normals[nidx[1]], normals[nidx[2]]))
        elif head == "g":
            if len(tokens) < 2:
                raise ParseError(f"invalid input: Malformed named group record on line {n}: missing group name")
            # triangles added so far keep their place in the parent's child order
            builders[id(default_group)].flush()
            if current is not None:
                builders[id(current)].flush()
            g = Group(transformation=Matrix.identity(), material=material)
            default_group.children.append(g)
            groups[tokens[1]] = g
            builders[id(g)] = _MeshBuilder(g, material)
            current = g
        else:
            ignored += 1
    for b in builders.values():
        b.flush()
    return ParsedObj(ignored, default_group, vertices, normals, groups)


def parse_obj_file(filename: str, material: Optional[Material] = None,
                   transformation: Optional[Matrix] = None) -> ParsedObj:
    """obj_file/src/lib.rs:232-242"""
    try:
        with open(filename, "r") as f:
            text = f.read()
    except OSError as e:
        raise ParseError(f"failed to open the file: {e}") from e
    return parse_obj_string(text, material, transformation)
