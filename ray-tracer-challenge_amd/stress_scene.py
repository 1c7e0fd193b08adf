"""Synthetic stress scene of BASELINE.json configs[4]: N random glass/mirror spheres + the teapot mesh.

Nothing in the reference defines this scene; every parameter below is this build's choice (SURVEY.md 8(d)
item 5) and the generator, not the data, is what is committed.  The objects are ordinary reference objects
(spheres with `translation * scaling` transforms, a plane, the teapot.yaml `obj` group), so the scene goes
through the same boundary (include/rtc.h) and the same oracle as the sample scenes.
"""
import math
import os

import numpy as np

from . import scene_yaml
from .api import Camera, Group, Material, Matrix, Plane, PointLight, Shape, World, compute_view_transformation
from . import _native as N

SEED = 0x5EED1234
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_doubles(seed: int, count: int) -> np.ndarray:
    """`count` doubles in [0, 1) from the splitmix64 stream started at `seed` (53 high bits / 2^53)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GAMMA * np.arange(1, count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _matrix_from_rows(rows) -> Matrix:
    m = Matrix()
    m.m = np.asarray(rows, dtype=np.float64).reshape(4, 4)
    return m


def build(n_spheres: int = 100_000, width: int = 3840, height: int = 2160, seed: int = SEED,
          with_teapot: bool = True, scenes_dir: str = None):
    """(World, Camera) of the stress scene.  Seven draws per sphere: x, y, z, radius, r, g, b."""
    u = splitmix64_doubles(seed, 7 * n_spheres).reshape(n_spheres, 7)
    cx = -60.0 + 120.0 * u[:, 0]
    cy = 0.5 + 29.5 * u[:, 1]
    cz = -60.0 + 120.0 * u[:, 2]
    rad = 0.15 + 0.45 * u[:, 3]
    col = 0.2 + 0.8 * u[:, 4:7]
    objects = []
    # floor
    objects.append(Plane(material=Material(color=(0.6, 0.6, 0.65), reflectivity=0.3)))
    for i in range(n_spheres):
        # translation(x, y, z) * scaling(r, r, r) -- the product of those two builder matrices is exactly this
        t = _matrix_from_rows([[rad[i], 0.0, 0.0, cx[i]], [0.0, rad[i], 0.0, cy[i]], [0.0, 0.0, rad[i], cz[i]],
                               [0.0, 0.0, 0.0, 1.0]])
        c = (float(col[i, 0]), float(col[i, 1]), float(col[i, 2]))
        if i % 2 == 0:  # mirror
            m = Material(color=c, reflectivity=0.9, diffuse=0.1, specular=1.0, shininess=300.0)
        else:  # glass
            m = Material(color=c, transparency=0.9, reflectivity=0.9, refractive_index=1.5, diffuse=0.1,
                         ambient=0.05)
        objects.append(Shape(N.SPHERE, t, m))
    if with_teapot:
        # the `obj` entry of scenes/teapot.yaml (rotate-x -pi/2, scale 0.1), lifted above the floor spheres
        root = scenes_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")
        t = Matrix.identity().rotate_x(-1.5707963268).scale(0.5, 0.5, 0.5).translate(0.0, 2.0, -30.0)
        teapot = scene_yaml.parse_obj_file(os.path.join(root, "obj", "teapot.obj"), Material(), t).default_group
        objects.append(teapot)
    lights = [PointLight((-10.0, 60.0, -40.0), (1.0, 1.0, 1.0))]
    cam = Camera(width, height, math.pi / 3.0,
                 compute_view_transformation((0.0, 25.0, -90.0), (0.0, 8.0, 0.0), (0.0, 1.0, 0.0)))
    return World(objects, lights), cam
