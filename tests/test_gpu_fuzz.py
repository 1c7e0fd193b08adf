"""GPU parity on SEEDED RANDOM scenes: every shape kind, nested groups, Csg trees, patterns, glass and mirrors, point
and area lights, random transforms (rotations, non-uniform scales, shears) and cameras -- combinations no hand-written
scene of test_gpu_features.py holds.  Each scene is rendered through the C ABI under both schedules, with the BVH and
with the linear scan, and by the oracle (oracle/rtc_oracle.c: the reference's algorithm, world.rs:30-148).

The generator is deterministic per seed (numpy Generator, PCG64): a failure names its seed.
"""
import numpy as np
import pytest

import scene_feed as sf
from fuzz_scenes import crowd_scene, random_scene
from scene_feed import rtc

pytestmark = pytest.mark.gpu

F64_TOL = 1e-9


def _render(scene, **kw):
    """(canvas, overflowed): RTC_ERR_OVERFLOW keeps the canvas that was written (include/rtc.h)"""
    try:
        return scene.render(**kw), False
    except rtc.RtcError as e:
        assert e.code == rtc._native.ERR_OVERFLOW, e
        return scene.last_canvas, True


@pytest.mark.parametrize("seed", list(range(1, 41)))
def test_random_scene_matches_oracle(seed):
    world, cam, depth = random_scene(seed)
    ow = sf.to_oracle(world, cam)
    ref = ow.render(depth=depth)
    # Pixels on which the reference does not terminate: refraction keeps the bounce budget (world.rs:142), a ray caught
    # around the apex of a glass double cone refracts without end, the reference overflows its stack (the oracle stops
    # at its nesting guard and marks the pixel).  No reference value exists for them.  This library either reaches its
    # ray cap (RTC_ERR_OVERFLOW, NaN in the pixel) or ends the chain once the ray's weight -- the product of the
    # transparencies on its way -- has underflowed to exactly 0.0 and gives the limit of the series; which of the two
    # depends on the schedule (a lane that takes over part of a ray tree counts its rays from 1, include/rtc.h).
    # Required: NaN at most THERE, and every other pixel of the frame still equal to the reference's.
    diverged = np.isnan(ref).any(axis=-1)
    assert int(diverged.sum()) == ow.last_stats["diverged"]
    scene = cam.compile(world, device=0)
    try:
        info = scene.info()
        static, over = _render(scene, depth=depth, schedule=1, exchange=1)
        default, over_d = _render(scene, depth=depth)
        persistent, over_p = _render(scene, depth=depth, schedule=2)
        brute, _ = _render(scene, depth=depth, schedule=1, exchange=1, use_bvh=2)
        bvh, _ = _render(scene, depth=depth, schedule=1, exchange=1, use_bvh=1)
    finally:
        scene.close()
    good = ~diverged
    rep = sf.compare(static[good], ref[good])
    print(f"[fuzz] seed {seed}: {info.get('n_prims', '?')} primitives, {len(world.lights)} lights, depth {depth}, "
          f"{cam.hsize}x{cam.vsize}, oracle {ow.last_stats['seconds']:.2f} s, {int(diverged.sum())} diverged pixels, "
          f"overflow reported {over}/{over_d}/{over_p}: {rep}")
    assert rep["nan"] == 0 and rep["max_8bit"] <= 1 and rep["pixels_over_1"] == 0 and rep["max_abs"] < F64_TOL, rep
    if not diverged.any():
        assert not (over or over_d or over_p)
    for img in (static, default, persistent, brute, bvh):
        assert not np.isnan(img[good]).any()
    # the hierarchy returns what the linear scan returns, bit for bit (NaN where both gave up)
    assert np.array_equal(brute, bvh, equal_nan=True) and np.array_equal(static, bvh, equal_nan=True)
    # the other schedules trace the same rays; parts of split pixels are added in another order
    assert float(np.max(np.abs(default[good] - static[good]))) < 1e-12
    assert float(np.max(np.abs(persistent[good] - static[good]))) < 1e-12
    assert np.all(static[-1] == 0.0) and np.all(static[:, -1] == 0.0)


@pytest.mark.parametrize("seed", [7, 8])
def test_crowd_of_mixed_shapes_in_a_quantised_hierarchy(seed):
    # 36 000 small shapes of every kind: the hierarchy is stored quantised (more than 16 384 inner nodes), its leaves
    # mix compact sphere records with full records of cubes, cylinders, cones and triangles under groups -- the stress
    # scene of the bench has spheres and one mesh only.  Auto schedule = persistent (more than 64 transparent shapes).
    world, cam = crowd_scene(seed)
    scene = cam.compile(world, device=0)
    try:
        info = scene.info()
        assert info["n_bvh_nodes"] > 16384 and info["n_unbounded"] > 1
        default = scene.render()
        ms = scene.last_stats["kernel_ms"]
        static = scene.render(schedule=1, exchange=1)
        persistent_whole = scene.render(schedule=2, exchange=1)
        ntiles = (cam.vsize + 7) // 8
        kw = dict(tile_rows=8, tile_index=ntiles // 2, tile_count=ntiles, depth=2, schedule=1, exchange=1)
        bvh = scene.render(use_bvh=1, **kw)
        brute = scene.render(use_bvh=2, **kw)
    finally:
        scene.close()
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(0, cam.hsize - 1, 2000), rng.integers(0, cam.vsize - 1, 2000)], axis=1)
    ow = sf.to_oracle(world, cam)
    ref = ow.render_pixels(xy)
    assert ow.last_stats["diverged"] == 0
    rep = sf.compare(static[xy[:, 1], xy[:, 0]], ref)
    print(f"[fuzz] crowd seed {seed}: {info['n_prims']} primitives, {info['n_bvh_nodes']} nodes, {info['n_unbounded']} "
          f"unbounded, kernel {ms:.2f} ms, oracle {ow.last_stats['seconds']:.1f} s for {len(xy)} pixels: {rep}")
    assert rep["nan"] == 0 and rep["max_8bit"] <= 1 and rep["pixels_over_1"] == 0 and rep["max_abs"] < F64_TOL, rep
    assert float(np.abs(ref).sum()) > 0 and float(np.abs(bvh).sum()) > 0
    assert np.array_equal(bvh, brute)
    assert np.array_equal(static, persistent_whole)  # one lane per pixel from start to end: bit-equal schedules
    assert float(np.max(np.abs(default - static))) < 1e-12
