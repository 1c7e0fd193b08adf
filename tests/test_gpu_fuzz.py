"""GPU parity on SEEDED RANDOM scenes: every shape kind, nested groups, Csg trees, patterns, glass and mirrors, point
and area lights, random transforms (rotations, non-uniform scales, shears) and cameras -- combinations no hand-written
scene of test_gpu_features.py holds.  Each scene is rendered through the C ABI under both schedules, with the BVH and
with the linear scan, and by the oracle (oracle/rtc_oracle.c: the reference's algorithm, world.rs:30-148).

The generator is deterministic per seed (numpy Generator, PCG64): a failure names its seed.
"""
import numpy as np
import pytest

import scene_feed as sf
from fuzz_scenes import random_scene
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
