"""CPU: the oracle's nesting guard, and the host side of the library (flattening, hierarchy build) on the fuzz scenes.

refracted_color keeps the bounce budget (world.rs:142), so the reference does not terminate on a ray that is caught
between refracting surfaces -- it overflows its stack.  The oracle must survive such a scene (it runs inside the test
process), say which pixels were concerned, and still answer the others."""
import numpy as np
import pytest

import scene_feed as sf
from fuzz_scenes import random_scene


def test_oracle_marks_pixels_the_reference_cannot_finish():
    # seed 10: a glass double cone; some tens of pixels look at its apex
    world, cam, depth = random_scene(10)
    ow = sf.to_oracle(world, cam)
    ref = ow.render(depth=depth)
    bad = np.isnan(ref).any(axis=-1)
    assert ow.last_stats["diverged"] == int(bad.sum()) > 0
    assert bad.sum() < 0.05 * bad.size and np.isfinite(ref[~bad]).all() and float(np.abs(ref[~bad]).sum()) > 0
    # the same pixels through the pixel-list entry point, and a second frame afterwards is not affected by the first
    ys, xs = np.nonzero(bad)
    px = ow.render_pixels(np.stack([xs, ys], axis=1), depth=depth)
    assert np.isnan(px).all()
    ys, xs = np.nonzero(~bad)
    pick = np.stack([xs[::97], ys[::97]], axis=1)
    assert np.array_equal(ow.render_pixels(pick, depth=depth), ref[pick[:, 1], pick[:, 0]])


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13, 21, 34])
def test_fuzz_scene_is_deterministic_and_commits_on_the_host(seed):
    world, cam, depth = random_scene(seed)
    world2, cam2, depth2 = random_scene(seed)
    a = sf.to_oracle(world, cam).render(depth=depth)
    b = sf.to_oracle(world2, cam2).render(depth=depth2)
    assert np.array_equal(a, b, equal_nan=True) and depth == depth2
    # rtc_scene_commit(device = -1): flattening + hierarchy build without a GPU
    scene = cam.compile(world, device=-1)
    try:
        info = scene.info()
    finally:
        scene.close()
    assert info["n_prims"] > 0 and info["hsize"] == cam.hsize
