"""GPU parity AT THE SIZES THAT ARE BENCHMARKED ("parity is thin where it matters most").

BASELINE.json north_star: "cover.yaml and teapot.yaml at 3840x2160 rendered within tolerance of the reference".  The
frames bench.py times are checked here against the oracle on seeded-random pixels (the oracle is brute force: a whole
4K frame of cover.yaml is minutes of CPU time, 10^5 pixels are seconds), and the stress frame also on pixels picked
inside glass spheres -- the deep ray trees that the persistent schedule's end-of-frame exchange splits between lanes,
i.e. the only part of a frame whose arithmetic order is timing dependent.
"""
import numpy as np
import pytest

import scene_feed as sf
from scene_feed import rtc

pytestmark = pytest.mark.gpu

F64_TOL = 1e-9


def _random_pixels(cam, n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, cam.hsize - 1, n), rng.integers(0, cam.vsize - 1, n)], axis=1)


def _check_pixels(world, cam, xy, label, **opts):
    scene = cam.compile(world, device=0)
    try:
        gpu = scene.render(**opts)
        ms = scene.last_stats["kernel_ms"]
    finally:
        scene.close()
    ow = sf.to_oracle(world, cam)
    ref = ow.render_pixels(xy)
    got = gpu[xy[:, 1], xy[:, 0]]
    rep = sf.compare(got, ref)
    print(f"[parity] {label}: {len(xy)} oracle pixels ({ow.last_stats['seconds']:.1f} s of oracle time), "
          f"kernel {ms:.3f} ms: {rep}")
    assert float(np.abs(ref).sum()) > 0
    assert rep["nan"] == 0 and rep["max_8bit"] <= 1 and rep["pixels_over_1"] == 0 and rep["max_abs"] < F64_TOL, rep
    assert not np.isnan(gpu).any()
    assert np.all(gpu[-1] == 0.0) and np.all(gpu[:, -1] == 0.0)  # Q1
    return gpu


@pytest.mark.parametrize("name,w,h,n", [("cover", 3840, 2160, 100_000), ("teapot", 3840, 2160, 40_000),
                                        ("cover", 1920, 1080, 60_000), ("reflective_world", 1920, 1080, 200_000),
                                        ("ice_cream", 1920, 1080, 200_000)])
def test_sample_scene_at_benchmark_size(name, w, h, n):
    # BASELINE.json configs[1..3] and the north_star's two 4K target frames, at the resolutions bench.py times them at
    world, cam = sf.load(name, w, h)
    _check_pixels(world, cam, _random_pixels(cam, n, 0xC0FFEE + w), f"{name} {w}x{h}")


def test_teapot_4k_sharded_like_config_3_sums_to_the_frame():
    # BASELINE.json configs[3]: "teapot.yaml ... at 3840x2160, row-tile sharded over 8 GPUs": the eight shares of the
    # mesh scene (groups + BVH + interleaved 8-row tiles) add up to the one-launch frame, bit for bit
    world, cam = sf.load("teapot", 3840, 2160)
    scene = cam.compile(world, device=0)
    try:
        full = scene.render()
        acc = np.zeros_like(full)
        for r in range(8):
            acc += scene.render(tile_rows=8, tile_index=r, tile_count=8)
    finally:
        scene.close()
    assert np.array_equal(full, acc)


def test_teapot_4k_on_eight_replicas_equals_one_device():
    # the in-library multi-GPU path (rtc_scene_commit_devices + rtc_render) on the sharded config itself: eight replicas
    # -- all on device 0 on a one-GPU box: every launch, peer copy and the assembly run -- against the single-device frame
    world, cam = sf.load("teapot", 3840, 2160)
    one = cam.compile(world, device=0)
    try:
        a = one.render()
    finally:
        one.close()
    many = cam.compile(world, device=[0] * 8)
    try:
        b = many.render()
    finally:
        many.close()
    assert np.array_equal(a, b)


@pytest.mark.parametrize("w,h,n_random,n_glass", [(1920, 1080, 24, 12), (3840, 2160, 36, 12)])
def test_stress_frame_random_and_glass_pixels(w, h, n_random, n_glass):
    # the headline workload (100k spheres + teapot, depth 5, persistent schedule WITH the exchange) on seeded-random
    # pixels and on pixels at the projected centres of the nearest glass spheres
    world, cam = rtc.stress_scene.build(100_000, w, h)
    glass = sf.glass_pixels(world, cam, n_glass)
    assert len(glass) == n_glass
    xy = np.concatenate([_random_pixels(cam, n_random, 0x5EED + w), glass])
    _check_pixels(world, cam, xy, f"stress 100k {w}x{h} depth 5 (random + glass pixels)")
