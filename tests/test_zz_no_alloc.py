"""rtc_render_device_static: the entry point that allocates nothing (its work buffer is a static device array of the
library's code object).  Checked through the C ABI against rtc_render_device, the existing entry point it is a variant
of, on two calls in a row -- the second on a smaller input, so that whatever the first call left in the shared array
(tile counters, mailbox states, overflow flag) must not leak into it."""
import math

import numpy as np
import pytest
import torch  # before anything initialises HIP: torch brings its own copy of the runtime (the import order matters)

import scene_feed as sf
from scene_feed import rtc

pytestmark = pytest.mark.gpu

M = rtc.Matrix


def _glass(w, h, n_side):
    objs = [rtc.Plane(material=rtc.Material(color=(0.9, 0.9, 0.9), reflectivity=0.4))]
    for i in range(n_side):
        for j in range(n_side):
            objs.append(rtc.Sphere(M.scaling(0.45, 0.45, 0.45).translate(i - n_side / 2 + 0.5, 0.45 + 0.9 * ((i + j) % 2), j - 1.0),
                                   rtc.Material(transparency=0.9, reflectivity=0.9, refractive_index=1.5, diffuse=0.1,
                                                color=(0.6 + 0.1 * (i % 3), 1.0, 0.7))))
    cam = rtc.Camera(w, h, math.pi / 3, rtc.compute_view_transformation((0, 2.5, -7), (0, 1, 0), (0, 1, 0)))
    return rtc.World(objs, [rtc.PointLight((-6, 8, -8), (1, 1, 1))]), cam


def _both(scene, cam, **kw):
    a = torch.full((cam.vsize, cam.hsize, 3), -1.0, dtype=torch.float64, device="cuda:0")
    b = torch.full((cam.vsize, cam.hsize, 3), -2.0, dtype=torch.float64, device="cuda:0")
    scene.render_device(a.data_ptr(), 0, **kw)
    scene.check()
    scene.render_device_static(b.data_ptr(), 0, **kw)
    scene.check()
    return a.cpu().numpy(), b.cpu().numpy()


def test_static_scratch_entry_point_equals_render_device_twice_in_a_row():
    big = _glass(256, 160, 4)      # 640 tiles: the persistent grid draws tiles from the counters in the scratch
    small = _glass(72, 40, 2)      # second call: a smaller input on the same GPU, i.e. the same static array
    scenes = [(cam.compile(world, device=0), cam) for world, cam in (big, small)]
    try:
        for scene, cam in scenes:
            # persistent schedule without the ray exchange: bit-reproducible, so the two entry points must agree exactly
            a, b = _both(scene, cam, schedule=2, exchange=1)
            assert np.array_equal(a, b) and float(np.abs(a).sum()) > 0 and not np.isnan(b).any()
            # with the exchange (mailboxes in the scratch): the same arithmetic, a few f64 additions in another order
            a2, b2 = _both(scene, cam, schedule=2)
            assert float(np.max(np.abs(a2 - b2))) < 1e-12 and float(np.max(np.abs(a - b2))) < 1e-12
            # static schedule: only the overflow flag of the scratch is used
            a3, b3 = _both(scene, cam, schedule=1, exchange=1)
            assert np.array_equal(a3, b3) and np.array_equal(a3, a)
            # (default: quarter tiles with in-wave hand-over -- deterministic, and within rounding of the above)
            a4, b4 = _both(scene, cam, schedule=1)
            assert np.array_equal(a4, b4) and float(np.max(np.abs(a4 - a))) < 1e-12
        # the overflow flag lives in the shared array too: a capped launch reports, the next clean one does not
        scene, cam = scenes[1]
        buf = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float64, device="cuda:0")
        # (exchange off: the cap then bounds a whole pixel's ray tree, whatever the timing)
        scene.render_device_static(buf.data_ptr(), 0, schedule=2, max_rays_per_pixel=6, exchange=1)
        with pytest.raises(rtc._native.RtcError) as e:
            scene.check()
        assert e.value.code == rtc._native.ERR_OVERFLOW
        scene.render_device_static(buf.data_ptr(), 0, schedule=2)
        scene.check()
        # "allocates nothing": device memory that is free before a static call is free after it (every kernel variant
        # used here has been launched above, so no code object is loaded inside the bracket either)
        for kw in (dict(schedule=2), dict(schedule=1), dict(schedule=2, exchange=1)):
            torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info()
            scene.render_device_static(buf.data_ptr(), 0, **kw)
            scene.check()
            free1, _ = torch.cuda.mem_get_info()
            assert free1 == free0, (kw, free0, free1)
    finally:
        for scene, _ in scenes:
            scene.close()


def test_static_scratch_entry_point_rejects_what_it_cannot_hold():
    world, cam = _glass(64, 40, 2)
    scene = cam.compile(world, device=0)
    try:
        buf = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float64, device="cuda:0")
        with pytest.raises(rtc._native.RtcError) as e:
            scene.render_device_static(buf.data_ptr(), 0, collect_stats=True)
        assert e.value.code == rtc._native.ERR_INVALID
    finally:
        scene.close()
    # a frame beyond 2^26 tiles is refused before anything is touched (both entry points share the check)
    huge = rtc.Camera(65536, 65600, math.pi / 3, rtc.Matrix())
    sc = huge.compile(rtc.World([rtc.Sphere()], [rtc.PointLight((0, 5, -5), (1, 1, 1))]), device=0)
    try:
        with pytest.raises(rtc._native.RtcError) as e:
            sc.render_device_static(8, 0)  # the pointer is never dereferenced
        assert e.value.code == rtc._native.ERR_INVALID and "2^26" in str(e.value)
    finally:
        sc.close()
