"""GPU parity beyond the sample scenes: every feature of the path that the reference's own tests exercise but the
YAML samples do not -- patterns (pattern.rs), area lights (light.rs:79-134), capped / truncated cylinders and cones
(Q7, Q8, Q10), nested groups with per-level normal renormalisation (shape.rs:82-103), flat triangles, overlapping
glass (n1/n2, intersection.rs:396-443), casts_shadow = false (Q5), several lights with secondary rays (Q3), bounce
budgets 0..5, portrait cameras.  Each scene is rendered through the C ABI on the GPU and by the oracle."""
import math

import numpy as np
import pytest

import scene_feed as sf
from scene_feed import rtc

pytestmark = pytest.mark.gpu

M = rtc.Matrix
PI = math.pi


def _cam(w=96, h=64, frm=(0, 2.5, -7), to=(0, 1, 0), fov=PI / 3):
    return rtc.Camera(w, h, fov, rtc.compute_view_transformation(frm, to, (0, 1, 0)))


def _check(world, cam, label, **opts):
    scene = cam.compile(world, device=0)
    try:
        gpu = scene.render(**opts)
        gpu2 = scene.render(schedule=2, **opts)  # the persistent schedule
    finally:
        scene.close()
    ow = sf.to_oracle(world, cam)
    ref = ow.render() if "depth" not in opts else ow.render(depth=opts["depth"])
    rep = sf.compare(gpu, ref)
    print(f"[parity] {label}: {rep}")
    assert rep["nan"] == 0 and rep["max_8bit"] <= 1 and rep["max_abs"] < 1e-9, rep
    # same arithmetic under both schedules; the persistent one may split the last pixels' ray trees between lanes
    # and add the parts (order of a few f64 additions is timing dependent)
    assert float(np.max(np.abs(gpu - gpu2))) < 1e-12
    rep2 = sf.compare(gpu2, ref)
    assert rep2["nan"] == 0 and rep2["max_8bit"] <= 1 and rep2["max_abs"] < 1e-9, rep2
    assert float(np.abs(gpu).sum()) > 0
    return gpu


def _floor(**kw):
    return rtc.Plane(material=rtc.Material(**kw))


LIGHT = [rtc.PointLight((-6, 8, -8), (1, 1, 1))]


def test_every_pattern_kind():
    P = rtc
    stripes = P.StripePattern(P.SolidPattern((1, 0.2, 0.2)), P.SolidPattern((0.2, 0.2, 1)), M.scaling(0.25, 1, 1).rotate_y(0.4))
    checker = P.CheckerPattern(P.SolidPattern((0.9, 0.9, 0.9)), stripes, M.scaling(0.7, 0.7, 0.7))
    objs = [
        rtc.Plane(material=rtc.Material(pattern=checker, reflectivity=0.2, specular=0.0)),
        rtc.Sphere(M.translation(-2.6, 1, 0), rtc.Material(pattern=P.GradientPattern((1, 0, 0), (0, 0, 1), M.scaling(2, 1, 1).translate(-1, 0, 0)))),
        rtc.Sphere(M.translation(0, 1, 0.5), rtc.Material(pattern=P.RingPattern((0.1, 0.8, 0.1), (1, 1, 0), M.scaling(0.2, 0.2, 0.2).rotate_x(1.0)))),
        rtc.Cube(M.scaling(0.8, 0.8, 0.8).rotate_y(0.6).translate(2.6, 0.8, 0), rtc.Material(pattern=P.RadialGradientPattern((1, 1, 1), (0.1, 0.1, 0.6), M.scaling(0.3, 0.3, 0.3)))),
        rtc.Cylinder(M.scaling(0.5, 1, 0.5).translate(-1.2, 0, -2.2), rtc.Material(pattern=P.BlendedPattern(P.StripePattern(transformation=M.scaling(0.3, 1, 1)), P.StripePattern(transformation=M.scaling(0.3, 1, 1).rotate_y(PI / 2)))), 0.0, 1.5, True),
        rtc.Sphere(M.scaling(0.6, 0.6, 0.6).translate(1.3, 0.6, -2.3), rtc.Material(pattern=P.TestPattern(), ambient=0.4)),
        rtc.Sphere(M.scaling(0.4, 0.4, 0.4).translate(0, 0.4, -3), rtc.Material(pattern=P.SolidPattern((0.3, 0.9, 0.9)))),
    ]
    _check(rtc.World(objs, LIGHT), _cam(), "patterns")


def test_negative_coordinates_in_patterns():
    # Rust's % is fmod: floor(-0.1) % 2 == -1 -> colour b, floor(-1.1) % 2 == -0 -> colour a (pattern.rs:152-162)
    pat = rtc.CheckerPattern(rtc.SolidPattern((1, 1, 1)), rtc.SolidPattern((0, 0, 0)))
    objs = [rtc.Plane(material=rtc.Material(pattern=pat, specular=0)),
            rtc.Plane(M.rotation_x(PI / 2).translate(0, 0, 6), rtc.Material(pattern=rtc.StripePattern(), specular=0))]
    _check(rtc.World(objs, LIGHT), _cam(frm=(0.3, 3, -6), to=(-0.4, 0, 0)), "negative pattern coordinates")


def test_area_light_soft_shadows():
    lights = [rtc.AreaLight((-1, 4, -5), (2, 0, 0), 3, (0, 2, 0), 2, (1.2, 1.1, 1.0)), rtc.PointLight((6, 6, -6), (0.3, 0.3, 0.4))]
    objs = [_floor(color=(1, 0.9, 0.9), specular=0), rtc.Sphere(M.translation(0, 1, 0), rtc.Material(color=(0.3, 0.6, 1), reflectivity=0.3)),
            rtc.Cube(M.scaling(0.5, 0.5, 0.5).translate(1.8, 0.5, -1), rtc.Material(color=(1, 0.5, 0.2)))]
    _check(rtc.World(objs, lights), _cam(), "area light")


def test_cylinders_and_cones_with_and_without_caps():
    objs = [_floor(color=(0.8, 0.8, 0.8), reflectivity=0.1)]
    x = -3.0
    for closed in (False, True):
        for ctor, lo, hi in ((rtc.Cylinder, 0.0, 1.5), (rtc.Cylinder, -0.5, 0.7), (rtc.Cone, -1.0, 0.0), (rtc.Cone, -0.8, 0.6), (rtc.Cone, 0.2, 1.0)):
            t = M.rotation_z(0.3 * x).rotate_x(0.2).scale(0.45, 0.9, 0.45).translate(x, 1.0, 0.4 * math.sin(x))
            objs.append(ctor(t, rtc.Material(color=(0.5 + 0.1 * x, 0.6, 0.5 - 0.1 * x), reflectivity=0.15), lo, hi, closed))
            x += 0.65
    objs.append(rtc.Cylinder(M.scaling(0.2, 1, 0.2).translate(0, 0, 3), rtc.Material(color=(1, 1, 0))))  # infinite
    objs.append(rtc.Cone(M.scaling(0.3, 0.6, 0.3).translate(2.5, 1.5, 3), rtc.Material(color=(0, 1, 1))))  # infinite, open
    _check(rtc.World(objs, LIGHT), _cam(128, 72, (0, 3, -8)), "cylinders and cones")


def _hexagon():
    # ray_tracer/examples/hexagon.rs:8-66: six sides, each a group of a corner sphere and an edge cylinder
    hexa = rtc.Group()
    for n in range(6):
        side = rtc.Group(transformation=M.rotation_y(n * PI / 3))
        side.add_child(rtc.Sphere(M.scaling(0.25, 0.25, 0.25).translate(0, 0, -1)))
        side.add_child(rtc.Cylinder(M.scaling(0.25, 1, 0.25).rotate_z(-PI / 2).rotate_y(-PI / 6).translate(0, 0, -1), None, 0.0, 1.0, False))
        hexa.add_child(side)
    return hexa


def test_nested_groups_of_analytic_shapes():
    hexa = _hexagon()
    hexa.transformation = M.rotation_x(-0.5).scale(1.5, 1.2, 1.5).translate(0, 1.2, 0)
    outer = rtc.Group(transformation=M.rotation_z(0.2).translate(0.3, 0, 0), children=[hexa, rtc.Cube(M.scaling(0.3, 0.3, 0.3).translate(0, 1.2, 0), rtc.Material(color=(1, 0.3, 0.3), reflectivity=0.4))])
    objs = [_floor(color=(0.7, 0.8, 0.7), reflectivity=0.2), outer]
    _check(rtc.World(objs, LIGHT), _cam(), "nested groups")


def test_flat_and_smooth_triangles_outside_and_inside_groups():
    tri = rtc.Triangle((0, 2, 0), (-1.5, 0.2, 0.3), (1.5, 0.2, -0.3), material=rtc.Material(color=(1, 0.8, 0.2), reflectivity=0.3))
    smooth = rtc.SmoothTriangle((2.5, 2, 1), (1.2, 0.1, 1), (3.6, 0.1, 1.2), (0, 1, -0.3), (-0.6, 0.2, -1), (0.6, 0.2, -1))
    g = rtc.Group(transformation=M.rotation_y(0.7).translate(-2.5, 0, 1), material=rtc.Material(color=(0.2, 0.9, 0.4)))
    g.add_child(rtc.Triangle((0, 1.8, 0), (-1, 0, 0), (1, 0, 0)))
    g.add_child(rtc.SmoothTriangle((0, 1.8, 0.5), (-1, 0, 0.5), (1, 0, 0.5), (0, 1, -1), (-1, 0, -1), (1, 0, -1)))
    objs = [_floor(color=(0.9, 0.9, 1.0)), tri, smooth, g]
    _check(rtc.World(objs, LIGHT), _cam(), "triangles")


def test_overlapping_glass_containers():
    # intersection.rs:396-443: A (n 1.5) contains B (2.0) and C (2.5) which overlap each other
    a = rtc.Sphere(M.scaling(2, 2, 2).translate(0, 2, 0), rtc.Material(transparency=0.9, reflectivity=0.3, refractive_index=1.5, diffuse=0.1, color=(1, 0.9, 0.9)))
    b = rtc.Sphere(M.translation(-0.45, 2, -0.25), rtc.Material(transparency=0.8, reflectivity=0.2, refractive_index=2.0, diffuse=0.2, color=(0.9, 1, 0.9)))
    c = rtc.Sphere(M.translation(0.45, 2, 0.25), rtc.Material(transparency=0.7, reflectivity=0.5, refractive_index=2.5, diffuse=0.2, color=(0.9, 0.9, 1)))
    wall = rtc.Plane(M.rotation_x(PI / 2).translate(0, 0, 6), rtc.Material(pattern=rtc.CheckerPattern(rtc.SolidPattern((0.9, 0.2, 0.2)), rtc.SolidPattern((1, 1, 1))), specular=0))
    pool = rtc.Cube(M.scaling(3, 0.4, 3).translate(0, -0.2, 0), rtc.Material(transparency=0.6, refractive_index=1.33, reflectivity=0.4, color=(0.2, 0.4, 0.8), diffuse=0.3))
    objs = [_floor(pattern=rtc.CheckerPattern(), specular=0), wall, a, b, c, pool]
    _check(rtc.World(objs, LIGHT), _cam(112, 72, (0, 3, -8), (0, 1.5, 0)), "overlapping glass")


def test_casts_shadow_false_and_nearest_hit_rule():
    # world.rs:86-87 (Q5): only the NEAREST positive hit decides; a non-casting object in front of a casting one
    # lets the light through
    ghost = rtc.Sphere(M.translation(0, 3, -1.2), rtc.Material(color=(1, 0.2, 0.2), casts_shadow=False))
    solid = rtc.Sphere(M.scaling(0.7, 0.7, 0.7).translate(-1.0, 1.6, 0.4), rtc.Material(color=(0.2, 0.2, 1)))
    glass = rtc.Sphere(M.translation(1.6, 1, 0), rtc.Material(transparency=0.9, refractive_index=1.5, casts_shadow=False, diffuse=0.1, reflectivity=0.9))
    objs = [_floor(color=(1, 1, 1), specular=0), ghost, solid, glass]
    _check(rtc.World(objs, [rtc.PointLight((0.5, 9, -3), (1, 1, 1))]), _cam(), "casts_shadow")


def test_three_lights_multiply_secondary_rays():
    # world.rs:45-68 (Q3): reflected + refracted colour is added once per light
    lights = [rtc.PointLight((-6, 8, -8), (0.5, 0.5, 0.5)), rtc.PointLight((6, 8, -8), (0.4, 0.3, 0.3)), rtc.PointLight((0, 9, 4), (0.2, 0.3, 0.4))]
    objs = [_floor(color=(0.9, 0.9, 0.9), reflectivity=0.5), rtc.Sphere(M.translation(-1.2, 1, 0), rtc.Material(color=(1, 0.3, 0.3), reflectivity=0.6)),
            rtc.Sphere(M.translation(1.2, 1, 0), rtc.Material(transparency=0.9, reflectivity=0.9, refractive_index=1.5, diffuse=0.1, color=(0.8, 1, 0.8)))]
    _check(rtc.World(objs, lights), _cam(), "three lights")


@pytest.mark.parametrize("depth", [0, 1, 3])
def test_bounce_budget(depth):
    world, cam = sf.load("reflective_world", 96, 48)
    _check(world, cam, f"depth {depth}", depth=depth)


def test_portrait_camera_and_no_lights():
    # camera.rs:25-33: aspect < 1; world.rs:42-76 with no lights: everything is black
    objs = [_floor(color=(0.9, 0.9, 0.9), reflectivity=0.3), rtc.Sphere(M.translation(0, 1, 0), rtc.Material(color=(0.2, 1, 0.4)))]
    _check(rtc.World(objs, LIGHT), _cam(40, 90), "portrait")
    scene = _cam(24, 16).compile(rtc.World(objs, []), device=0)
    try:
        img = scene.render()
    finally:
        scene.close()
    assert np.all(img == 0.0)


def _glass_scene(w=96, h=64):
    objs = [_floor(color=(0.9, 0.9, 0.9), reflectivity=0.4),
            rtc.Sphere(M.translation(-1.1, 1, 0), rtc.Material(transparency=0.9, reflectivity=0.9, refractive_index=1.5, diffuse=0.1)),
            rtc.Sphere(M.translation(1.1, 1, 0), rtc.Material(transparency=0.9, reflectivity=0.9, refractive_index=1.5, diffuse=0.1)),
            rtc.Sphere(M.scaling(0.5, 0.5, 0.5).translate(0, 0.5, -1.5), rtc.Material(color=(1, 0.2, 0.2), reflectivity=0.8))]
    return rtc.World(objs, LIGHT), _cam(w, h)


@pytest.mark.parametrize("schedule", [1, 2])
def test_ray_cap_is_reported_without_the_counting_kernel(schedule):
    # a lane that exceeds a per-lane bound used to leave a NaN pixel behind an RTC_OK
    import torch
    world, cam = _glass_scene()
    scene = cam.compile(world, device=0)
    try:
        good = scene.render(schedule=schedule)
        assert not np.isnan(good).any()
        # (without the ray exchange: with it every lane that takes over part of a pixel's ray tree counts from 1 again,
        # and whether some lane still reaches the cap depends on how the tree happened to be split)
        with pytest.raises(rtc._native.RtcError) as e:
            scene.render(schedule=schedule, max_rays_per_pixel=6, exchange=1)
        assert e.value.code == rtc._native.ERR_OVERFLOW
        bad = scene.last_canvas
        assert np.isnan(bad).any()                       # the pixels concerned are NaN ...
        ok = ~np.isnan(bad).any(axis=-1)
        # ... every other pixel is what it was (the persistent schedule may have added a shared pixel's parts in
        # another order)
        assert ok.any() and float(np.max(np.abs(bad[ok] - good[ok]))) < 1e-12
        if schedule == 1:
            assert np.array_equal(bad[ok], good[ok])
        # the asynchronous entry point: ask rtc_scene_check afterwards
        buf = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float64, device="cuda:0")
        scene.render_device(buf.data_ptr(), 0, schedule=schedule, max_rays_per_pixel=6, exchange=1)
        with pytest.raises(rtc._native.RtcError) as e2:
            scene.check()
        assert e2.value.code == rtc._native.ERR_OVERFLOW
        scene.render_device(buf.data_ptr(), 0, schedule=schedule)
        scene.check()  # a clean launch clears the report
        assert np.array_equal(buf.cpu().numpy(), good) or schedule == 2
    finally:
        scene.close()


def test_persistent_schedule_without_exchange_is_bit_reproducible():
    # with rtc_render_opts.exchange = 1 neither the in-wave nor the between-wave hand-over runs, every pixel
    # is traced by one lane, and the persistent frame equals the static one bit for bit -- twice in a row
    for world, cam in (_glass_scene(160, 96), rtc.stress_scene.build(1500, 128, 72, with_teapot=False)):
        scene = cam.compile(world, device=0)
        try:
            static = scene.render(schedule=1, exchange=1)
            a = scene.render(schedule=2, exchange=1)
            b = scene.render(schedule=2, exchange=1)
            c = scene.render(schedule=2)  # exchange on: the same arithmetic, a few additions in another order
            # static schedule, default: tiles that show glass are rendered as quarters whose free lanes take over waiting
            # rays (round 3) -- the same arithmetic again, and in-wave hand-overs follow the wave's lockstep: reproducible
            d, e = scene.render(schedule=1), scene.render(schedule=1)
        finally:
            scene.close()
        assert np.array_equal(static, a) and np.array_equal(a, b)
        assert float(np.max(np.abs(static - c))) < 1e-12
        assert float(np.max(np.abs(static - d))) < 1e-12 and np.array_equal(d, e)


def test_one_process_two_replicas_assemble_the_same_canvas():
    # rtc_scene_commit_devices: the sharded single-process path (interleaved 8-row tiles, peer copies into the primary
    # device's canvas), rehearsed on one GPU by listing it twice (three times: a tile count that does not divide)
    world, cam = _glass_scene(120, 70)
    one = cam.compile(world, device=0)
    try:
        ref = one.render(schedule=1)
    finally:
        one.close()
    for devs in ([0, 0], [0, 0, 0]):
        multi = cam.compile(world, device=devs)
        try:
            assert multi.info()["n_devices"] == len(devs)
            got = multi.render(schedule=1)
            got2 = multi.render()  # auto schedule
            st = None
            multi.render(collect_stats=True)
            st = multi.last_stats
        finally:
            multi.close()
        assert np.array_equal(got, ref)
        assert float(np.max(np.abs(got2 - ref))) < 1e-12
        assert st["primary_rays"] == (cam.hsize - 1) * (cam.vsize - 1)  # counters are summed over the replicas


def test_alternating_shares_on_a_side_stream_keep_their_own_tile_order():
    # rtc_render_device is asynchronous on the caller's stream; every share of a frame brings a tile order of
    # its own, and a launch that is still in flight on a non-blocking stream must keep reading ITS order while the next
    # call (another share) uploads another.  Two shares alternate without a synchronisation in between; every canvas
    # must be the share's own, bit for bit (exchange = 1: reproducible), as the synchronous rtc_render returns it.
    import torch
    world, cam = _glass_scene(512, 384)
    scene = cam.compile(world, device=0)
    try:
        kw = [dict(tile_rows=8, tile_index=r, tile_count=2, exchange=1) for r in range(2)]
        want = [scene.render(**k) for k in kw]
        maps = [scene.local_row_map(tile_rows=8, tile_index=r, tile_count=2) for r in range(2)]
        fresh = cam.compile(world, device=0)  # (its orders have not been built yet: the first launches build them)
        try:
            stream = torch.cuda.Stream(device="cuda:0")
            bufs = [torch.full((len(maps[i % 2]), cam.hsize, 3), float("nan"), dtype=torch.float64, device="cuda:0")
                    for i in range(8)]
            torch.cuda.synchronize()
            for i, buf in enumerate(bufs):
                fresh.render_device(buf.data_ptr(), stream.cuda_stream, **kw[i % 2])
            stream.synchronize()
            fresh.check()
            for i, buf in enumerate(bufs):
                got = buf.cpu().numpy()
                assert np.array_equal(got, want[i % 2][maps[i % 2]]), i
        finally:
            fresh.close()
    finally:
        scene.close()


def test_frames_in_flight_on_two_streams_with_tail_polls():
    # rtc_render_opts.tail_polls: at the end of a frame of the persistent schedule a wave without work looks N times for
    # rays of others and then leaves -- so that the next frame's grid (another scene handle = another work buffer, another stream)
    # moves in while the last ray trees finish.  Six frames alternate between two handles and two streams without a
    # synchronisation; every canvas must be the frame itself (the parts of split pixels are added in another order).
    import torch
    world, cam = rtc.stress_scene.build(20_000, 960, 540, with_teapot=False)
    a, b = cam.compile(world, device=0), cam.compile(world, device=0)
    try:
        want = a.render(schedule=2, exchange=1)
        lone = a.render(schedule=2, tail_polls=8)  # a frame alone: slower, same pixels
        assert float(np.max(np.abs(lone - want))) < 1e-12
        streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
        bufs = [torch.full((cam.vsize, cam.hsize, 3), float("nan"), dtype=torch.float64, device="cuda:0") for _ in range(6)]
        torch.cuda.synchronize()
        for i, buf in enumerate(bufs):
            (a, b)[i % 2].render_device(buf.data_ptr(), streams[i % 2].cuda_stream, schedule=2, tail_polls=8)
        torch.cuda.synchronize()
        a.check()
        b.check()
        for i, buf in enumerate(bufs):
            got = buf.cpu().numpy()
            assert not np.isnan(got).any(), i
            assert float(np.max(np.abs(got - want))) < 1e-12, i
    finally:
        a.close()
        b.close()


def test_two_real_gpus_equal_one():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    world, cam = _glass_scene(200, 120)
    one = cam.compile(world, device=0)
    two = cam.compile(world, device=[0, 1])
    try:
        assert np.array_equal(one.render(schedule=1), two.render(schedule=1))
    finally:
        one.close()
        two.close()


def test_shadow_rays_behind_the_surface_are_counted_not_traced():
    # material.rs:49-52: with the light behind the surface the result of is_shadowed cannot reach the pixel.  traced +
    # skipped must equal the reference's count of shadow rays (the oracle traces every one)
    world, cam = sf.load("reflective_world", 96, 48)
    scene = cam.compile(world, device=0)
    try:
        scene.render(collect_stats=True)
        st = scene.last_stats
    finally:
        scene.close()
    ow = sf.to_oracle(world, cam)
    ow.render()
    assert st["shadow_rays_skipped"] > 0
    # one light: the reference traces one shadow ray per shaded hit and per re-traced subtree; the GPU traces each
    # subtree once (Q3), so with a single light the two counts coincide
    assert st["shadow_rays"] + st["shadow_rays_skipped"] == ow.last_stats["shadow"]


def _csg_scene():
    G = rtc
    red = G.Material(color=(1, 0.3, 0.2), reflectivity=0.2)
    blue = G.Material(color=(0.2, 0.4, 1), specular=0.4)
    glass = G.Material(color=(0.9, 1, 0.9), transparency=0.85, reflectivity=0.5, refractive_index=1.5, diffuse=0.15)
    # difference: a cube with a spherical bite taken out (shape.rs:2195)
    bite = G.CsgDifference(G.Cube(M.scaling(0.8, 0.8, 0.8).rotate_y(0.5).translate(-2.4, 0.8, 0), red),
                           G.Sphere(M.translation(-2.0, 1.5, -0.6), blue))
    # intersection: the lens two spheres share (shape.rs:2194), in glass (containers see both leaves)
    lens = G.CsgIntersection(G.Sphere(M.translation(-0.45, 1, -1), glass), G.Sphere(M.translation(0.45, 1, -1), glass))
    # union of a capped cylinder and a cone, inside a transformed group (shape.rs:2193; chain of group + Csg levels)
    grp = G.Group(transformation=M.rotation_z(0.25).translate(2.3, 0, 0.5))
    grp.add_child(G.CsgUnion(G.Cylinder(M.scaling(0.5, 1, 0.5), red, 0.0, 1.2, True),
                             G.Cone(M.scaling(0.8, 0.8, 0.8).translate(0, 2.0, 0), blue, -1.0, 0.0, True)))
    # nested: (cube - sphere) intersected with a group of two spheres
    inner = G.CsgDifference(G.Cube(M.scaling(0.6, 0.6, 0.6), blue), G.Sphere(M.scaling(0.75, 0.75, 0.75), red))
    pair = G.Group(children=[G.Sphere(M.translation(0.3, 0.2, 0), red), G.Sphere(M.translation(-0.3, -0.2, 0), blue)])
    nested = G.Group(transformation=M.rotation_y(0.6).rotate_x(-0.4).translate(0.2, 0.7, -3.2), children=[G.CsgIntersection(inner, pair)])
    objs = [_floor(color=(0.85, 0.85, 0.8), reflectivity=0.15), bite, lens, grp, nested]
    return rtc.World(objs, LIGHT), _cam(144, 88, (0, 3, -8), (0, 1, 0))


def test_csg_union_intersection_difference_nested():
    # shape.rs:2146-2291 through the C ABI: rtc_scene_begin_csg / end_csg
    world, cam = _csg_scene()
    img = _check(world, cam, "csg")
    scene = cam.compile(world, device=0)
    try:
        a = scene.render(use_bvh=1, schedule=1)
        b = scene.render(use_bvh=2, schedule=1)  # the tree as an entry of the brute-force list
        c = scene.render(depth=2, use_bvh=1)
        d = scene.render(depth=2, use_bvh=2)
    finally:
        scene.close()
    assert np.array_equal(a, b) and np.array_equal(a, img)
    assert np.array_equal(c, d)


def test_many_csg_objects_go_through_the_bvh():
    # a grid of small Csg objects: enough bounded entries for the BVH, each tree one leaf of it
    G = rtc
    objs = [_floor(color=(0.9, 0.9, 0.9), reflectivity=0.1)]
    k = 0
    for i in range(-3, 4):
        for j in range(3):
            t = M.scaling(0.35, 0.35, 0.35).translate(i * 0.9, 0.35, j * 1.1 - 1.0)
            t2 = M.scaling(0.25, 0.25, 0.25).translate(i * 0.9 + 0.15, 0.55, j * 1.1 - 1.15)
            mat = G.Material(color=(0.3 + 0.1 * (k % 7), 0.8 - 0.1 * (k % 5), 0.5), reflectivity=0.2 * (k % 2))
            op = (G.CsgUnion, G.CsgIntersection, G.CsgDifference)[k % 3]
            objs.append(op(G.Cube(t, mat), G.Sphere(t2, mat)))
            k += 1
    _check(rtc.World(objs, LIGHT), _cam(128, 80, (0, 3.5, -7), (0, 0.3, 0)), "csg grid")


def test_edge_cases_empty_world_tiny_cameras_unbounded_only_and_depth_limit():
    # the corners of the boundary: nothing to hit (world.rs:30-38 -> black), cameras one pixel wide / high (Q1: the
    # last row and column are never rendered, camera.rs:72,78), a world of unbounded primitives only (no BVH at
    # all), the deepest bounce budget the ring of waiting rays holds (8) and one beyond it (an error, not a canvas)
    light = LIGHT
    for objs, cam, label in (
            ([], _cam(40, 24), "empty world"),
            ([_floor(color=(0.3, 0.6, 0.9), reflectivity=0.5)], _cam(33, 17), "one plane, odd canvas"),
            ([_floor(color=(1, 1, 1)), rtc.Plane(M.rotation_x(PI / 2).translate(0, 0, 6), rtc.Material(color=(0.8, 0.3, 0.3), reflectivity=0.9)),
              rtc.Cylinder(M.scaling(0.5, 1, 0.5).translate(0, 0, 1), rtc.Material(transparency=0.8, refractive_index=1.3, reflectivity=0.4))],
             _cam(48, 32), "unbounded primitives only"),
            ([rtc.Sphere(M.translation(0, 1, 0))], _cam(1, 9), "one column"),
            ([rtc.Sphere(M.translation(0, 1, 0))], _cam(9, 1), "one row"),
            ([rtc.Sphere(M.translation(0, 1, 0))], _cam(2, 2), "one rendered pixel")):
        scene = cam.compile(rtc.World(objs, light), device=0)
        try:
            for schedule in (1, 2):
                gpu = scene.render(schedule=schedule)
                ref = sf.to_oracle(rtc.World(objs, light), cam).render()
                rep = sf.compare(gpu, ref)
                print(f"[parity] {label} schedule {schedule}: {rep}")
                assert rep["nan"] == 0 and rep["max_8bit"] <= 1 and rep["max_abs"] < 1e-9, (label, rep)
                if not objs or min(cam.hsize, cam.vsize) == 1:
                    assert np.all(gpu == 0.0), label
        finally:
            scene.close()
    world, cam = _glass_scene(64, 40)
    _check(world, cam, "depth 8", depth=8)
    scene = cam.compile(world, device=0)
    try:
        with pytest.raises(rtc.RtcError):
            scene.render(depth=9)
        assert not np.isnan(scene.render(depth=0)).any()
    finally:
        scene.close()


def test_large_canvas_download_is_exact_whole_and_tiled():
    # rtc_render moves a 4K canvas (199 MB) to the caller through three pinned staging buffers with several host
    # threads copying out of them: every byte must land where rtc_render_device put it, for the whole frame and for
    # an interleaved row-tile share (whose rows are scattered into the full-size canvas)
    import torch
    w, h = 3840, 2160
    objs = [_floor(color=(0.9, 0.8, 0.7), reflectivity=0.2),
            rtc.Sphere(M.translation(0, 1, 0), rtc.Material(color=(0.2, 0.9, 0.4), reflectivity=0.3)),
            rtc.Cube(M.scaling(0.6, 0.6, 0.6).rotate_y(0.5).translate(2.2, 0.6, 0.5), rtc.Material(color=(0.9, 0.3, 0.2)))]
    scene = _cam(w, h).compile(rtc.World(objs, LIGHT), device=0)
    try:
        dev = torch.empty((h, w, 3), dtype=torch.float64, device="cuda:0")
        scene.render_device(dev.data_ptr(), 0, schedule=1)
        scene.check()
        want = dev.cpu().numpy()
        got = scene.render(schedule=1)
        assert np.array_equal(got, want) and float(np.abs(got).sum()) > 0
        tile = dict(tile_rows=8, tile_index=1, tile_count=3)
        rows = scene.local_row_map(**tile)
        part = scene.render(out=np.full((h, w, 3), -1.0), schedule=1, **tile)
        assert np.array_equal(part[rows], want[rows])
        others = np.setdiff1d(np.arange(h), rows)
        assert np.all(part[others] == 0.0)  # rows of the other shares come back black (rtc_render clears the canvas)
    finally:
        scene.close()


def test_wavefront_schedule_matches_oracle_and_megakernel():
    # schedule = 3 (experimental): rays in device queues, one traversal kernel (the megakernel's pass code) and one
    # shading kernel per kind of ray and generation.  Same arithmetic per stage, contributions added with f64 atomics.
    import os
    scenes = []
    for name, w, h in (("cover", 96, 96), ("reflective_world", 120, 60), ("teapot", 64, 32)):
        sc = rtc.parse_scene_file(os.path.join(sf.ROOT, "scenes", name + ".yaml"), w, h)
        scenes.append((name, rtc.World(sc.objects, sc.lights), sc.camera))
    scenes.append(("glass",) + _glass_scene(96, 64))
    stripes = rtc.StripePattern(rtc.SolidPattern((1, 0.2, 0.2)), rtc.SolidPattern((0.2, 0.2, 1)), M.scaling(0.25, 1, 1))
    area = rtc.AreaLight((-3, 6, -6), (2, 0, 0), 3, (0, 2, 0), 3, (1, 1, 1))
    scenes.append(("patterns + area light + group",
                   rtc.World([_floor(pattern=stripes, reflectivity=0.2),
                              rtc.Group(transformation=M.translation(0, 1, 0), material=None, children=[rtc.Sphere(M.scaling(0.8, 0.8, 0.8), rtc.Material(transparency=0.7, refractive_index=1.4, reflectivity=0.3)),
                                                                 rtc.Cube(M.scaling(0.4, 0.4, 0.4).translate(1.6, 0, 0))])],
                             [area, rtc.PointLight((4, 5, -3), (0.4, 0.4, 0.4))]), _cam(80, 48)))
    for label, world, cam in scenes:
        scene = cam.compile(world, device=0)
        try:
            mega = scene.render(schedule=1)
            wave = scene.render(schedule=3)
            part = scene.render(schedule=3, tile_rows=8, tile_index=1, tile_count=3)
            rows = scene.local_row_map(tile_rows=8, tile_index=1, tile_count=3)
        finally:
            scene.close()
        ref = sf.to_oracle(world, cam).render()
        rep = sf.compare(wave, ref)
        print(f"[parity] wavefront {label}: {rep}")
        assert rep["nan"] == 0 and rep["max_8bit"] <= 1 and rep["max_abs"] < 1e-9, (label, rep)
        assert float(np.max(np.abs(wave - mega))) < 1e-12, label
        assert float(np.max(np.abs(part[rows] - mega[rows]))) < 1e-12, label
