#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path on MI355X.

Metric (BASELINE.json): Mrays/s (primary + shadow + secondary) and frame time at 4K on 1/2/4/8 GPUs.
A "step" is one full frame: every active pixel of the workload goes through the hot path once
(Camera::render).  Workload at every N: BASELINE.json configs[4], the synthetic stress scene (100k glass/mirror
spheres + the teapot mesh, 3840x2160, depth 5) -- the configuration the metric's "at 4K, 1/2/4/8 GPUs" is quoted
on; it fits one GPU.  The reference sample scenes (configs[1..3]: cover, ice_cream / reflective_world, teapot at
4K) are timed beside it at N = 1 and reported under "extra_workloads".  With N > 1 the SAME frame is sharded by
interleaved 8-row tiles over the ranks (strong scaling) and the canvas is assembled on rank 0 with an RCCL
gather, as BASELINE.json's north_star prescribes.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Inputs (scene tables) are resident in HBM before the timed region; the timed region is K x (render kernel
[+ gather + de-interleave on rank 0]) bracketed by barrier + synchronize; the value uses the MAX over ranks.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene file, width, height)
    "teapot_4k": ("teapot", 3840, 2160),
    "stress_4k": ("stress", 3840, 2160),  # BASELINE.json configs[4]: 100k glass/mirror spheres + teapot
    "stress_1080p": ("stress", 1920, 1080),  # (diagnostic: how much of the 4K frame time is its tail)
    "stress_8k": ("stress", 7680, 4320),
    "cover_4k": ("cover", 3840, 2160),
    "cover_1080p": ("cover", 1920, 1080),
    "reflective_world_1080p": ("reflective_world", 1920, 1080),
    "ice_cream_1080p": ("ice_cream", 1920, 1080),
}
# oracle pixels per extra workload (seeded-random; ~2 s of oracle time each on 16 host threads)
PARITY_PIXELS = {"cover_4k": 100_000, "cover_1080p": 100_000, "teapot_4k": 40_000, "reflective_world_1080p": 300_000,
                 "ice_cream_1080p": 300_000, "stress_1080p": 32, "stress_8k": 32, "stress_4k": 32}
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop (FMA) x 2.4 GHz; strict mode cannot use FMA
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector peak (the slab tests' packed FMA)
TILE_ROWS = 8


def compare(a, b):
    """two canvases: max |delta| in f64 and the 8-bit channels (Canvas::to_ppm's scale_and_clamp, canvas.rs:86-89)"""
    import numpy as np

    def to8(img):
        v = np.asarray(img, dtype=np.float64) * 255.0
        r = np.where(np.isnan(v), 0.0, np.sign(v) * np.floor(np.abs(v) + 0.5))
        return np.clip(r, 0, 255).astype(np.int32)
    d = np.abs(np.asarray(a) - np.asarray(b))
    d8 = np.abs(to8(a) - to8(b))
    return dict(max_abs=float(np.nanmax(d)) if d.size else 0.0, nan=int(np.isnan(d).sum()),
                max_8bit=int(d8.max()) if d8.size else 0, mismatched_channels=int((d8 > 0).sum()),
                pixels_over_1=int((d8.max(axis=-1) > 1).sum()) if d8.size else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="stress_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a one-GPU box: every rank renders on cuda:0 and the gather runs over gloo "
                         "through host memory (RCCL refuses two ranks on one device); numbers are not comparable")
    ap.add_argument("--cpu-sample-stride", type=int, default=0, help="0 = choose for ~10-20 s of CPU work")
    ap.add_argument("--use-bvh", type=int, default=0, help="0 auto, 1 force BVH, 2 force brute force (A/B only)")
    ap.add_argument("--schedule", type=int, default=0, help="0 auto, 1 workgroup per tile, 2 persistent (A/B only)")
    ap.add_argument("--extra-stress", action="store_true", help="include the (slow) stress scene in --extra")
    ap.add_argument("--extra", action="store_true", help="(default at N=1) also time the reference sample scenes")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the frames-in-flight measurement")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-flavours", action="store_true", help="(accepted and ignored: the contracted flavour is gone)")
    ap.add_argument("--parity-pixels", type=int, default=0,
                    help="oracle pixels per extra workload (0 = sized for ~2 s of oracle time each)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rtc = importlib.import_module("ray-tracer-challenge_amd")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    distributed = world_size > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert world_size == n_gpus, f"--gpus {n_gpus} but WORLD_SIZE={world_size}"
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def load(workload):
        name, w, h = WORKLOADS[workload]
        if name == "stress":
            return rtc.stress_scene.build(100_000, w, h)
        sc = rtc.parse_scene_file(os.path.join(ROOT, "scenes", name + ".yaml"), w, h)
        return rtc.World(sc.objects, sc.lights), sc.camera

    world, cam = load(args.workload)
    scene = cam.compile(world, device=local_rank)
    hsize, vsize = cam.hsize, cam.vsize
    tile = dict(tile_rows=TILE_ROWS, tile_index=rank, tile_count=world_size) if distributed else {}
    ropts = dict(tile)
    if args.schedule:
        ropts["schedule"] = args.schedule
    if args.use_bvh:
        ropts["use_bvh"] = args.use_bvh
    local_rows = scene.local_rows(**tile) if distributed else vsize
    max_rows = local_rows
    if distributed:
        t = torch.tensor([local_rows], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        max_rows = int(t.item())
    buf = torch.zeros((max_rows, hsize, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)

    gather_list = None
    canvas_ext = None
    index = None
    if distributed and rank == 0:
        # one preallocated buffer: rank r's rows land in gathered[r], and the de-interleave reads it in place
        gathered = torch.empty((world_size,) + tuple(buf.shape), dtype=torch.float64, device=dev)
        gather_list = [gathered[r] for r in range(world_size)]
        canvas_ext = torch.zeros((vsize + 1, hsize, 3), dtype=torch.float64, device=dev)  # +1: dump row for padding
        idx = np.full((world_size, max_rows), vsize, dtype=np.int64)
        for r in range(world_size):
            m = scene.local_row_map(tile_rows=TILE_ROWS, tile_index=r, tile_count=world_size)
            idx[r, : len(m)] = m
        index = torch.from_numpy(idx.reshape(-1)).to(dev)

    def exchange():
        """the one exchange step of the path: every rank's row tiles -> rank 0 -> canvas (camera.rs:87-91)"""
        if args.rehearse_on_one_gpu:
            host = buf.cpu()
            parts = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                canvas_ext.index_copy_(0, index, torch.cat(parts, dim=0).to(dev))
            return
        dist.gather(buf, gather_list, dst=0)
        if rank == 0:
            canvas_ext.index_copy_(0, index, gathered.view(-1, hsize, 3))

    def step():
        scene.render_device(buf.data_ptr(), stream.cuda_stream, **ropts)
        if distributed:
            exchange()

    def sync_all():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- counters of one frame (counting kernel variant, outside the timed region) ----
    scene.render_device(buf.data_ptr(), stream.cuda_stream, collect_stats=True, **ropts)
    st = scene.stats()
    keys = ["primary_rays", "shadow_rays", "secondary_rays", "secondary_rays_reference", "box_tests",
            "shaded_hits", "light_evals", "container_passes", "xform_levels", "overflow", "shadow_rays_skipped"]
    vec = [float(st[k]) for k in keys] + [float(x) for x in st["prim_tests"]] + [st["algorithmic_flops"]]
    if distributed:
        t = torch.tensor(vec, device=dev, dtype=torch.float64)
        dist.all_reduce(t)
        vec = t.tolist()
    frame = dict(zip(keys, vec[: len(keys)]))
    frame_prim_tests = vec[len(keys): len(keys) + 7]
    frame_flops = vec[-1]
    local_flops = st["algorithmic_flops"]
    rays_per_frame = frame["primary_rays"] + frame["shadow_rays"] + frame["secondary_rays"]
    assert frame["overflow"] == 0, "kernel iteration cap hit"

    # ---- timed region ----
    for _ in range(args.warmup):
        step()
    sync_all()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        scene.render_device(buf.data_ptr(), stream.cuda_stream, **ropts)
        ev[i][1].record(stream)
        if distributed:
            exchange()
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    if distributed:
        t = torch.tensor([elapsed, kernel_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max = t.tolist()
    else:
        kernel_ms_max = kernel_ms

    if distributed and args.rehearse_on_one_gpu and rank == 0:
        # the assembled canvas must equal a one-rank render of the whole frame (bit for bit except where the persistent
        # schedule split a pixel's ray tree between lanes and added the parts: a few f64 additions in another order)
        full = torch.zeros((vsize, hsize, 3), dtype=torch.float64, device=dev)
        kw = {k: v for k, v in ropts.items() if not k.startswith("tile_")}
        scene.render_device(full.data_ptr(), stream.cuda_stream, **kw)
        torch.cuda.synchronize(dev)
        assert float((full - canvas_ext[:vsize]).abs().max()) < 1e-12, "gathered canvas differs from the single-rank frame"
        print("[rehearsal] gathered canvas == single-rank frame", file=sys.stderr)
    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = rays_per_frame * args.steps / elapsed / 1e6
    # Roofline, in two parts: the reference's own arithmetic runs in f64 on the vector ALU; the BVH box tests
    # are this build's culling arithmetic and run in f32 (packed FMA), so they are priced against the f32 vector peak.
    box_flops_local = 12.0 * float(st["box_tests"])
    fp64_flops_local = local_flops - box_flops_local
    secs = kernel_ms * 1e-3
    fp64_tflops = fp64_flops_local / secs / 1e12 if secs > 0 else 0.0
    f32_tflops = box_flops_local / secs / 1e12 if secs > 0 else 0.0
    achieved_tflops = local_flops / secs / 1e12 if secs > 0 else 0.0
    # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc, separate runs): read from
    # that file, NOT measured in this run
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(args.workload if world_size == 1 else "", {})
            traffic = rec.get("bytes_per_launch")
            if traffic is not None:
                traffic_source = "%s (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, separate passes, of tree %s); read from " \
                                 "profiles/hbm_traffic.json, not measured in this run" % (rec.get("source"), rec.get("commit", "?"))
        except Exception:
            traffic = None
    out = {
        "metric": "Mrays/s (primary+shadow+secondary), %s @%dx%d" % (WORKLOADS[args.workload][0], hsize, vsize),
        "value": value,
        "unit": "Mrays/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "frame_ms": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": ("synthetic stress scene (stress_scene.py, seed 0x5EED1234)" if WORKLOADS[args.workload][0] == "stress" else
                 "reference sample scene (scenes/%s.yaml, camera resized)" % WORKLOADS[args.workload][0]),
        "config": {
            "workload": "%s: %s at %dx%d, depth 5, %d active pixels" % (
                args.workload, "stress_scene.py (100 000 glass / mirror spheres + teapot mesh)" if WORKLOADS[args.workload][0] == "stress"
                else "scenes/%s.yaml" % WORKLOADS[args.workload][0], hsize, vsize, (hsize - 1) * (vsize - 1)),
            "sharding": "interleaved %d-row tiles over %d rank(s); RCCL gather to rank 0" % (TILE_ROWS, world_size)
            if distributed else "single GPU, whole frame in one launch",
            "fp_mode": "strict (no FMA contraction: the reference's arithmetic; there is no other flavour)",
        },
        "rays_per_frame": {
            "primary": frame["primary_rays"], "shadow": frame["shadow_rays"], "secondary": frame["secondary_rays"],
            "total_unique": rays_per_frame,
            # is_shadowed queries whose answer cannot reach the pixel (light behind the surface, material.rs:49-52):
            # the reference traces them, this kernel does not, and `value` does not count them
            "shadow_skipped_light_behind_surface": frame["shadow_rays_skipped"],
            "secondary_reference_equivalent": frame["secondary_rays_reference"],
        },
        "work_per_frame": {"box_tests": frame["box_tests"], "prim_tests": frame_prim_tests,
                           "shaded_hits": frame["shaded_hits"], "container_passes": frame["container_passes"],
                           "algorithmic_flops": frame_flops},
        "roofline": {
            # neither hbm nor mfma bounds this path: the scene is cache-resident, nothing is a GEMM.  The headline pair
            # (achieved / peak / frac) is the reference's f64 arithmetic against the f64 vector peak; the box tests are
            # reported beside it against the f32 peak, and `frac_time_at_peaks` says what share of the kernel's time the
            # two parts would take together if each ran at its own peak.
            "bound": "valu_fp64",
            "kernel": "rtc::k_render",
            "achieved": fp64_tflops,
            "peak": FP64_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": fp64_tflops / FP64_VALU_PEAK_TFLOPS,
            "frac_of_strict_ceiling": fp64_tflops / (FP64_VALU_PEAK_TFLOPS / 2),
            "fp64_reference_flops": fp64_flops_local,
            "f32_culling": {"flops": box_flops_local, "achieved": f32_tflops, "peak": FP32_VALU_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": f32_tflops / FP32_VALU_PEAK_TFLOPS,
                            "what": "BVH box slab tests (12 flops each, SURVEY 8(d)): the build's own arithmetic, f32"},
            "frac_time_at_peaks": (fp64_flops_local / (FP64_VALU_PEAK_TFLOPS * 1e12) +
                                   box_flops_local / (FP32_VALU_PEAK_TFLOPS * 1e12)) / secs if secs > 0 else 0.0,
            # (rounds 1-2 reported every algorithmic flop against the f64 peak: kept for comparison across rounds)
            "frac_all_flops_at_fp64_peak": achieved_tflops / FP64_VALU_PEAK_TFLOPS,
            "kernel_ms": kernel_ms,
            "kernel_ms_max_over_ranks": kernel_ms_max,
            "algorithmic_flops_per_launch": local_flops,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "hbm_algorithmic_bytes_per_launch": 24.0 * hsize * local_rows,
            "hbm": {"achieved": 24.0 * hsize * local_rows / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                    "peak": 8000.0, "unit": "GB/s"},
        },
    }

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    threads = min(os.cpu_count() or 1, 16)
    if not args.no_extra and not distributed:
        import scene_feed as sf
        extra = {}
        for wl in WORKLOADS:
            if wl == args.workload or (wl.startswith("stress") and not args.extra_stress):
                continue
            w2, c2 = load(wl)
            s2 = c2.compile(w2, device=local_rank)
            b2 = torch.zeros((c2.vsize, c2.hsize, 3), dtype=torch.float64, device=dev)
            kw2 = {"use_bvh": args.use_bvh} if args.use_bvh else {}
            if args.schedule:
                kw2["schedule"] = args.schedule
            s2.render_device(b2.data_ptr(), stream.cuda_stream, collect_stats=True, **kw2)
            st2 = s2.stats()
            for _ in range(2):
                s2.render_device(b2.data_ptr(), stream.cuda_stream, **kw2)
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            a.record(stream)
            for _ in range(reps):
                s2.render_device(b2.data_ptr(), stream.cuda_stream, **kw2)
            b.record(stream)
            torch.cuda.synchronize(dev)
            ms = a.elapsed_time(b) / reps
            rays = st2["primary_rays"] + st2["shadow_rays"] + st2["secondary_rays"]
            box2 = 12.0 * float(st2["box_tests"])
            extra[wl] = {"frame_ms": ms, "Mrays_per_s": rays / ms / 1e3, "rays": rays,
                         "tflops": st2["algorithmic_flops"] / ms / 1e9,
                         "roofline_frac": (st2["algorithmic_flops"] - box2) / ms / 1e9 / FP64_VALU_PEAK_TFLOPS,
                         "f32_culling_frac": box2 / ms / 1e9 / FP32_VALU_PEAK_TFLOPS,
                         "frac_all_flops_at_fp64_peak": st2["algorithmic_flops"] / ms / 1e9 / FP64_VALU_PEAK_TFLOPS}
            # the reference-shaped call (Camera::render -> Canvas): kernel + device-to-host copy of the canvas into the
            # caller's pageable buffer, end to end, second call (the first one allocates the kept buffers)
            host = np.empty((c2.vsize, c2.hsize, 3), dtype=np.float64)
            s2.render(out=host, **kw2)
            t1 = time.perf_counter()
            s2.render(out=host, **kw2)
            extra[wl]["render_call_ms"] = (time.perf_counter() - t1) * 1e3
            # parity of the frame just timed: seeded-random pixels against the oracle (brute force -- a sample sized for
            # a couple of seconds of CPU time; `host` holds the canvas of the second rtc_render call above)
            ow2 = sf.to_oracle(w2, c2)
            rng = np.random.default_rng(0xBE7C4 + c2.hsize)
            n_pix = args.parity_pixels or PARITY_PIXELS.get(wl, 20_000)
            xy = np.stack([rng.integers(0, c2.hsize - 1, n_pix), rng.integers(0, c2.vsize - 1, n_pix)], axis=1)
            ref = ow2.render_pixels(xy, threads=threads)
            dev_canvas = b2.cpu().numpy()
            rep = compare(dev_canvas[xy[:, 1], xy[:, 0]], ref)
            extra[wl]["parity"] = {"against": "C oracle, %d seeded-random pixels of the timed frame" % n_pix,
                                   "pixels": n_pix, "max_abs": rep["max_abs"], "mismatched_8bit": rep["mismatched_channels"],
                                   "pixels_over_1_of_255": rep["pixels_over_1"], "nan": rep["nan"],
                                   "oracle_seconds": ow2.last_stats["seconds"],
                                   "render_call_equals_device_frame": bool(np.array_equal(host, dev_canvas)),
                                   "within_tolerance": bool(rep["pixels_over_1"] == 0 and rep["nan"] == 0)}
            s2.close()
        out["extra_workloads"] = extra

    if not args.no_pipelined and not distributed and WORKLOADS[args.workload][0] == "stress" and not args.schedule:
        # Frames in flight (reported BESIDE the headline, never as `value`): the same frame launched alternately through
        # two scene handles (each its own work buffer) on two streams with rtc_render_opts.tail_polls = 32 -- at the end
        # of a frame a wave without work looks 32 times for rays of others and leaves, the next frame's grid moves in
        # while the last ray trees finish, so the end of a frame (a chain of dependent rays, ~43 ms whatever the pixel count) no longer
        # costs throughput.  `value` above stays one frame at a time: that is what `roofline` and profiles/ describe.
        scene_b = cam.compile(world, device=local_rank)
        pbufs = [torch.zeros((vsize, hsize, 3), dtype=torch.float64, device=dev) for _ in range(2)]
        pstreams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        pk = dict(ropts, tail_polls=32)
        handles = [scene, scene_b]
        frames = 2 * max(2, min(args.steps, 8) // 2)

        def in_flight(n):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for k in range(n):
                handles[k % 2].render_device(pbufs[k % 2].data_ptr(), pstreams[k % 2].cuda_stream, **pk)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t1) * 1e3 / n

        in_flight(2)
        pms = in_flight(frames)
        scene.check()
        scene_b.check()
        out["frames_in_flight"] = {
            "frames_in_flight": 2, "tail_polls": 32, "frames": frames, "ms_per_frame": pms,
            "Mrays_per_s": rays_per_frame / pms / 1e3,
            "max_abs_vs_timed_frame": max(float((pb - buf).abs().max()) for pb in pbufs),
            "what": "throughput of a stream of frames, two in flight on two HIP streams (two scene handles); not the "
                    "headline: `value`, `ms_per_step` and `roofline` are one frame at a time"}
        scene_b.close()

    if not args.no_cpu_baseline and not distributed:
        # The reference is Rust and can not be built here (no toolchain): the baseline is the C oracle (brute force,
        # no BVH, the reference's recursion), pixels over threads like the reference's rayon loop over rows, on a
        # time-bounded sample of the SAME frame: batches of seeded-random pixels until ~10 s of CPU time are spent.
        # The same pixels are the parity check of the timed frame; the stress scene adds pixels picked inside its
        # nearest glass spheres (the deep ray trees the end-of-frame exchange splits) to the check, not to the baseline.
        import scene_feed as sf
        gpu_canvas = buf.cpu().numpy()  # the frame of the timed region
        ow = sf.to_oracle(world, cam)
        heavy = WORKLOADS[args.workload][0] == "stress"  # ~3 ms of CPU work per ray (106k primitives per ray)
        batch = 32 if heavy else 20_000
        if args.cpu_sample_stride > 0:
            batch = args.cpu_sample_stride
        rng = np.random.default_rng(0xBA5E11)
        max_abs, mism8, over1, npix, nan = 0.0, 0, 0, 0, 0
        cpu_rays, cpu_secs, cpu_pixels = 0, 0.0, 0

        def check(xy):
            nonlocal max_abs, mism8, over1, npix, nan
            ref = ow.render_pixels(xy, threads=threads)
            # north_star: "a reported per-pixel max-abs float delta" -- the oracle's pixels against the GPU frame
            rep = compare(gpu_canvas[xy[:, 1], xy[:, 0]], ref)
            max_abs, nan = max(max_abs, rep["max_abs"]), nan + rep["nan"]
            mism8, over1 = mism8 + rep["mismatched_channels"], over1 + rep["pixels_over_1"]
            npix += len(xy)
            return ow.last_stats

        while cpu_secs < 10.0 and cpu_pixels < (hsize - 1) * (vsize - 1):
            xy = np.stack([rng.integers(0, hsize - 1, batch), rng.integers(0, vsize - 1, batch)], axis=1)
            cs = check(xy)
            cpu_rays += cs["primary"] + cs["shadow"] + cs["secondary"]
            cpu_pixels += cs["primary"]
            cpu_secs += cs["seconds"]
            if not heavy and cs["seconds"] < 1.0:
                batch *= 4
        glass = sf.glass_pixels(world, cam, 12) if heavy else np.zeros((0, 2), dtype=np.int64)
        if len(glass):
            check(glass)
        out["parity"] = {"against": "C oracle (oracle/rtc_oracle.c), pixels of the timed frame: %d seeded-random%s" % (
                             cpu_pixels, " + %d at the centres of the nearest glass spheres" % len(glass) if len(glass) else ""),
                         "pixels": npix, "max_abs": max_abs, "mismatched_8bit": mism8, "pixels_over_1_of_255": over1,
                         "nan": nan, "within_tolerance": bool(over1 == 0 and nan == 0)}
        out["cpu_baseline"] = {
            "value": cpu_rays / max(cpu_secs, 1e-9) / 1e6,
            "unit": "Mrays/s",
            "cores": threads,
            "kind": "port",
            "sample": "SAMPLED: C oracle (brute force, no BVH, reference recursion, reference-equivalent ray count) on %d "
                      "seeded-random pixels of the same %dx%d frame: %d rays in %.1f s; not a whole-frame time" % (
                          cpu_pixels, hsize, vsize, cpu_rays, cpu_secs),
        }
    print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
