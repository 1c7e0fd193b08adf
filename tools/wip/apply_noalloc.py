#!/usr/bin/env python3
"""Applies the allocation-free entry point (rtc_render_device_static) to the tree.  Kept as a script until its GPU
test has passed, so that it lands as one commit of its own."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def patch(path, pairs):
    p = os.path.join(ROOT, path)
    s = open(p).read()
    for a, b in pairs:
        assert s.count(a) == 1, (path, a[:60], s.count(a))
        s = s.replace(a, b)
    open(p, "w").write(s)


patch("include/rtc.h", [('''/* Number of canvas rows owned by (tile_index, tile_count, tile_rows) and their canvas row numbers. */''',
'''/* rtc_render_device for callers that cannot allocate and cannot rely on memory the library allocated for this
 * scene's launches: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has NO
 * counterpart in the reference.  Same arguments, same pixels.  The kernel's work buffer (tile counters, overflow flag,
 * mailboxes of the end-of-frame ray exchange) is a file-scope `static __device__` array of the library's code object
 * instead of the per-scene buffer that rtc_scene_commit allocated: RTC_STATIC_SCRATCH_BYTES per GPU, held until the
 * process ends.  Limits: collect_stats is refused; a persistent launch whose grid would exceed
 * RTC_STATIC_SCRATCH_WAVES resident waves is refused with RTC_ERR_INVALID (rtc_render_device switches the ray exchange
 * off instead).  Every scene on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race -- serialise them.  The call itself makes no hipMalloc / hipFree / allocating runtime call. */
#define RTC_STATIC_SCRATCH_WAVES 4096u
#define RTC_STATIC_SCRATCH_BYTES (36864u + RTC_STATIC_SCRATCH_WAVES * 64u * 80u)
int rtc_render_device_static(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream);
/* Number of canvas rows owned by (tile_index, tile_count, tile_rows) and their canvas row numbers. */''')])

patch("ray-tracer-challenge_amd/csrc/rtc_scene.h", [('''#define RTC_MAX_DEVICES 64
''', '''#define RTC_MAX_DEVICES 64
// rtc_kernels.hip: the file-scope static device array behind rtc_render_device_static, resolved once per device
int static_scratch(int device, unsigned **out);
''')])

patch("ray-tracer-challenge_amd/csrc/rtc_scene.cpp", [('''    rep.ev_start = a, rep.ev_stop = b;
''', '''    rep.ev_start = a, rep.ev_stop = b;
    // the static scratch array of rtc_render_device_static is resolved here, once per device, with the device current
    unsigned *unused = nullptr;
    if (int rcs = static_scratch(dev, &unused)) return rcs;
''')])

patch("ray-tracer-challenge_amd/csrc/rtc_api.cpp", [('''int rtc_render_device(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream) {''',
'''static int render_device_common(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream,
                                bool static_scratch_only) {'''), ('''    return launch_on(s, r, p, o, d_rgb, work, (hipStream_t)hip_stream);
}
''', '''    // (the static scratch holds mailboxes for RTC_MAX_WAVES resident waves: the static variant refuses a larger grid
    // with a clear error where rtc_render_device switches the ray exchange off)
    return launch_on(s, r, p, o, d_rgb, work, (hipStream_t)hip_stream, static_scratch_only);
}

int rtc_render_device(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream) {
    return render_device_common(scene, opts, d_rgb, hip_stream, false);
}
'''), ('''    DeviceReplica &r = s.reps[0];
    unsigned *work = r.d_work;
''', '''    DeviceReplica &r = s.reps[0];
    unsigned *work = r.d_work;
    if (static_scratch_only) {
        if (o.collect_stats) {
            set_error("rtc_render_device_static does not run the counting kernel (collect_stats)");
            return RTC_ERR_INVALID;
        }
        if ((rc = static_scratch(r.device, &work))) return rc;
    }
'''), ('''int rtc_scene_check(rtc_scene *scene) {''', '''int rtc_render_device_static(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream) {
    return render_device_common(scene, opts, d_rgb, hip_stream, true);
}

int rtc_scene_check(rtc_scene *scene) {''')])

patch("ray-tracer-challenge_amd/csrc/rtc_kernels.hip", [('''template <class K>
static unsigned resident_waves_per_cu(K kernel) {''', '''// ---- rtc_render_device_static: the work buffer as a static device array ------------------------------------------
// One copy per GPU, loaded with the code object and held until the process ends.  `scratch` is a host-side stand-in
// for the symbol, not a device address: host code only ever uses the per-device pointer that hipGetSymbolAddress
// returned (kernel argument, hipMemsetAsync).  Contents survive from call to call; what must start at zero (tile
// counters, mailbox states) is cleared by launch_render's hipMemsetAsync before every persistent launch, and the
// overflow flag is compared with a per-launch id, so stale values never match.
static __device__ __align__(256) unsigned char scratch[RTC_WORK_TOTAL_BYTES];
static_assert(RTC_WORK_TOTAL_BYTES == RTC_STATIC_SCRATCH_BYTES, "include/rtc.h states the size of the static scratch");
static_assert(RTC_MAX_WAVES == RTC_STATIC_SCRATCH_WAVES, "include/rtc.h states the wave limit of the static scratch");
static std::mutex g_scratch_mu;
static unsigned *g_scratch_ptr[RTC_MAX_DEVICES];

int static_scratch(int device, unsigned **out) {
    if (device < 0 || device >= RTC_MAX_DEVICES) {
        set_error("device %d: the static scratch is kept for devices 0..%d", device, RTC_MAX_DEVICES - 1);
        return RTC_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        DeviceGuard guard(device);  // each GPU holds its own copy of the array at its own address
        void *p = nullptr;
        const hipError_t e = guard.ok ? hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) : hipErrorInvalidDevice;
        if (e != hipSuccess || !p) {
            set_error("hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
            return RTC_ERR_HIP;
        }
        g_scratch_ptr[device] = (unsigned *)p;
    }
    *out = g_scratch_ptr[device];
    return RTC_OK;
}

template <class K>
static unsigned resident_waves_per_cu(K kernel) {'''), ('''#include <atomic>
''', '''#include <atomic>
#include <mutex>
''')])

patch("ray-tracer-challenge_amd/_native.py", [('''    ("rtc_local_rows", C.c_uint32, [C.c_uint32] * 4),''',
'''    ("rtc_render_device_static", C.c_int, [C.c_void_p, C.POINTER(RenderOpts), C.c_void_p, C.c_void_p]),
    ("rtc_local_rows", C.c_uint32, [C.c_uint32] * 4),''')])

patch("ray-tracer-challenge_amd/api.py", [('''    def stats(self):
        st = N.Stats()''', '''    def render_device_static(self, device_ptr: int, stream: int = 0, **kw):
        """rtc_render_device_static: the same launch with the library's static device array as its work buffer."""
        o = self._opts(**kw)
        N.check(self.L.rtc_render_device_static(self.h, C.byref(o), C.c_void_p(device_ptr), C.c_void_p(stream)))

    def stats(self):
        st = N.Stats()''')])
os.rename(os.path.join(ROOT, "tests", "wip_no_alloc.py"), os.path.join(ROOT, "tests", "test_zz_no_alloc.py")) if False else None
print("applied")
