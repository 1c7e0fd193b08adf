// rtc_api.cpp -- the extern "C" boundary declared in include/rtc.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <atomic>
#include <limits>
#include <vector>

#include "rtc_scene.h"

using namespace rtc;

namespace rtc {
const char *last_error();
}

struct rtc_scene {
    Scene s;
};

extern "C" {

const char *rtc_last_error(void) { return rtc::last_error(); }
const char *rtc_version(void) { return "rtc_amd 0.2 (gfx950; fp64, strict and fast flavours)"; }

void rtc_material_default(rtc_material *m) {
    // material.rs:83-96
    m->color[0] = m->color[1] = m->color[2] = 1.0;
    m->ambient = 0.1, m->diffuse = 0.9, m->specular = 0.9, m->shininess = 200.0;
    m->reflectivity = 0.0, m->transparency = 0.0, m->refractive_index = 1.0;
    m->casts_shadow = 1;
    m->pattern = -1;
}

void rtc_render_opts_default(rtc_render_opts *o) {
    std::memset(o, 0, sizeof *o);
    o->depth = 5;  // camera.rs:80
    o->tile_count = 1;
}

// ---- host math ---------------------------------------------------------------------------
void rtc_matrix_identity(double out[16]) { Mat4::identity().to16(out); }
void rtc_matrix_translation(double x, double y, double z, double out[16]) { mat_translation(x, y, z).to16(out); }
void rtc_matrix_scaling(double x, double y, double z, double out[16]) { mat_scaling(x, y, z).to16(out); }
void rtc_matrix_rotation_x(double r, double out[16]) { mat_rotation_x(r).to16(out); }
void rtc_matrix_rotation_y(double r, double out[16]) { mat_rotation_y(r).to16(out); }
void rtc_matrix_rotation_z(double r, double out[16]) { mat_rotation_z(r).to16(out); }
void rtc_matrix_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double out[16]) {
    mat_shearing(xy, xz, yx, yz, zx, zy).to16(out);
}
void rtc_matrix_multiply(const double a[16], const double b[16], double out[16]) {
    mat_mul(Mat4::from16(a), Mat4::from16(b)).to16(out);
}
int rtc_matrix_inverse(const double a[16], double out[16]) {
    Mat4 inv;
    if (!mat_inverse(Mat4::from16(a), inv)) {
        set_error("matrix is not invertible (determinant == 0)");
        return RTC_ERR_NON_INVERTIBLE;
    }
    inv.to16(out);
    return RTC_OK;
}
void rtc_view_transform(const double from[3], const double to[3], const double up[3], double out[16]) {
    mat_view({from[0], from[1], from[2]}, {to[0], to[1], to[2]}, {up[0], up[1], up[2]}).to16(out);
}

// ---- scene construction --------------------------------------------------------------------
rtc_scene *rtc_scene_create(void) { return new (std::nothrow) rtc_scene(); }
void rtc_scene_destroy(rtc_scene *scene) { delete scene; }

static int check_parent(Scene &s, bool triangles);

#define CHECK_SCENE(sc)                      \
    if (!(sc)) {                             \
        set_error("scene handle is NULL");   \
        return RTC_ERR_INVALID;              \
    }
#define CHECK_OPEN(sc)                                        \
    CHECK_SCENE(sc)                                           \
    if ((sc)->s.committed) {                                  \
        set_error("scene is committed and can not be edited"); \
        return RTC_ERR_STATE;                                 \
    }

int rtc_scene_set_camera(rtc_scene *scene, uint32_t hsize, uint32_t vsize, double fov, const double view[16]) {
    CHECK_OPEN(scene)
    if (hsize == 0 || vsize == 0) {
        set_error("camera hsize/vsize must be positive");
        return RTC_ERR_INVALID;
    }
    Scene &s = scene->s;
    s.has_camera = true;
    s.hsize = hsize, s.vsize = vsize, s.fov = fov;
    s.view = view ? Mat4::from16(view) : Mat4::identity();
    return RTC_OK;
}

int rtc_scene_add_point_light(rtc_scene *scene, const double position[3], const double intensity[3]) {
    CHECK_OPEN(scene)
    DLight l{};
    for (int i = 0; i < 3; i++) l.position[i] = position[i], l.intensity[i] = intensity[i];
    l.samples = 1;
    scene->s.lights.push_back(l);
    return RTC_OK;
}

int rtc_scene_add_area_light(rtc_scene *scene, const double corner[3], const double full_uvec[3], uint32_t usteps,
                             const double full_vvec[3], uint32_t vsteps, const double intensity[3]) {
    CHECK_OPEN(scene)
    // light.rs:80-104
    DLight l{};
    l.is_area = 1;
    l.usteps = usteps, l.vsteps = vsteps, l.samples = usteps * vsteps;
    for (int i = 0; i < 3; i++) {
        l.corner[i] = corner[i];
        l.uvec[i] = full_uvec[i] / (double)usteps;
        l.vvec[i] = full_vvec[i] / (double)vsteps;
        l.position[i] = (full_uvec[i] + full_vvec[i]) / 2.;  // Q12: corner is not added (light.rs:91-92)
        l.intensity[i] = intensity[i];
    }
    scene->s.lights.push_back(l);
    return RTC_OK;
}

int rtc_scene_add_material(rtc_scene *scene, const rtc_material *m, uint32_t *out_id) {
    CHECK_OPEN(scene)
    if (!m) {
        set_error("material is NULL");
        return RTC_ERR_INVALID;
    }
    if (out_id) *out_id = (uint32_t)scene->s.materials.size();
    scene->s.materials.push_back(*m);
    return RTC_OK;
}

int rtc_scene_add_pattern(rtc_scene *scene, int type, const double a[3], const double b[3], int32_t sub_a,
                          int32_t sub_b, const double transform[16], uint32_t *out_id) {
    CHECK_OPEN(scene)
    Scene &s = scene->s;
    if (type < RTC_PAT_STRIPE || type > RTC_PAT_TEST) {
        set_error("unknown pattern type %d", type);
        return RTC_ERR_INVALID;
    }
    const bool needs_sub = type == RTC_PAT_STRIPE || type == RTC_PAT_CHECKER || type == RTC_PAT_BLENDED;
    const int32_t n = (int32_t)s.patterns.size();
    if (needs_sub && (sub_a < 0 || sub_a >= n || sub_b < 0 || sub_b >= n)) {
        set_error("pattern type %d needs two existing sub-pattern ids", type);
        return RTC_ERR_INVALID;
    }
    HostPattern p{};
    p.kind = type, p.sub_a = sub_a, p.sub_b = sub_b;
    for (int i = 0; i < 3; i++) p.a[i] = a ? a[i] : 1.0, p.b[i] = b ? b[i] : 0.0;
    p.transform = transform ? Mat4::from16(transform) : Mat4::identity();
    if (out_id) *out_id = (uint32_t)n;
    s.patterns.push_back(p);
    return RTC_OK;
}

int rtc_scene_begin_group(rtc_scene *scene, const double transform[16], int32_t material_id) {
    CHECK_OPEN(scene)
    if (int rc = check_parent(scene->s, false)) return rc;
    Item it;
    it.kind = Item::GROUP;
    it.transform = transform ? Mat4::from16(transform) : Mat4::identity();
    it.material = material_id;
    const int idx = scene->s.add_item(std::move(it));
    scene->s.open_groups.push_back(idx);
    return RTC_OK;
}

int rtc_scene_end_group(rtc_scene *scene) {
    CHECK_OPEN(scene)
    if (scene->s.open_groups.empty() || scene->s.items[scene->s.open_groups.back()].kind != Item::GROUP) {
        set_error("rtc_scene_end_group without a matching rtc_scene_begin_group");
        return RTC_ERR_STATE;
    }
    scene->s.open_groups.pop_back();
    return RTC_OK;
}

// a Csg takes exactly a left and a right operand, and no bare triangles
static int check_parent(Scene &s, bool triangles) {
    if (s.open_groups.empty()) return RTC_OK;
    const Item &parent = s.items[s.open_groups.back()];
    if (parent.kind != Item::CSG) return RTC_OK;
    if (triangles) {
        set_error("triangles cannot be direct children of a Csg (Triangle::material() panics, shape.rs:1521-1523): "
                  "put them in a group");
        return RTC_ERR_INVALID;
    }
    if (parent.children.size() >= 2) {
        set_error("a Csg takes exactly a left and a right shape (Csg::new, shape.rs:2166)");
        return RTC_ERR_STATE;
    }
    return RTC_OK;
}

int rtc_scene_begin_csg(rtc_scene *scene, int operation) {
    CHECK_OPEN(scene)
    if (operation < RTC_CSG_UNION || operation > RTC_CSG_DIFFERENCE) {
        set_error("unknown Csg operation %d", operation);
        return RTC_ERR_INVALID;
    }
    if (int rc = check_parent(scene->s, false)) return rc;
    Item it;
    it.kind = Item::CSG;
    it.type = operation;
    const int idx = scene->s.add_item(std::move(it));
    scene->s.open_groups.push_back(idx);
    return RTC_OK;
}

int rtc_scene_end_csg(rtc_scene *scene) {
    CHECK_OPEN(scene)
    Scene &s = scene->s;
    if (s.open_groups.empty() || s.items[s.open_groups.back()].kind != Item::CSG) {
        set_error("rtc_scene_end_csg without a matching rtc_scene_begin_csg");
        return RTC_ERR_STATE;
    }
    if (s.items[s.open_groups.back()].children.size() != 2) {
        set_error("a Csg takes exactly a left and a right shape (Csg::new, shape.rs:2166), %zu added",
                  s.items[s.open_groups.back()].children.size());
        return RTC_ERR_STATE;
    }
    s.open_groups.pop_back();
    return RTC_OK;
}

int rtc_scene_add_shape(rtc_scene *scene, int type, const double transform[16], uint32_t material_id,
                        double minimum, double maximum, int closed) {
    CHECK_OPEN(scene)
    if (int rc = check_parent(scene->s, false)) return rc;
    if (type < RTC_SPHERE || type > RTC_CONE) {
        set_error("unknown shape type %d", type);
        return RTC_ERR_INVALID;
    }
    Item it;
    it.kind = Item::SHAPE;
    it.type = type;
    it.transform = transform ? Mat4::from16(transform) : Mat4::identity();
    it.material = (int32_t)material_id;
    it.minimum = minimum, it.maximum = maximum, it.closed = closed ? 1 : 0;
    scene->s.add_item(std::move(it));
    return RTC_OK;
}

int rtc_scene_add_triangles(rtc_scene *scene, size_t n, const double *points, const double *normals,
                            uint32_t material_id) {
    CHECK_OPEN(scene)
    if (n == 0) return RTC_OK;
    if (int rc = check_parent(scene->s, true)) return rc;
    if (!points) {
        set_error("points is NULL");
        return RTC_ERR_INVALID;
    }
    Item it;
    it.kind = Item::TRIANGLES;
    it.material = (int32_t)material_id;
    it.points.assign(points, points + 9 * n);
    if (normals) it.normals.assign(normals, normals + 9 * n);
    scene->s.add_item(std::move(it));
    return RTC_OK;
}

int rtc_scene_commit(rtc_scene *scene, int device) {
    CHECK_SCENE(scene)
    return scene->s.commit(&device, 1);
}

int rtc_scene_commit_devices(rtc_scene *scene, const int *devices, uint32_t n_devices) {
    CHECK_SCENE(scene)
    if (!devices || n_devices == 0 || n_devices > RTC_MAX_DEVICES) {
        set_error("device list must hold 1..%d entries", RTC_MAX_DEVICES);
        return RTC_ERR_INVALID;
    }
    return scene->s.commit(devices, n_devices);
}

// ---- render ----------------------------------------------------------------------------------

uint32_t rtc_local_rows(uint32_t vsize, uint32_t tile_rows, uint32_t tile_index, uint32_t tile_count) {
    return local_rows(vsize, tile_rows, tile_index, tile_count);
}

int rtc_local_row_map(uint32_t vsize, uint32_t tile_rows, uint32_t tile_index, uint32_t tile_count,
                      uint32_t *rows_out, uint32_t cap) {
    if (tile_count == 0) tile_count = 1;
    if (tile_rows == 0 || tile_rows > vsize) tile_rows = vsize ? vsize : 1;
    const uint32_t ntiles = (vsize + tile_rows - 1) / tile_rows;
    uint32_t n = 0;
    for (uint32_t t = tile_index; t < ntiles; t += tile_count)
        for (uint32_t r = t * tile_rows; r < std::min(vsize, (t + 1) * tile_rows); r++) {
            if (n < cap && rows_out) rows_out[n] = r;
            n++;
        }
    return (int)n;
}

static int resolve_params(Scene &s, const rtc_render_opts *opts_in, DRenderParams &p, rtc_render_opts &o) {
    if (opts_in)
        o = *opts_in;
    else
        rtc_render_opts_default(&o);
    if (!s.committed || s.device < 0 || s.reps.empty()) {
        set_error("scene is not committed to a HIP device; librtc_amd renders on the GPU only");
        return s.committed ? RTC_ERR_NO_DEVICE : RTC_ERR_STATE;
    }
    if (o.tile_count == 0) o.tile_count = 1;
    if (o.tile_index >= o.tile_count) {
        set_error("tile_index %u >= tile_count %u", o.tile_index, o.tile_count);
        return RTC_ERR_INVALID;
    }
    if (o.tile_rows == 0 || o.tile_rows > s.vsize) {
        if (o.tile_count > 1 && o.tile_rows == 0) {
            set_error("tile_rows must be set when tile_count > 1");
            return RTC_ERR_INVALID;
        }
        o.tile_rows = s.vsize;
    }
    // the explicit per-lane stack holds one pending refraction per distinct `remaining` value
    if (o.depth > 8) {
        set_error("depth %u exceeds the per-lane stack of waiting refracted rays (8)", o.depth);
        return RTC_ERR_INVALID;
    }
    if (o.fp_mode != 0) {
        set_error("fp_mode %u: only 0 (strict: the reference's arithmetic) exists; the contracted flavour of rounds 1-2 "
                  "did not meet the 1/255 bar behind glass and was removed", o.fp_mode);
        return RTC_ERR_INVALID;
    }
    if (o.schedule > 3) {
        set_error("schedule %u: 0 = auto, 1 = workgroup per tile, 2 = persistent, 3 = wavefront", o.schedule);
        return RTC_ERR_INVALID;
    }
    if (o.schedule == 3 && o.collect_stats) {
        set_error("the wavefront schedule has no counting variant (collect_stats)");
        return RTC_ERR_INVALID;
    }
    p = DRenderParams{};
    p.depth = o.depth;
    p.tile_rows = o.tile_rows, p.tile_index = o.tile_index, p.tile_count = o.tile_count;
    p.local_rows = local_rows(s.vsize, o.tile_rows, o.tile_index, o.tile_count);
    // auto: the BVH pays as soon as a few bounded primitives exist
    const size_t bounded = s.prims.size() - s.unbounded.size();
    p.use_bvh = (o.use_bvh == 1) ? 1u : (o.use_bvh == 2) ? 0u : (bounded > 4 ? 1u : 0u);
    // auto: deep ray trees come from transparent surfaces (refraction keeps the bounce budget, world.rs:142)
    p.persistent = (o.schedule == 2) ? 1u : (o.schedule == 1) ? 0u : (s.n_transparent_prims > 64 ? 1u : 0u);
    p.share = (o.exchange == 1) ? 0u : 1u;
    // (share = 1 + N: a wave without work waits N polls for rays of others before it leaves, rtc_render_opts.tail_polls;
    // share = 1: until the frame ends)
    if (p.share && o.tail_polls) p.share = 1u + std::min<uint32_t>(o.tail_polls, 1u << 20);
    {
        const char *e = getenv("RTC_TAIL_POLLS");  // (A/B)
        if (p.share && e && atoi(e) > 0) p.share = 1u + (unsigned)atoi(e);
    }
    p.max_rays = o.max_rays_per_pixel;
    p.tiles_x = (s.hsize + 7u) / 8u;
    p.tiles_y = (p.local_rows + 7u) / 8u;
    // a lane's pixel is tile * 64 + position in 32 bits
    if ((uint64_t)p.tiles_x * p.tiles_y >= (1ull << 26)) {
        set_error("%u x %u pixels: a launch covers at most 2^26 8x8 tiles", s.hsize, p.local_rows);
        return RTC_ERR_INVALID;
    }
    return RTC_OK;
}

static void fill_stats(Scene &s, const std::vector<uint64_t> &c, rtc_stats &st) {
    st.primary_rays = c[ST_PRIMARY];
    st.shadow_rays = c[ST_SHADOW];
    st.shadow_rays_skipped = c[ST_SHADOW_SKIPPED];
    st.secondary_rays = c[ST_SECONDARY];
    st.secondary_rays_reference = c[ST_SECONDARY_REF];
    st.container_passes = c[ST_CONTAINER];
    st.box_tests = c[ST_BOX];
    for (int i = 0; i < 7; i++) st.prim_tests[i] = c[ST_PRIM0 + i];
    st.xform_levels = c[ST_XFORM];
    st.shaded_hits = c[ST_SHADED];
    st.light_evals = c[ST_LIGHT];
    st.schlick_evals = c[ST_SCHLICK];
    st.refract_evals = c[ST_REFRACT];
    st.overflow = c[ST_OVERFLOW];
    for (int i = 0; i < 10; i++) st.diag[i] = c[ST_CYC_REFILL + i];
    // SURVEY.md 8(d) flop table (+,-,*,/,sqrt,pow = 1 each)
    double cyl = 26, cone = 32;
    {
        double ncyl = 0, ncyl_closed = 0, ncone = 0, ncone_closed = 0;
        for (const DPrim &p : s.prims) {
            if (p.type == PT_CYLINDER) ncyl++, ncyl_closed += p.closed ? 1 : 0;
            if (p.type == PT_CONE) ncone++, ncone_closed += p.closed ? 1 : 0;
        }
        if (ncyl > 0) cyl = 26 + 18 * (ncyl_closed / ncyl);
        if (ncone > 0) cone = 32 + 18 * (ncone_closed / ncone);
    }
    const double per_type[7] = {27, 1, 12, cyl, cone, 46, 46};
    // a skipped shadow query still forms its ray (18 flops, world.rs:79-81): that is how the kernel learns that the
    // light is behind the surface
    double f = 33.0 * (double)st.xform_levels + 12.0 * (double)st.box_tests + 79.0 * (double)st.shaded_hits +
               67.0 * (double)st.light_evals + 18.0 * (double)(st.shadow_rays + st.shadow_rays_skipped) +
               25.0 * (double)st.schlick_evals + 26.0 * (double)st.refract_evals;
    for (int i = 0; i < 7; i++) f += per_type[i] * (double)st.prim_tests[i];
    st.algorithmic_flops = f;
}

// one asynchronous launch on replica r with `work` as the grid's work buffer
// Queues of the wavefront schedule, allocated at the first use: twelve items per pixel and kind (RTC_WF_CAP_FACTOR) --
// the generations of a glass-heavy frame grow to ten times its pixel count before they shrink (stress scene:
// profiles/r2_ab/r2_wf_trace.txt) -- at 448 bytes per item over both queues.  A frame whose queues would pass 96 GB is
// refused: render it in shares of its rows (rtc_render_opts.tile_*).
static int ensure_wavefront(DeviceReplica &r, size_t pixels) {
    static const unsigned factor_env = getenv("RTC_WF_CAP_FACTOR") ? (unsigned)atoi(getenv("RTC_WF_CAP_FACTOR")) : 0u;
    const size_t want = std::max<size_t>(pixels * (factor_env ? factor_env : 12u), (size_t)1 << 20);
    const size_t bytes = 3 * want * (2 * (sizeof(WfRay) + sizeof(WfCtx)) + sizeof(WfHit));
    if (want >= ((size_t)1 << 30) || bytes > ((size_t)96 << 30)) {
        set_error("wavefront schedule: %zu pixels need %zu GB of queues; render the frame in shares of its rows", pixels,
                  bytes >> 30);
        return RTC_ERR_INVALID;
    }
    WfBuffers &b = r.wf;
    if (b.cap >= want) return RTC_OK;
    for (int k = 0; k < 2; k++) {
        if (b.rays[k]) (void)hipFree(b.rays[k]);
        if (b.ctx[k]) (void)hipFree(b.ctx[k]);
        b.rays[k] = nullptr, b.ctx[k] = nullptr;
    }
    if (b.hits) (void)hipFree(b.hits);
    b.hits = nullptr, b.cap = 0;
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++)
        ok = hipMalloc((void **)&b.rays[k], 3 * want * sizeof(WfRay)) == hipSuccess &&
             hipMalloc((void **)&b.ctx[k], 3 * want * sizeof(WfCtx)) == hipSuccess;
    ok = ok && hipMalloc((void **)&b.hits, 3 * want * sizeof(WfHit)) == hipSuccess;
    if (ok && !b.counts) ok = hipMalloc(&b.counts, 256) == hipSuccess && hipHostMalloc(&b.h_counts, 256, hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        set_error("wavefront schedule: allocating queues for %zu items per kind failed: %s", want,
                  hipGetErrorString(hipGetLastError()));
        return RTC_ERR_HIP;
    }
    b.cap = (uint32_t)want;
    return RTC_OK;
}

static int launch_on(Scene &s, DeviceReplica &r, const DRenderParams &p, const rtc_render_opts &o, double *d_rgb,
                     unsigned *work, hipStream_t st, bool refuse_oversize = false) {
    if (o.schedule == 3) {
        // the wavefront schedule: a sequence of launches driven from here, complete when the call returns
        if (refuse_oversize) {
            set_error("the wavefront schedule allocates its queues: not through the allocation-free entry point");
            return RTC_ERR_INVALID;
        }
        int rc = ensure_wavefront(r, (size_t)p.local_rows * s.hsize);
        if (rc) return rc;
        r.last_work = work;
        return launch_wavefront(r.dscene, p, d_rgb, r.wf, work, r.n_cu, st,
                                                                      &r.last_launch_id, s.hsize, &r.wf_generations);
    }
    if (o.collect_stats) {
        if (hipMemsetAsync(r.d_stats, 0, sizeof(uint64_t) * ST_COUNT, st) != hipSuccess) {
            set_error("hipMemsetAsync(stats) failed");
            return RTC_ERR_HIP;
        }
        s.have_last_stats = true;
    }
    r.last_work = work;
    DRenderParams pp = p;
    // Tile order, heaviest tiles first.  The static schedule may render the heavy ones in parts whose free lanes take over
    // waiting rays (rtc_types.h): pixels then depend on which lane traced which part of a ray tree to ~1e-16, so
    // exchange = 1 (the bit-reproducible frame) keeps whole tiles.  The persistent schedule always takes whole tiles.
    pp.tile_perm = nullptr, pp.n_heavy = 0, pp.part_shift = 0;
    uint32_t n_heavy = 0;
    const bool whole = p.tile_count <= 1 && (size_t)p.tiles_x * p.tiles_y == s.tile_perm.size();
    if (whole) {
        pp.tile_perm = r.d_tile_perm;
        n_heavy = s.n_heavy_tiles;
    } else if (p.tile_count > 1 && !s.tile_score.empty()) {
        // a share of the frame (one rank's or one replica's interleaved row tiles): its tiles in the order of the whole
        // frame's weights.  Built and uploaded at the first launch with this share -- the one allocation a render entry
        // point makes after commit, which is why the allocation-free entry point only uses an order that exists already.
        // Every share keeps a buffer of its own (never rewritten): a launch of another share that is still in flight on
        // another stream goes on reading its own order.
        const uint32_t n = p.tiles_x * p.tiles_y;
        DeviceReplica::SharePerm *have = nullptr;
        for (DeviceReplica::SharePerm &sp : r.share_perms)
            if (sp.key[0] == p.tile_rows && sp.key[1] == p.tile_index && sp.key[2] == p.tile_count && sp.n_tiles == n) have = &sp;
        if (!have && !refuse_oversize) {
            const uint32_t gx = (s.hsize + 7u) / 8u, gy = (s.vsize + 7u) / 8u;
            std::vector<int32_t> score(n);
            for (uint32_t ty = 0; ty < p.tiles_y; ty++) {
                const uint32_t l = 8u * ty, k = l / p.tile_rows;
                const uint32_t y = (k * p.tile_count + p.tile_index) * p.tile_rows + l % p.tile_rows;  // canvas row
                const uint32_t row = std::min(y / 8u, gy - 1u);
                for (uint32_t tx = 0; tx < p.tiles_x; tx++) score[(size_t)ty * p.tiles_x + tx] = s.tile_score[(size_t)row * gx + std::min(tx, gx - 1u)];
            }
            std::vector<uint32_t> perm(n);
            for (uint32_t t = 0; t < n; t++) perm[t] = t;
            std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t c) { return score[a] > score[c]; });
            DeviceReplica::SharePerm sp{{p.tile_rows, p.tile_index, p.tile_count}, n, nullptr, 0};
            for (uint32_t t = 0; t < n; t++) sp.n_heavy += score[t] >= RTC_HEAVY_TILE_SCORE ? 1u : 0u;
            if (hipMalloc((void **)&sp.d_perm, sizeof(uint32_t) * n) == hipSuccess &&
                hipMemcpy(sp.d_perm, perm.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice) == hipSuccess) {
                r.share_perms.push_back(sp);
                have = &r.share_perms.back();
            } else {  // (no order: the raster order is always correct)
                if (sp.d_perm) (void)hipFree(sp.d_perm);
                (void)hipGetLastError();
            }
        }
        if (have) pp.tile_perm = have->d_perm, n_heavy = have->n_heavy;
    }
    if (!p.persistent && p.share != 0u && pp.tile_perm && n_heavy > 0) {
        // Parts per heavy tile ("heavy": the projected bounds of something reflective or transparent touch it).  Splitting
        // shortens a tile's critical path (cover.yaml at 1080p, the band through the glass sphere alone: 1.6 ms whole,
        // 0.5 ms in quarters) and costs instructions (the lanes of a part sit in different blocks of the state machine:
        // about twice the wave-cycles in quarters), so it pays while the device has room.  Calibrated on cover.yaml and
        // reflective_world.yaml from 320x180 to 3840x2160 (profiles/r3_ab/r3_parts_ladder_heavy1.txt; with "heavy" =
        // transparent only, r3_parts_ladder_heavy4.txt, small frames were 10-20 % slower): quarters up to twice the
        // resident waves' worth of heavy tiles, halves up to ten times, whole tiles beyond.  The two scenes disagree by
        // 10-20 % on where the lines should be (cover likes quarters longer); these minimise the worst case.
        // RTC_TILE_SPLIT = 1 / 2 / 4 forces the number of parts.
        const int forced = getenv("RTC_TILE_SPLIT") ? atoi(getenv("RTC_TILE_SPLIT")) : 0;  // (read per launch: an A/B knob)
        const uint64_t slots = (uint64_t)r.n_cu * 4u * RTC_MIN_WAVES;
        uint32_t shift = (uint64_t)n_heavy <= 2u * slots ? 2u : ((uint64_t)n_heavy <= 10u * slots ? 1u : 0u);
        if (forced == 1) shift = 0;
        if (forced == 2) shift = 1;
        if (forced == 4) shift = 2;
        if (shift) pp.n_heavy = n_heavy, pp.part_shift = shift;
    }
    return launch_render(r.dscene, pp, d_rgb, r.d_stats, work, r.n_cu,
                                                            o.collect_stats != 0, st, &r.last_launch_id, refuse_oversize);
}

// Did a lane of replica r's most recent launch exceed a per-lane bound?  (the device must be idle)
static int read_overflow(DeviceReplica &r, bool &overflow) {
    overflow = false;
    if (!r.last_work || r.last_launch_id == 0) return RTC_OK;
    uint32_t flag = 0;
    if (hipMemcpy(&flag, r.last_work + RTC_WORK_FLAG, sizeof flag, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("hipMemcpy(overflow flag) failed: %s", hipGetErrorString(hipGetLastError()));
        return RTC_ERR_HIP;
    }
    overflow = flag == r.last_launch_id;
    return RTC_OK;
}

static int overflow_error() {
    set_error("a lane exceeded a per-lane bound of the kernel (rays per pixel, waiting refracted rays, traversal stack, "
              "records per Csg tree): the pixels concerned are undefined (NaN for the ray cap), discard the frame");
    return RTC_ERR_OVERFLOW;
}

static int render_device_common(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream,
                                bool static_scratch_only) {
    CHECK_SCENE(scene)
    Scene &s = scene->s;
    DRenderParams p;
    rtc_render_opts o;
    int rc = resolve_params(s, opts, p, o);
    if (rc) return rc;
    if (!d_rgb) {
        set_error("d_rgb is NULL");
        return RTC_ERR_INVALID;
    }
    DeviceReplica &r = s.reps[0];
    unsigned *work = r.d_work;
    if (static_scratch_only) {
        if (o.collect_stats) {
            set_error("rtc_render_device_static does not run the counting kernel (collect_stats)");
            return RTC_ERR_INVALID;
        }
        if ((rc = static_scratch(r.device, &work))) return rc;
    }
    if (p.local_rows == 0) return RTC_OK;
    DeviceGuard guard(r.device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", r.device);
        return RTC_ERR_HIP;
    }
    // (the static scratch holds mailboxes for RTC_MAX_WAVES resident waves: the static variant refuses a larger grid
    // with a clear error where rtc_render_device switches the ray exchange off)
    return launch_on(s, r, p, o, d_rgb, work, (hipStream_t)hip_stream, static_scratch_only);
}

int rtc_render_device(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream) {
    return render_device_common(scene, opts, d_rgb, hip_stream, false);
}

int rtc_render_device_static(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream) {
    return render_device_common(scene, opts, d_rgb, hip_stream, true);
}

int rtc_scene_check(rtc_scene *scene) {
    CHECK_SCENE(scene)
    Scene &s = scene->s;
    if (!s.committed || s.device < 0 || s.reps.empty()) {
        set_error("scene is not committed to a device");
        return RTC_ERR_STATE;
    }
    bool any = false;
    for (DeviceReplica &r : s.reps) {
        DeviceGuard guard(r.device);
        if (hipDeviceSynchronize() != hipSuccess) {
            set_error("hipDeviceSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
            return RTC_ERR_HIP;
        }
        bool ov = false;
        if (int rc = read_overflow(r, ov)) return rc;
        any = any || ov;
    }
    return any ? overflow_error() : RTC_OK;
}

int rtc_scene_last_stats(rtc_scene *scene, rtc_stats *stats) {
    CHECK_SCENE(scene)
    Scene &s = scene->s;
    if (!s.committed || s.device < 0 || s.reps.empty()) {
        set_error("scene is not committed to a device");
        return RTC_ERR_STATE;
    }
    if (!stats) return RTC_ERR_INVALID;
    std::vector<uint64_t> total(ST_COUNT, 0), c(ST_COUNT, 0);
    bool overflow = false;
    for (DeviceReplica &r : s.reps) {
        DeviceGuard guard(r.device);
        if (hipDeviceSynchronize() != hipSuccess) {
            set_error("hipDeviceSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
            return RTC_ERR_HIP;
        }
        if (hipMemcpy(c.data(), r.d_stats, sizeof(uint64_t) * ST_COUNT, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("hipMemcpy(stats) failed");
            return RTC_ERR_HIP;
        }
        for (int i = 0; i < ST_COUNT; i++) total[i] += c[i];
        bool ov = false;
        if (int rc = read_overflow(r, ov)) return rc;
        overflow = overflow || ov;
    }
    rtc_stats st{};
    fill_stats(s, total, st);
    st.pixels = st.primary_rays;
    st.kernel_ms = s.last_stats.kernel_ms;
    *stats = st;
    return (st.overflow || overflow) ? overflow_error() : RTC_OK;
}

// Device-to-host copy of `bytes` through the scene's two pinned staging buffers: chunk k+1 crosses PCIe while
// chunk k is copied out of its staging buffer.  place(offset, src, n) stores n bytes that start `offset` bytes into
// the device buffer.
typedef std::function<void(size_t, const char *, size_t)> Place;
// Three staging buffers, and the copy out of them shared by a few host threads: with the DMA engine writing into host
// memory at the same time ONE thread moves ~27 GB/s, which made the copy -- not PCIe -- the limit of a 4K frame's
// download (tools/micro/d2h_pipeline.hip: 7.1 ms with one thread, 4.3 ms with two, DMA alone 3.5 ms).  The helpers live
// for one call; place() is called on disjoint byte ranges.  RTC_COPY_THREADS overrides the count (1 = the caller only).
static int staged_download(Scene &s, const char *d_src, size_t bytes, hipStream_t st, const Place &place) {
    const size_t CHUNK = (size_t)16 << 20;
    constexpr size_t NBUF = sizeof(s.h_stage) / sizeof(s.h_stage[0]);
    if (!s.h_stage[0]) {
        for (void *&h : s.h_stage)
            if (hipHostMalloc(&h, CHUNK, hipHostMallocDefault) != hipSuccess) {
                set_error("hipHostMalloc(%zu) for the staging buffer failed", CHUNK);
                return RTC_ERR_HIP;
            }
        s.stage_bytes = CHUNK;
    }
    hipEvent_t ev[NBUF];
    for (hipEvent_t &e : ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            set_error("hipEventCreate failed");
            return RTC_ERR_HIP;
        }
    int rc = RTC_OK;
    const size_t n_chunks = (bytes + CHUNK - 1) / CHUNK;
    static const int threads_env = getenv("RTC_COPY_THREADS") ? atoi(getenv("RTC_COPY_THREADS")) : 0;
    const int threads = n_chunks < 2 ? 1 : std::max(1, std::min(8, threads_env > 0 ? threads_env : 3));
    auto chunk_len = [&](size_t k) { return std::min(CHUNK, bytes - k * CHUNK); };
    // helper t copies slice t of every chunk once the caller's thread has seen the chunk arrive
    std::atomic<size_t> ready{0};
    std::atomic<bool> stop{false};
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[std::max<size_t>(n_chunks, 1)]);
    for (size_t k = 0; k < n_chunks; k++) done[k].store(0);
    auto slice = [&](size_t k, int t) {
        const size_t n = chunk_len(k), a = n * (size_t)t / (size_t)threads, b = n * (size_t)(t + 1) / (size_t)threads;
        if (b > a) place(k * CHUNK + a, (const char *)s.h_stage[k % NBUF] + a, b - a);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++)
        pool.emplace_back([&, t] {
            for (size_t k = 0; k < n_chunks; k++) {
                while (ready.load(std::memory_order_acquire) <= k) {
                    if (stop.load(std::memory_order_relaxed)) return;
                    std::this_thread::yield();
                }
                slice(k, t);
                done[k].fetch_add(1, std::memory_order_release);
            }
        });
    auto issue = [&](size_t k) {
        // (the buffer is free again: every thread is through with the chunk that used it last)
        if (k >= NBUF)
            while (done[k - NBUF].load(std::memory_order_acquire) < threads - 1) std::this_thread::yield();
        if (hipMemcpyAsync(s.h_stage[k % NBUF], d_src + k * CHUNK, chunk_len(k), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipEventRecord(ev[k % NBUF], st) != hipSuccess) {
            set_error("device-to-host copy of the canvas failed: %s", hipGetErrorString(hipGetLastError()));
            rc = RTC_ERR_HIP;
        }
    };
    for (size_t k = 0; k + 1 < NBUF && k < n_chunks && rc == RTC_OK; k++) issue(k);
    for (size_t k = 0; k < n_chunks && rc == RTC_OK; k++) {
        if (k + NBUF - 1 < n_chunks) issue(k + NBUF - 1);
        if (rc != RTC_OK) break;
        if (hipEventSynchronize(ev[k % NBUF]) != hipSuccess) {
            set_error("device-to-host copy of the canvas failed: %s", hipGetErrorString(hipGetLastError()));
            rc = RTC_ERR_HIP;
            break;
        }
        ready.store(k + 1, std::memory_order_release);
        slice(k, 0);
    }
    if (rc != RTC_OK) {
        stop.store(true);
        (void)hipStreamSynchronize(st);  // nothing may still write the staging buffers
    }
    for (std::thread &th : pool) th.join();
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    return rc;
}

static int ensure_canvas(DeviceReplica &r, size_t bytes) {
    if (r.canvas_bytes >= bytes) return RTC_OK;
    if (r.d_canvas) (void)hipFree(r.d_canvas);
    r.d_canvas = nullptr, r.canvas_bytes = 0;
    if (hipMalloc((void **)&r.d_canvas, bytes) != hipSuccess) {
        set_error("hipMalloc(%zu) for the canvas failed", bytes);
        return RTC_ERR_HIP;
    }
    r.canvas_bytes = bytes;
    return RTC_OK;
}

int rtc_render(rtc_scene *scene, const rtc_render_opts *opts, double *rgb_out, rtc_stats *stats) {
    CHECK_SCENE(scene)
    Scene &s = scene->s;
    DRenderParams p;
    rtc_render_opts o;
    int rc = resolve_params(s, opts, p, o);
    if (rc) return rc;
    if (!rgb_out) {
        set_error("rgb_out is NULL");
        return RTC_ERR_INVALID;
    }
    const size_t row_doubles = (size_t)s.hsize * 3, row_bytes = row_doubles * sizeof(double);
    const uint32_t n_rep = (uint32_t)s.reps.size();
    const bool multi = n_rep > 1 && o.tile_count == 1;
    if (stats) std::memset(stats, 0, sizeof *stats);
    // Canvas::new: all black (canvas.rs:14-29).  A whole-frame render writes every row, a tiled one only its own.
    if (o.tile_count > 1) std::memset(rgb_out, 0, row_bytes * s.vsize);
    if (p.local_rows == 0) return RTC_OK;
    DeviceReplica &r0 = s.reps[0];
    DeviceGuard guard(r0.device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", r0.device);
        return RTC_ERR_HIP;
    }
    float ms = 0.f;
    if (!multi) {
        const size_t bytes = row_bytes * p.local_rows;
        if ((rc = ensure_canvas(r0, bytes))) return rc;
        hipStream_t st = nullptr;
        (void)hipEventRecord((hipEvent_t)r0.ev_start, st);
        rc = launch_on(s, r0, p, o, r0.d_canvas, r0.d_work, st);
        (void)hipEventRecord((hipEvent_t)r0.ev_stop, st);
        if (rc) return rc;
        if (o.tile_count == 1) {
            rc = staged_download(s, (const char *)r0.d_canvas, bytes, st,
                                 [&](size_t off, const char *src, size_t n) { std::memcpy((char *)rgb_out + off, src, n); });
        } else {
            std::vector<uint32_t> rows(p.local_rows);
            rtc_local_row_map(s.vsize, o.tile_rows, o.tile_index, o.tile_count, rows.data(), p.local_rows);
            rc = staged_download(s, (const char *)r0.d_canvas, bytes, st, [&](size_t off, const char *src, size_t n) {
                while (n > 0) {  // a chunk holds parts of several compact rows
                    const size_t l = off / row_bytes, in_row = off - l * row_bytes, m = std::min(n, row_bytes - in_row);
                    std::memcpy((char *)rgb_out + (size_t)rows[l] * row_bytes + in_row, src, m);
                    off += m, src += m, n -= m;
                }
            });
        }
        if (rc) return rc;
        if (hipEventSynchronize((hipEvent_t)r0.ev_stop) != hipSuccess) {
            set_error("render kernel failed: %s", hipGetErrorString(hipGetLastError()));
            return RTC_ERR_HIP;
        }
        (void)hipEventElapsedTime(&ms, (hipEvent_t)r0.ev_start, (hipEvent_t)r0.ev_stop);
    } else {
        // ---- one process, several GPUs (camera.rs:72-91): replica k renders the interleaved 8-row
        // tiles t % N == k into a compact buffer on its own device and stream; each buffer then moves into the
        // canvas on the primary device with ONE strided device-to-device copy over its xGMI link (7 peers = 7
        // distinct links into device 0; no ring, no reduction), and the canvas goes to the host once.  The copy of a
        // replica starts when ITS kernel ends, so copies overlap the renders of slower replicas; a replica's own render
        // and copy do not overlap -- a launch is monolithic, and cutting it into bands would cost the heaviest-first tile
        // order, which is worth more wherever the render is not dwarfed by a 25 MB copy. ----
        const uint32_t TILE = 8;
        std::vector<DRenderParams> ps(n_rep);
        std::vector<rtc_render_opts> os(n_rep);
        for (uint32_t k = 0; k < n_rep; k++) {
            rtc_render_opts ok = o;
            ok.tile_rows = TILE, ok.tile_index = k, ok.tile_count = n_rep;
            if ((rc = resolve_params(s, &ok, ps[k], os[k]))) return rc;
        }
        // the canvas lives on the primary device; replica 0's compact buffer sits behind it
        if ((rc = ensure_canvas(r0, row_bytes * (s.vsize + (size_t)ps[0].local_rows)))) return rc;
        auto compact = [&](uint32_t k) { return k == 0 ? r0.d_canvas + row_doubles * s.vsize : s.reps[k].d_canvas; };
        for (uint32_t k = 0; k < n_rep && rc == RTC_OK; k++) {
            DeviceReplica &r = s.reps[k];
            if (ps[k].local_rows == 0) continue;
            DeviceGuard g(r.device);
            if (k > 0 && (rc = ensure_canvas(r, row_bytes * ps[k].local_rows))) break;
            (void)hipEventRecord((hipEvent_t)r.ev_start, (hipStream_t)r.stream);
            rc = launch_on(s, r, ps[k], os[k], compact(k), r.d_work, (hipStream_t)r.stream);
            (void)hipEventRecord((hipEvent_t)r.ev_stop, (hipStream_t)r.stream);
        }
        for (uint32_t k = 0; k < n_rep && rc == RTC_OK; k++) {
            DeviceReplica &r = s.reps[k];
            if (ps[k].local_rows == 0) continue;
            DeviceGuard g(r.device);
            // compact tile j of replica k is canvas tile j * N + k: a 2-d copy whose "rows" are whole tiles (TILE canvas
            // rows = tile_bytes, contiguous on both sides), source pitch one tile, destination pitch N tiles; the last
            // tile of the frame may be shorter than TILE rows and goes separately
            const size_t tile_bytes = row_bytes * TILE;
            const uint32_t full_tiles = ps[k].local_rows / TILE, rest_rows = ps[k].local_rows % TILE;
            char *dst = (char *)r0.d_canvas + (size_t)k * tile_bytes;
            const char *src = (const char *)compact(k);
            hipError_t e = hipSuccess;
            if (full_tiles > 0) {
                e = hipMemcpy2DAsync(dst, (size_t)n_rep * tile_bytes, src, tile_bytes, tile_bytes, full_tiles,
                                     hipMemcpyDeviceToDevice, (hipStream_t)r.stream);
                if (e != hipSuccess) {
                    // (a runtime that refuses the strided form between these two devices -- no peer mapping: one peer
                    // copy per tile, which every pair of devices supports)
                    (void)hipGetLastError();
                    e = hipSuccess;
                    for (uint32_t j = 0; j < full_tiles && e == hipSuccess; j++)
                        e = hipMemcpyPeerAsync(dst + (size_t)j * n_rep * tile_bytes, r0.device, src + (size_t)j * tile_bytes,
                                               r.device, tile_bytes, (hipStream_t)r.stream);
                }
            }
            if (e == hipSuccess && rest_rows > 0)
                e = hipMemcpyPeerAsync(dst + (size_t)full_tiles * n_rep * tile_bytes, r0.device,
                                       src + (size_t)full_tiles * tile_bytes, r.device, row_bytes * rest_rows,
                                       (hipStream_t)r.stream);
            if (e != hipSuccess) {
                set_error("device-to-device copy of replica %u's tiles failed: %s", k, hipGetErrorString(e));
                rc = RTC_ERR_HIP;
            }
        }
        for (uint32_t k = 0; k < n_rep; k++) {
            DeviceGuard g(s.reps[k].device);
            if (hipStreamSynchronize((hipStream_t)s.reps[k].stream) != hipSuccess && rc == RTC_OK) {
                set_error("replica %u failed: %s", k, hipGetErrorString(hipGetLastError()));
                rc = RTC_ERR_HIP;
            }
        }
        if (rc) return rc;
        // kernel time of the frame: the slowest replica's, from the events around its launch on its own stream
        for (uint32_t k = 0; k < n_rep; k++) {
            if (ps[k].local_rows == 0) continue;
            DeviceGuard g(s.reps[k].device);
            float t = 0.f;
            if (hipEventElapsedTime(&t, (hipEvent_t)s.reps[k].ev_start, (hipEvent_t)s.reps[k].ev_stop) == hipSuccess) ms = std::max(ms, t);
        }
        rc = staged_download(s, (const char *)r0.d_canvas, row_bytes * s.vsize, nullptr,
                             [&](size_t off, const char *src, size_t n) { std::memcpy((char *)rgb_out + off, src, n); });
        if (rc) return rc;
    }
    s.last_stats.kernel_ms = ms;
    bool overflow = false;
    for (uint32_t k = 0; k < (multi ? n_rep : 1u); k++) {
        DeviceGuard g(s.reps[k].device);
        bool ov = false;
        if ((rc = read_overflow(s.reps[k], ov))) return rc;
        overflow = overflow || ov;
    }
    if (stats) {
        if (o.collect_stats) {
            rc = rtc_scene_last_stats(scene, stats);
            if (rc && rc != RTC_ERR_OVERFLOW) return rc;
        }
        stats->kernel_ms = ms;
    }
    return overflow ? overflow_error() : RTC_OK;
}

int rtc_scene_get_info(rtc_scene *scene, rtc_scene_info *out) {
    CHECK_SCENE(scene)
    if (!out) return RTC_ERR_INVALID;
    Scene &s = scene->s;
    std::memset(out, 0, sizeof *out);
    out->hsize = s.hsize, out->vsize = s.vsize;
    out->n_prims = (uint32_t)s.prims.size();
    out->n_unbounded = (uint32_t)s.unbounded.size();
    out->n_bvh_nodes = (uint32_t)((s.quantized_nodes && RTC_BVH4) ? s.bvh4.size() : s.bvh.size());
    out->n_xforms = (uint32_t)s.xforms.size();
    out->n_materials = (uint32_t)s.materials.size();
    out->n_lights = (uint32_t)s.lights.size();
    out->n_patterns = (uint32_t)s.patterns.size();
    out->n_triangles = (uint32_t)s.tris.size();
    out->bvh_max_depth = s.bvh_max_depth, out->bvh_max_leaf = s.bvh_max_leaf;
    out->committed = s.committed ? 1 : 0;
    out->device = (uint32_t)s.device;
    out->device_bytes = s.device_bytes;
    out->n_devices = (uint32_t)s.reps.size();
    out->n_heavy_tiles = s.n_heavy_tiles;
    return RTC_OK;
}

// ---- builder-tree introspection -----------------------------------------------------------------
uint32_t rtc_scene_item_count(rtc_scene *scene) { return scene ? (uint32_t)scene->s.items.size() : 0u; }

int rtc_scene_get_item(rtc_scene *scene, uint32_t item, rtc_item_info *out) {
    CHECK_SCENE(scene)
    if (!out || item >= scene->s.items.size()) {
        set_error("item %u out of range", item);
        return RTC_ERR_INVALID;
    }
    const Item &it = scene->s.items[item];
    std::memset(out, 0, sizeof *out);
    out->kind = it.kind == Item::GROUP ? 0 : it.kind == Item::SHAPE ? 1 : it.kind == Item::TRIANGLES ? 2 : 3;
    out->parent = it.parent;
    out->shape_type = it.type;
    out->material_id = it.material;
    out->n_children = (uint32_t)it.children.size();
    out->n_triangles = (uint32_t)(it.points.size() / 9);
    out->closed = it.closed;
    out->smooth = it.normals.empty() ? 0 : 1;
    out->minimum = it.minimum, out->maximum = it.maximum;
    it.transform.to16(out->transform);
    return RTC_OK;
}

uint32_t rtc_scene_child_count(rtc_scene *scene, int32_t parent) {
    if (!scene) return 0;
    if (parent < 0) return (uint32_t)scene->s.top_level.size();
    if ((size_t)parent >= scene->s.items.size()) return 0;
    return (uint32_t)scene->s.items[parent].children.size();
}

int rtc_scene_get_child(rtc_scene *scene, int32_t parent, uint32_t k, uint32_t *out_item) {
    CHECK_SCENE(scene)
    const std::vector<int> *list = nullptr;
    if (parent < 0)
        list = &scene->s.top_level;
    else if ((size_t)parent < scene->s.items.size())
        list = &scene->s.items[parent].children;
    if (!list || k >= list->size() || !out_item) {
        set_error("child %u of item %d out of range", k, parent);
        return RTC_ERR_INVALID;
    }
    *out_item = (uint32_t)(*list)[k];
    return RTC_OK;
}

int rtc_scene_get_material(rtc_scene *scene, uint32_t id, rtc_material *out) {
    CHECK_SCENE(scene)
    if (!out || id >= scene->s.materials.size()) {
        set_error("material %u out of range", id);
        return RTC_ERR_INVALID;
    }
    *out = scene->s.materials[id];
    return RTC_OK;
}

int rtc_scene_get_triangles(rtc_scene *scene, uint32_t item, uint32_t first, uint32_t n, double *points,
                            double *normals) {
    CHECK_SCENE(scene)
    if (item >= scene->s.items.size() || scene->s.items[item].kind != Item::TRIANGLES) {
        set_error("item %u is not a run of triangles", item);
        return RTC_ERR_INVALID;
    }
    const Item &it = scene->s.items[item];
    if ((size_t)first + n > it.points.size() / 9) {
        set_error("triangle range out of bounds");
        return RTC_ERR_INVALID;
    }
    if (points) std::memcpy(points, it.points.data() + (size_t)first * 9, sizeof(double) * 9 * n);
    if (normals) {
        if (it.normals.empty()) {
            set_error("item %u holds flat triangles", item);
            return RTC_ERR_INVALID;
        }
        std::memcpy(normals, it.normals.data() + (size_t)first * 9, sizeof(double) * 9 * n);
    }
    return RTC_OK;
}

int rtc_scene_get_light(rtc_scene *scene, uint32_t index, rtc_light_info *out) {
    CHECK_SCENE(scene)
    if (!out || index >= scene->s.lights.size()) {
        set_error("light %u out of range", index);
        return RTC_ERR_INVALID;
    }
    const DLight &l = scene->s.lights[index];
    std::memset(out, 0, sizeof *out);
    out->is_area = l.is_area, out->usteps = l.usteps, out->vsteps = l.vsteps;
    for (int i = 0; i < 3; i++) {
        out->position[i] = l.position[i], out->intensity[i] = l.intensity[i];
        out->corner[i] = l.corner[i], out->uvec[i] = l.uvec[i], out->vvec[i] = l.vvec[i];
    }
    return RTC_OK;
}

int rtc_scene_get_camera(rtc_scene *scene, uint32_t *hsize, uint32_t *vsize, double *field_of_view, double view[16]) {
    CHECK_SCENE(scene)
    if (!scene->s.has_camera) {
        set_error("scene has no camera");
        return RTC_ERR_STATE;
    }
    if (hsize) *hsize = scene->s.hsize;
    if (vsize) *vsize = scene->s.vsize;
    if (field_of_view) *field_of_view = scene->s.fov;
    if (view) scene->s.view.to16(view);
    return RTC_OK;
}

}  // extern "C"
