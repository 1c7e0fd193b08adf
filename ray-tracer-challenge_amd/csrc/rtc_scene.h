// rtc_scene.h -- host-side scene: builder tree, flattener, BVH builder, device upload.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rtc.h"
#include "rtc_host_math.h"
#include "rtc_types.h"

namespace rtc {

void set_error(const char *fmt, ...);

// One entry of the builder tree (mirrors the reference's Shape tree: shape.rs Group / leaf shapes).
struct Item {
    enum Kind { GROUP, SHAPE, TRIANGLES, CSG } kind;  // CSG: `type` holds the rtc_csg_op, children = {left, right}
    int parent = -1;  // item index of the enclosing group or -1 (member of World.objects)
    Mat4 transform = Mat4::identity();
    int32_t material = -1;
    int32_t type = 0;  // rtc_shape_type for SHAPE
    double minimum = 0, maximum = 0;
    int32_t closed = 0;
    std::vector<int> children;     // GROUP: items in insertion order
    std::vector<double> points;    // TRIANGLES: 9 doubles per triangle
    std::vector<double> normals;   // TRIANGLES: 9 doubles per triangle or empty
};

struct HostPattern {
    int32_t kind, sub_a, sub_b;
    double a[3], b[3];
    Mat4 transform;
};

struct HostLight {
    DLight d;
};

struct Bounds {
    double lo[3], hi[3];
};

// One copy of the committed scene on one GPU (the scene is replicated per device: SURVEY.md 8(e)), with the buffers
// the render entry points keep between calls.
// Queues of the wavefront schedule on one device (allocated at the first schedule-3 render, kept)
struct WfBuffers {
    WfRay *rays[2] = {nullptr, nullptr};  // queue 0 / 1: three regions of `cap` items (closest, containers, shadow)
    WfCtx *ctx[2] = {nullptr, nullptr};
    WfHit *hits = nullptr;                // answers of the queue being traversed
    void *counts = nullptr;               // WfCounts (device)
    void *h_counts = nullptr;             // its pinned host copy
    uint32_t cap = 0;
};

struct DeviceReplica {
    int device = -1;
    DScene dscene{};
    std::vector<void *> allocs;
    uint64_t bytes = 0;
    uint64_t *d_stats = nullptr;  // ST_COUNT counters
    unsigned *d_work = nullptr;   // work buffer of the render grid (layout: rtc_types.h; partly zeroed per launch)
    unsigned n_cu = 0;            // CU count of the device (persistent grid = CUs x resident waves per CU)
    void *ev_start = nullptr, *ev_stop = nullptr;
    void *stream = nullptr;       // replicas of a multi-device scene render on a stream of their own
    double *d_canvas = nullptr;   // rtc_render's device canvas: allocated at the first call, kept, grown on demand
    size_t canvas_bytes = 0;
    const uint32_t *d_tile_perm = nullptr;  // Scene::tile_perm on this device (or null)
    WfBuffers wf;
    uint32_t wf_generations = 0;  // generations of the most recent wavefront frame
    // the same orders for the SHARES of the frame rendered here (rtc_render_opts.tile_*): each built and uploaded at the
    // first launch with that share into a buffer of its own and kept until the scene is released -- a buffer is never
    // rewritten, so launches of different shares that are still in flight on other streams keep reading their own
    struct SharePerm {
        uint32_t key[3];          // tile_rows, tile_index, tile_count
        uint32_t n_tiles;         // tiles of the share
        uint32_t *d_perm;         // the share's tiles in the order of their weights
        uint32_t n_heavy;         // how many of them (the first ones) show a transparent object
    };
    std::vector<SharePerm> share_perms;
    uint32_t last_launch_id = 0;  // id of the most recent launch and the work buffer it used (overflow flag)
    unsigned *last_work = nullptr;
};

struct Scene {
    // ---- builder state ----
    std::vector<Item> items;
    std::vector<int> top_level;    // World.objects order
    std::vector<int> open_groups;  // begin_group / end_group stack
    std::vector<rtc_material> materials;
    std::vector<HostPattern> patterns;
    std::vector<DLight> lights;
    bool has_camera = false;
    uint32_t hsize = 0, vsize = 0;
    double fov = 0;
    Mat4 view = Mat4::identity();

    // ---- committed (flattened) state ----
    bool committed = false;
    int device = -1;
    DCamera cam{};
    std::vector<DXform> xforms;
    std::vector<Mat4> xform_fwd;  // forward matrices, host only (bounds)
    std::vector<int32_t> chains;
    std::vector<DPrim> prims;
    std::vector<DTri> tris;
    std::vector<DMaterial> dmaterials;
    std::vector<DPattern> dpatterns;
    std::vector<DBvhNode> bvh;     // binary f32 nodes; large hierarchies are uploaded as bvh4 instead (quantized_nodes)
    std::vector<DBvhNode4> bvh4;
    std::vector<DBvhNodeQ> bvh_q;
    bool quantized_nodes = false;
    std::vector<uint8_t> bvh_flags;
    double q_org[3] = {0, 0, 0}, q_cell[3] = {1, 1, 1};
    std::vector<int32_t> bvh_prims;
    std::vector<int32_t> compact_prims;  // primitives of the sphere-only leaf runs, in DScene::leaves_s order
    std::vector<int32_t> unbounded;
    std::vector<Bounds> prim_bounds;  // world-space, padded; only meaningful for bounded prims
    std::vector<uint8_t> prim_bounded;
    std::vector<int32_t> prim_anc_id;  // ancestor-chain identity per primitive (-1: top level)
    std::vector<int32_t> prim_csg;     // Csg tree a primitive is a leaf of (-1: none); such primitives are reached only
                                       // through the tree's pseudo-primitive (PT_CSG, appended after the real ones)
    std::vector<DCsgRoot> csg_roots;
    std::vector<DCsgNode> csg_nodes;
    uint32_t bvh_max_depth = 0, bvh_max_leaf = 0;
    uint32_t n_transparent_prims = 0;
    int32_t root_ref = 0;
    double root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};

    // ---- device state: reps[0] is the primary device (`device`), where a multi-device frame is assembled ----
    std::vector<DeviceReplica> reps;
    uint64_t device_bytes = 0;
    // whole-frame tile order of the static schedule: tiles that show transparent / reflective objects first (empty: raster)
    std::vector<uint32_t> tile_perm;
    std::vector<int32_t> tile_score;  // the weights the order was sorted by, per whole-frame tile (empty: all alike)
    uint32_t n_heavy_tiles = 0;  // tiles at the head of tile_perm that show a transparent object (RTC_HEAVY_TILE_SCORE)
    void build_tile_order();
    void *h_stage[3] = {nullptr, nullptr, nullptr};  // pinned staging buffers of rtc_render's device-to-host copy
    size_t stage_bytes = 0;
    rtc_stats last_stats{};
    bool have_last_stats = false;

    ~Scene();
    int add_item(Item &&it);
    int commit(const int *devices, uint32_t n_devices);  // devices[0] == -1: host-side work only
    int flatten();
    void build_bvh();
    int upload(int device, DeviceReplica &r);
    void release_device();
};

// makes `device` current for the lifetime of the guard and restores the caller's device afterwards
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device);
    ~DeviceGuard();
};

#define RTC_MAX_DEVICES 64
// rtc_kernels.hip: the file-scope static device array behind rtc_render_device_static, resolved once per device
int static_scratch(int device, unsigned **out);


// kernels (rtc_kernels.hip)
// launch_id_out receives the id an overflowing lane of THIS launch stores into d_work[RTC_WORK_FLAG]
int launch_render(const DScene &scene, const DRenderParams &params, double *d_rgb, uint64_t *d_stats,
                  unsigned *d_work, unsigned n_cu, bool collect_stats, void *stream, uint32_t *launch_id_out,
                  bool refuse_oversize);

// the wavefront schedule: a host-driven sequence of launches, synchronous (rtc_wavefront.inc)
int launch_wavefront(const DScene &scene, const DRenderParams &params, double *d_rgb, const WfBuffers &b,
                     unsigned *d_work, unsigned n_cu, void *stream, uint32_t *launch_id_out, uint32_t hsize,
                     uint32_t *generations_out);

// launch ids: unique per process (scenes on one GPU may share a work buffer), never 0
uint32_t next_launch_id();

// tiling helpers shared by the API and the launcher
uint32_t local_rows(uint32_t vsize, uint32_t tile_rows, uint32_t tile_index, uint32_t tile_count);

}  // namespace rtc
