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
    enum Kind { GROUP, SHAPE, TRIANGLES } kind;
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
    std::vector<DBvhNode> bvh;     // f32 nodes; bvh_q holds the same tree quantised when quantized_nodes
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
    uint32_t bvh_max_depth = 0, bvh_max_leaf = 0;
    uint32_t n_transparent_prims = 0;
    int32_t root_ref = 0;
    double root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};

    // ---- device state ----
    DScene dscene{};
    std::vector<void *> device_allocs;
    uint64_t device_bytes = 0;
    uint64_t *d_stats = nullptr;  // ST_COUNT counters
    unsigned *d_work = nullptr;   // tile counters of the persistent render grid (8 x 128 B, zeroed per launch)
    unsigned max_waves = 0;       // CU count of the device (persistent grid = CUs x resident waves per CU)
    void *ev_start = nullptr, *ev_stop = nullptr;
    rtc_stats last_stats{};
    bool have_last_stats = false;

    ~Scene();
    int add_item(Item &&it);
    int commit(int device);
    int flatten();
    void build_bvh();
    int upload(int device);
    void release_device();
};

// kernels (rtc_kernels.hip)
int launch_render(const DScene &scene, const DRenderParams &params, double *d_rgb, uint64_t *d_stats,
                  unsigned *d_work, unsigned n_cu, bool collect_stats, void *stream);

// tiling helpers shared by the API and the launcher
uint32_t local_rows(uint32_t vsize, uint32_t tile_rows, uint32_t tile_index, uint32_t tile_count);

}  // namespace rtc
