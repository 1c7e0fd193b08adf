// rtc_types.h -- POD tables shared by the host flattener and the HIP kernels.
//
// Data layout in HBM (all tables are immutable after rtc_scene_commit, one copy per GPU):
//   DXform[]    rows 0..2 of a cofactor inverse (96 B each); one per Group / analytic shape
//   int chains[] root-first lists of DXform indices (a primitive's parent chain + its own transform)
//   DPrim[]     one per leaf shape in depth-first (== reference traversal) order, 72 B
//   DTri[]      triangle payloads (p1, e1, e2, n1..n3), 144 B
//   DMaterial[] 96 B, DLight[] 136 B, DPattern[] 160 B
//   DBvhNode[] / DBvhNodeQ[]  64 B (f32) or 32 B (16-bit quantised) internal nodes of the binary BVH over the
//               bounded primitives (breadth-first order), each holding BOTH children's boxes
//   DLeaf[]     intersection-loop records: unbounded primitives (tested for every ray) followed by the BVH
//               leaves' primitive runs; a second copy in primitive order serves brute-force mode
#pragma once
#include <stdint.h>

namespace rtc {

enum PrimType : int32_t {
    PT_SPHERE = 0,
    PT_PLANE = 1,
    PT_CUBE = 2,
    PT_CYLINDER = 3,
    PT_CONE = 4,
    PT_TRIANGLE = 5,
    PT_SMOOTH_TRIANGLE = 6,
    PT_CSG = 7  // pseudo-primitive: one whole Csg tree (shape.rs:2146-2291) as the traversal sees it
};

// Csg (shape.rs:2146-2291).  A Csg tree is one entry of the intersection structures (always-tested list or BVH leaf):
// filter_intersections needs the records of its leaves in sorted order, which the streaming fold of the rest of the
// path does not keep, so the kernel collects them into a small per-lane array, sorts, runs the filters of the tree's
// Csg nodes bottom-up and feeds what survives to the fold.  The leaves of a tree are consecutive primitives (depth-
// first order); their DLeaf records carry RTC_LEAF_IN_CSG so that the lists skip them.
struct DCsgRoot {
    int32_t prim_first, n_prims;  // the tree's leaf primitives
    int32_t node_first, n_nodes;  // its Csg nodes in post-order (children before parents; the last one is the root)
};
struct DCsgNode {
    int32_t op;                 // rtc_csg_op
    int32_t l_first, l_count;   // leaves of the left / right operand, relative to DCsgRoot::prim_first
    int32_t r_first, r_count;
};
#define RTC_LEAF_IN_CSG 2    // bit of DLeaf::closed
#define RTC_CSG_MAX_RECORDS 32  // records one ray may collect from one Csg tree (more: overflow flag)

struct DXform {
    double inv[12];  // inverse(T) rows 0..2, row-major 3x4
};

struct DPrim {
    int32_t type;
    int32_t material;
    int32_t chain_off;  // into chains[]; root-first; for triangles the chain holds the ancestors only
    int32_t chain_len;
    double minimum, maximum;
    int32_t closed;
    int32_t tri;        // index into DTri[] or -1
    int32_t own_xform;  // DXform index of the shape's OWN transform (pattern lookup, Q13) or -1 (triangles)
    int32_t chain_id;   // primitives with equal chain_id share the same chain (lane-local ray cache key)
    // PT_PLANE only: Shape::normal_at of a plane does not depend on the point (shape.rs:550-553), so the whole
    // normal_to_world chain (shape.rs:94-103) is evaluated once on the host with the same operations
    double plane_normal[3];
};

// Intersection-loop record (128 B), one per primitive and per list position: everything a ray/primitive test
// needs sits in ONE record, so a test costs a single memory round trip (the uniform list is read through the
// scalar cache, BVH leaves through the vector cache).  `a` holds rows 0..2 of inverse(T_own) for analytic
// shapes and p1,e1,e2 for triangles (whose own transform is the identity and is skipped, shape.rs:1505-1509).
// anc_* describe the ANCESTOR (group) transforms only, applied root first before `a`.
struct DLeaf {  // 128 B, 128-byte aligned: one test = one cache line
    uint8_t type;
    uint8_t closed;
    uint16_t anc_len;  // number of enclosing groups
    int32_t prim;      // flattened (depth-first) primitive index: identity, tie-break order, shading lookup
    int32_t anc_off;   // into chains[]
    int32_t anc_id;    // equal ids <=> same ancestor chain (ray-in-group-space cache key); -1 when anc_len == 0
    double minimum, maximum;
    double a[12];
};
static_assert(sizeof(DLeaf) == 128, "DLeaf must fill exactly one 128-byte line");

// Compact BVH-leaf record (64 B) for the commonest primitive of big scenes: a sphere outside any group whose
// inverse transform is diagonal + translation (translation * scaling).  The six off-diagonal entries are +-0:
// only their signs are kept, the kernel rebuilds the 3x4 matrix and runs the SAME arithmetic as for a DLeaf, so
// results are bit-identical; what is saved is half of the leaf traffic (leaf records are what misses in L2).
// Leaf runs made of such spheres only are stored in DScene::leaves_s; RTC_LEAF_COMPACT marks their references.
struct DLeafS {
    int32_t prim;
    uint32_t zsign;  // bit k: off-diagonal entry k (row-major order a01 a02 a10 a12 a20 a21) is -0.0
    double d[3];     // a00 a11 a22
    double t[3];     // a03 a13 a23
    uint32_t pad[2];
};
static_assert(sizeof(DLeafS) == 64, "DLeafS is half a cache line");
#define RTC_LEAF_COMPACT (1 << 27)  // in the `first` field of a leaf reference / the kernel's leaf cursor

struct DTri {
    double p1[3], e1[3], e2[3];
    double n1[3], n2[3], n3[3];  // flat triangles: n1 = precomputed face normal
};

struct DMaterial {
    double color[3];
    double ambient, diffuse, specular, shininess;
    double reflectivity, transparency, refractive_index;
    int32_t casts_shadow;
    int32_t pattern;
    // reflect.dot(eye) below which pow(rde, shininess) < 2^-66: the specular term is then too small to change
    // `ambient + specular` in f64 and need not be evaluated (see lighting()); 0 = never skip
    double spec_skip_rde;
};

struct DLight {
    int32_t is_area;
    uint32_t usteps, vsteps, samples;
    double position[3], intensity[3];
    double corner[3], uvec[3], vvec[3];
};

struct DPattern {
    int32_t kind;
    int32_t sub_a, sub_b;
    int32_t pad;
    double a[3], b[3];
    double inv[12];
};

// One per INTERNAL node of the binary BVH: both children's padded world-space boxes live in the parent, so a
// traversal step costs a single fetch.  The boxes are culling volumes of this build, not reference arithmetic; the
// slab test runs in f64 on the decoded planes.  Two encodings, chosen per scene at commit:
//   DBvhNode  (64 B) f32 planes rounded OUTWARD -- small hierarchies (they stay in L2 anyway; cheapest decode);
//   DBvhNodeQ (32 B) 16-bit cell indices in ONE frame for the whole hierarchy (DScene::q_org / q_cell: the root box
//             cut into 65535 cells per axis), rounded OUTWARD -- large hierarchies: incoherent rays make traversal
//             a chain of dependent cache misses whose cost is set by how much of the tree stays in the XCD's 4 MB
//             L2 (stress scene: 6.6 MB -> 3.3 MB of nodes, 3.73 s -> 3.13 s per 4K frame).
// Child references: >= 0 is the index of another internal node; < 0 encodes a leaf as
// ~((first << 3) | (count - 1)) with `first` an offset into DScene::leaves (or RTC_LEAF_COMPACT | offset into
// DScene::leaves_s).  The "paired" flags of the children
// (see Scene::build_bvh) live in DScene::node_flags, read by the containers pass only.
#ifndef RTC_QF32
#define RTC_QF32 0  // A/B: large hierarchies keep 64-byte f32 nodes (no conversions in the box test, twice the node bytes)
#endif
struct DBvhNode {
    // {min, max} per axis, left child then right child: each pair arrives in an even-aligned register pair, which is what
    // the slab test's packed FMA takes (with min[3], max[3] the compiler spent ten moves per visit on pairing them up)
    float lx[2], ly[2], lz[2];
    float rx[2], ry[2], rz[2];
    int32_t left, right;
    int32_t pad[2];
};
struct DBvhNodeQ {
    uint16_t lmin[3], lmax[3];
    uint16_t rmin[3], rmax[3];
    int32_t left, right;
};
// Large hierarchies, round 2: FOUR children per node (the binary SAH tree with every other level folded into its
// parent), 64 B = one fetch per TWO levels of the binary tree.  Traversal of a large scene is a chain of dependent
// cache misses (waves wait on memory for more than half of their life with 3 waves per SIMD): halving the length of
// that chain is worth more than the few extra box tests of a wider node now that a box test is 20 f32 instructions.
//   box[3c .. 3c+2]  child c: min.x | min.y << 16,  min.z | max.x << 16,  max.y | max.z << 16   (cell indices, outward)
//   child[c]         as above; an unused slot holds RTC_NO_CHILD and an inverted box
#ifndef RTC_BVH4
#define RTC_BVH4 0  // 1: large hierarchies are traversed four wide (DBvhNode4), 0: binary (DBvhNodeQ)
#endif
struct DBvhNode4 {
    uint32_t box[12];
    int32_t child[4];
};
static_assert(sizeof(DBvhNode4) == 64, "DBvhNode4 is one 64-byte fetch");
#define RTC_NO_CHILD ((int32_t)0x80000000)

struct DCamera {
    double inv[12];
    double half_width, half_height, pixel_size;
    uint32_t hsize, vsize;
};

// Everything a kernel needs, passed by value as the kernel argument.
struct DScene {
    // The struct is the kernel argument: ~100 SGPRs fetched in blocks of up to 16 registers, and what does not
    // fit is parked in VGPR lanes and brought back a whole block at a time.  The first 64 bytes are therefore
    // exactly what the traversal loop dereferences, so that this block stays resident.
    const DBvhNode4 *bvh4;      // quantized_nodes: bvh4 (RTC_BVH4) or bvh_q is set, else bvh
    const DBvhNodeQ *bvh_q;
    const uint8_t *node_flags;  // per node: bit c = the subtree of child c holds "paired" primitives only
    const DLeaf *leaves;        // [0, n_unbounded): primitives tested for every ray; then the BVH leaves' runs
    const DLeafS *leaves_s;     // compact records of sphere-only leaf runs (large hierarchies)
    const DBvhNode *bvh;
    const DXform *xforms;
    const int32_t *chains;
    const DTri *tris;
    // ---- shading ----
    const DPrim *prims;
    const DMaterial *materials;
    const DLight *lights;
    const DPattern *patterns;
    const DLeaf *leaves_all;    // DLeaf records in primitive order (brute-force mode)
    int32_t root_ref;           // child reference of the BVH root (valid when n_bounded > 0)
    int32_t n_prims, n_unbounded, n_bounded, n_bvh_nodes, n_lights, n_materials;
    int32_t all_cast_shadow;  // 1 if every material has casts_shadow: shadow rays may stop at any hit
    int32_t any_pattern;
    int32_t any_group;        // 1 if some primitive sits inside a group (ancestor transforms exist)
    int32_t quantized_nodes;  // 1: bvh4, 0: bvh
    int32_t pad0;
    double q_org[3], q_cell[3]; // box plane along axis a = q_org[a] + q * q_cell[a]
    double root_min[3], root_max[3];
    // The camera is read once per pixel: it stays in memory (32 fewer argument registers, which the loops would
    // otherwise pay for with v_readlane reloads of other arguments) and only its two sizes travel by value.
    const DCamera *cam;
    uint32_t hsize, vsize;
    // ---- rarely used (kept last so that the blocks above keep their places in the argument registers) ----
    const DCsgRoot *csg_roots;
    const DCsgNode *csg_nodes;
    int32_t any_csg;
    int32_t pad1;
};

// Work buffer of the persistent schedule (one per scene, device memory; the first RTC_WORK_RESET_BYTES are zeroed
// before every launch).  u32 indices unless noted:
//   [0, 256)            tile counters, RTC_SHARDS of them on 128-byte lines
//   RTC_WORK_BUSY       waves that still have rays of their own to trace
//   RTC_WORK_HUNGRY     waves waiting for rays of others (a hint: donors skip the search when it is 0)
//   RTC_WORK_HINT_TAIL  number of hints published so far; RTC_WORK_HINTS: ring of 256 wave ids that went hungry
//   RTC_WORK_FLAG       overflow flag: a lane that exceeds a per-lane bound (rays per pixel, pending refractions,
//                       traversal stack) stores DRenderParams::launch_id here; launch ids are unique per process, so
//                       the word needs no reset and the host compares it with the id of the launch it asks about
//   RTC_WORK_STATES     per wave: mailbox state (RTC_MB_*);  RTC_WORK_COUNTS: per wave: rays in its mailbox
//   byte RTC_WORK_MBOX_BYTES on: per wave one mailbox of 64 rays x RTC_MBOX_ITEM_U64 x 8 bytes
#define RTC_MAX_WAVES 4096u
#define RTC_WORK_BUSY 256u
#define RTC_WORK_HUNGRY 288u
#define RTC_WORK_HINT_TAIL 320u
#define RTC_WORK_FLAG 352u  // id of the most recent launch in which a lane exceeded one of the kernel's per-lane bounds
#define RTC_WORK_HINTS 512u
#define RTC_WORK_STATES 1024u
#define RTC_WORK_COUNTS (RTC_WORK_STATES + RTC_MAX_WAVES)
#define RTC_WORK_RESET_BYTES ((RTC_WORK_COUNTS + RTC_MAX_WAVES) * 4u)
#define RTC_WORK_MBOX_BYTES RTC_WORK_RESET_BYTES
#define RTC_MBOX_ITEM_U64 10u
#define RTC_WORK_RING_OFFSET ((size_t)RTC_WORK_MBOX_BYTES + (size_t)RTC_MAX_WAVES * 64u * RTC_MBOX_ITEM_U64 * 8u)
//   byte RTC_WORK_RING_OFFSET on (RTC_RING_GLOBAL): per wave and lane the ring of waiting refracted rays, RTC_RING_SLOTS
//   entries of 64 bytes -- one cache line per entry, so that a push is one whole-line write and a pop one line read (as
//   a private array the ring lives in swizzled scratch memory: 15 dwords of an entry in 15 different 256-byte rows)
#ifndef RTC_RING_GLOBAL
#define RTC_RING_GLOBAL 1
#endif
#define RTC_RING_SLOTS 8u
//   and, behind each lane's ring, one more 64-byte line (RTC_SIDE_SLOT): the pixel's accumulated colour, the ray's weight
//   and the hit's Schlick factor -- the five values of the parked shading context that are touched least often; without
//   them a wave's LDS drops from 12.8 to 10 KB, i.e. from 12 to 16 waves per CU (the traversal is latency bound)
#ifndef RTC_SIDE_SLOT
#define RTC_SIDE_SLOT 1
#endif
#define RTC_LANE_LINES (RTC_RING_SLOTS + 1u)
#define RTC_WORK_TOTAL_BYTES (RTC_WORK_RING_OFFSET + (RTC_RING_GLOBAL ? (size_t)RTC_MAX_WAVES * 64u * RTC_LANE_LINES * 64u : 0u))
// Heavy tiles of the static schedule (they show a transparent object: a ray TREE per pixel, and refraction keeps the bounce
// budget) may be rendered in 2 or 4 PARTS of 32 / 16 pixels, one wave each; the wave's other lanes take over waiting
// refracted rays from the first round on, so the tile's critical path is nearer the depth of its ray trees than their
// size.  The heavy tiles are the first DRenderParams::n_heavy entries of the (sorted) tile list; the launcher picks the
// number of parts by how many such tiles there are for the device's resident waves (rtc_api.cpp, launch_on).
#define RTC_HEAVY_TILE_SCORE 1  // Scene::build_tile_order: a tile that shows anything reflective (1) or transparent (4)
#ifndef RTC_MIN_WAVES
#define RTC_MIN_WAVES 3  // waves per SIMD the register allocator must leave room for (static schedule)
#endif
enum { RTC_MB_IDLE = 0, RTC_MB_HUNGRY = 1, RTC_MB_FILLING = 2, RTC_MB_FULL = 3, RTC_MB_CLOSED = 4 };

struct DRenderParams {
    uint32_t depth;
    uint32_t tile_rows, tile_index, tile_count;
    uint32_t local_rows;  // rows owned by this rank (compact output)
    uint32_t use_bvh;     // resolved: 1 = BVH + unbounded list, 0 = brute force over all prims
    uint32_t tiles_x, tiles_y;  // 8x8 pixel tiles covering hsize x local_rows
    uint32_t persistent;        // 1: persistent grid + lane refill, 0: one workgroup per tile
    uint32_t dyn_base;          // first tile handed out through the device-wide counters (= grid size)
    uint32_t stay_min;          // waiting lanes at which a wave leaves the traversal loop to serve them (persistent)
    uint32_t refill_min;        // free lanes a wave waits for before it assigns new pixels (1..64)
    uint32_t share;             // 1: at the end of a frame waves without work take over waiting rays of other waves
    uint32_t n_waves;           // size of the persistent grid
    uint32_t tile_order;        // 0: tiles are handed out top row first, 1: bottom row first
    uint32_t leaf_den;          // primitive tests run when at least 1/leaf_den of the traversing lanes wait at a leaf
    uint32_t launch_id;         // process-wide unique, never 0: what an overflowing lane stores into work[RTC_WORK_FLAG]
    uint32_t max_rays;          // rays one lane may trace for one pixel (or adopted subtree) before it gives up
    uint32_t part_shift;        // 0, 1 or 2
    uint32_t refuse_oversize;   // host only: 1 = a grid beyond RTC_MAX_WAVES with the exchange on is an error (2 = it was)
    uint32_t n_heavy;           // static schedule: the first n_heavy tiles of the list are rendered in 1 << part_shift parts
    // static schedule, whole frame: workgroup b renders tile tile_perm[b] -- the tiles that show transparent or reflective
    // objects first, because a frame of few such objects lasts as long as its slowest tile and that tile should not
    // start last (null: workgroup b renders tile b)
    const uint32_t *tile_perm;
};

// ---- wavefront schedule (rtc_wavefront.inc): queue records ----
struct WfRay {  // 64 B: what one intersection pass needs
    double o[3], d[3];
    double lim;     // closest hit: +inf; shadow: distance to the light sample; containers: t of the hit
    int32_t prim;   // containers: the hit primitive; otherwise -1
    uint32_t mode;  // MODE_CLOSEST / MODE_SHADOW / MODE_CONTAINER
};
struct WfHit {  // 32 B: what the pass found (per mode: see k_traverse)
    double x0, x1, x2;
    int32_t i0, i1;
};
struct WfCtx {  // 144 B: the path context an item carries from stage to stage
    uint32_t pixel;     // index of the pixel in this launch's compact canvas
    int32_t remaining;  // bounce budget of the ray (world.rs:93-106: only reflection spends it)
    double w;           // what a colour found along this ray weighs in the pixel
    // the hit being shaded (valid from the closest-hit stage to the hit's last light):
    double point[3], eye[3], normal[3];
    double kr;          // Schlick reflectance (1 unless reflective and transparent)
    double surface[3];  // Material::lighting summed over the lights evaluated so far
    int32_t h_prim, h_mat;
    int32_t li;         // light in hand, and the area-light sample in hand
    uint32_t su, sv, unshadowed;
};
static_assert(sizeof(WfRay) == 64 && sizeof(WfHit) == 32 && sizeof(WfCtx) == 144, "queue records are read as whole lines");
#ifndef RTC_WF_MIN_WAVES
#define RTC_WF_MIN_WAVES 4  // waves per SIMD the traversal kernel's register allocation leaves room for
#endif
#ifndef RTC_WF_CLAIM
#define RTC_WF_CLAIM 64u    // items a wave claims from the queue per request
#endif

// device-side counters, one uint64 each (index constants)
enum StatIdx {
    ST_PRIMARY = 0,
    ST_SHADOW,
    ST_SECONDARY,
    ST_SECONDARY_REF,
    ST_CONTAINER,
    ST_BOX,
    ST_PRIM0,  // 7 entries
    ST_XFORM = ST_PRIM0 + 7,
    ST_SHADED,
    ST_LIGHT,
    ST_SCHLICK,
    ST_REFRACT,
    ST_OVERFLOW,
    ST_SHADOW_SKIPPED,  // is_shadowed queries answered without a trace: the light is behind the surface
    // diagnostics of the counting kernel: shader cycles (s_memtime) a wave spent per stage, summed over waves, and
    // how many lanes took part -- where the time of a divergent workload goes
    ST_CYC_REFILL,
    ST_CYC_START,     // (A) lockstep list + traversal set-up
    ST_CYC_TRAVERSE,  // (B)
    ST_CYC_SHADE,     // (C) state machine
    ST_ROUNDS_START, ST_LANES_START,
    ST_ROUNDS_TRAVERSE, ST_LANES_TRAVERSE,  // one round = one look at the wave (RTC_STEPS_PER_CHECK steps)
    ST_ROUNDS_SHADE, ST_LANES_SHADE,
    ST_COUNT
};

}  // namespace rtc
