/*
 * rtc.h -- C ABI of librtc_amd.so: the MI355X (gfx950) implementation of the per-pixel render path
 * of anna-hope/ray-tracer-challenge (Camera::render -> World::color_at -> intersect ->
 * prepare_computations -> shade_hit -> reflected_color / refracted_color).
 *
 * The reference boundary is one Rust method, `Camera::render(&self, &World) -> Result<Canvas>`
 * (ray_tracer/src/camera.rs:64), fed by `World::new(objects, lights)` (world.rs:19-21).  This header
 * is what a Rust `extern "C"` block binds instead: plain pointers and sizes,
 * all floating point is IEEE f64, matrices are 4x4 row-major, colours are RGB triples.
 *
 * Conventions
 *   - every function returns RTC_OK (0) or an RTC_ERR_* code; rtc_last_error() gives the text
 *     (thread-local).  The only error the reference path itself can raise is NonInvertibleMatrix
 *     (error.rs:3-7) -> RTC_ERR_NON_INVERTIBLE.
 *   - one scene per handle; a handle is not thread-safe, different handles are independent.
 *   - the render entry points run on the GPU only: there is no CPU fallback; without a usable HIP
 *     device they fail with RTC_ERR_NO_DEVICE.
 *
 * Citations are `path:line` relative to the reference root.
 */
#ifndef RTC_H
#define RTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtc_scene rtc_scene;

enum rtc_status {
    RTC_OK = 0,
    RTC_ERR_NON_INVERTIBLE = 1, /* RayTracerError::NonInvertibleMatrix, error.rs:3-7 */
    RTC_ERR_INVALID = 2,        /* bad argument / index / call order */
    RTC_ERR_NO_DEVICE = 3,      /* no usable HIP device (the product never renders on the CPU) */
    RTC_ERR_HIP = 4,            /* a HIP runtime call failed */
    RTC_ERR_STATE = 5,          /* scene not committed / already committed */
    RTC_ERR_IO = 6,             /* file could not be read / written */
    RTC_ERR_PARSE = 7,          /* YAML / OBJ syntax or semantic error (parse_scene ParseError) */
    RTC_ERR_OVERFLOW = 8        /* a per-lane bound of the kernel was exceeded: the frame is not valid (see rtc_render) */
};

/* ShapeType, shape.rs:27-38.  Triangles are added with rtc_scene_add_triangles, groups with
 * rtc_scene_begin_group/end_group, Csg nodes with rtc_scene_begin_csg/end_csg. */
enum rtc_shape_type { RTC_SPHERE = 0, RTC_PLANE = 1, RTC_CUBE = 2, RTC_CYLINDER = 3, RTC_CONE = 4 };
/* CsgOp, shape.rs:2149-2154 */
enum rtc_csg_op { RTC_CSG_UNION = 0, RTC_CSG_INTERSECTION = 1, RTC_CSG_DIFFERENCE = 2 };

/* PatternType, pattern.rs:5-15 (Perturbed excluded: arithmetic lives in the un-vendored perlin_noise crate) */
enum rtc_pattern_type {
    RTC_PAT_STRIPE = 0,
    RTC_PAT_GRADIENT = 1,
    RTC_PAT_RING = 2,
    RTC_PAT_CHECKER = 3,
    RTC_PAT_RADIAL = 4,
    RTC_PAT_SOLID = 5,
    RTC_PAT_BLENDED = 6,
    RTC_PAT_TEST = 7 /* pattern.rs:652-676: colour = point; kept for the reference's own known-answer test */
};

/* Material, material.rs:4-15; defaults material.rs:83-96 (rtc_material_default) */
typedef struct rtc_material {
    double color[3];
    double ambient, diffuse, specular, shininess;
    double reflectivity, transparency, refractive_index;
    int32_t casts_shadow;
    int32_t pattern; /* index from rtc_scene_add_pattern, or -1 for None */
} rtc_material;

/* Render options.  Zero-initialise and call rtc_render_opts_default(). */
typedef struct rtc_render_opts {
    uint32_t depth;      /* bounce budget; the reference uses the literal 5 (camera.rs:80).  At most 8: the per-lane
                            stack of waiting refracted rays holds one entry per distinct `remaining` value (a larger
                            value is refused with RTC_ERR_INVALID) */
    uint32_t tile_rows;  /* rows per row tile for multi-GPU sharding (0 = whole image is one tile); must be set
                            (non-zero) whenever tile_count > 1 */
    uint32_t tile_index; /* this rank renders tiles t with t % tile_count == tile_index */
    uint32_t tile_count; /* number of ranks (1 = everything) */
    uint32_t use_bvh;    /* 0 = auto, 1 = force BVH traversal, 2 = force brute-force linear scan */
    uint32_t collect_stats; /* 1 = run the counting variant of the kernel and fill the ray/test counters */
    uint32_t schedule;   /* 0 = auto (1 or 2 by scene), 1 = one workgroup per 8x8 tile, 2 = persistent grid with lane
                            refill, 3 = wavefront (experimental, never chosen by auto: queues of rays in device memory,
                            one traversal kernel and one shading kernel per kind of ray and generation; same pixels
                            within ~1e-14; the call is synchronous, allocates its queues at the first use -- 5.4 KB per
                            pixel -- and is refused by rtc_render_device_static and with collect_stats) */
    uint32_t exchange;   /* persistent schedule only.  0 = auto (on): at the end of a frame lanes and waves without work
                            take over waiting rays of busy ones and ADD their part to the pixel with f64 atomics -- the
                            order of those few additions is timing dependent (observed <= 2e-15).  1 = off: every pixel
                            is traced by one lane from start to end; the frame is bit-reproducible and bit-equal to
                            schedule 1's */
    uint32_t max_rays_per_pixel; /* guard against runaway ray trees (refraction keeps the bounce budget, world.rs:142):
                            a lane that traces more rays than this for one pixel gives up, the pixel becomes NaN and the
                            call reports RTC_ERR_OVERFLOW.  0 = default (65536).  With the ray exchange (schedule 2,
                            exchange != 1) every lane that takes over part of a pixel's ray tree counts from 1 again, so
                            the cap bounds the work of one lane, not the size of the tree */
    uint32_t fp_mode;    /* must be 0: strict -- no FMA contraction, IEEE division and sqrt, the reference's operation
                            order.  (Rounds 1-2 offered 1 = contraction and shared reciprocals: 2 % faster, but a ray that
                            grazes an edge landed on its other side and glass / mirrors carried that on, up to 95/255 in
                            single pixels; it did not meet the parity bar and was removed.  Any other value is refused
                            with RTC_ERR_INVALID.) */
    uint32_t tail_polls; /* persistent schedule with the exchange on.  0 = a wave of the grid that has run out of work stays
                            resident until the frame ends and takes over waiting rays of others: the shortest time for ONE
                            frame.  N > 0 = such a wave looks N times (every ~5 us) for rays of others and then leaves,
                            freeing its slot: for callers that keep two or more frames in flight on different streams
                            (rtc_render_device with one scene handle -- its own work buffer -- and one stream per frame
                            in flight): the grid of the next frame moves in while this frame's last ray trees finish, so
                            the end of a frame -- a chain of dependent rays that does not shrink with the pixel count --
                            costs throughput no longer (a frame alone gets slower with N > 0).
                            Same pixels as with 0 (to the order of the additions of split pixels) */
    uint32_t reserved[5];
} rtc_render_opts;

/* Work counters (filled when collect_stats != 0) and timing of the last launch. */
typedef struct rtc_stats {
    uint64_t pixels;          /* active pixels rendered by this call: (hsize-1) x rows */
    uint64_t primary_rays;    /* one per active pixel */
    uint64_t shadow_rays;     /* one per (shaded hit x light sample) */
    uint64_t secondary_rays;  /* reflect + refract rays actually traced ("rays_unique") */
    uint64_t secondary_rays_reference; /* what the reference traces: subtrees re-traced once per light (Q3) */
    uint64_t container_passes; /* n1/n2 line passes (not rays) */
    uint64_t box_tests;
    uint64_t prim_tests[7];   /* sphere, plane, cube, cylinder, cone, triangle, smooth triangle */
    uint64_t xform_levels;    /* ray -> object-space transforms applied (33 flops each) */
    uint64_t shaded_hits;
    uint64_t light_evals;     /* Material::lighting evaluations */
    uint64_t schlick_evals;
    uint64_t refract_evals;
    uint64_t overflow;        /* lanes that hit the iteration cap (0 in a valid render) */
    uint64_t shadow_rays_skipped; /* is_shadowed queries whose answer cannot change the pixel (the light is behind the
                                 surface: material.rs:49-52 sets diffuse and specular to black whatever the intensity)
                                 and that were therefore not traced; the reference traces shadow_rays + this many */
    uint64_t diag[10];        /* counting kernel only: shader cycles per stage summed over waves {refill, start, traverse,
                                 shade}, then {rounds, lanes} of start / traverse / shade */
    double kernel_ms;         /* HIP-event time of the render kernel on its stream (rtc_render only) */
    double algorithmic_flops; /* SURVEY.md 8(d) flop table x the counters above */
} rtc_stats;

const char *rtc_last_error(void);
const char *rtc_version(void);
void rtc_material_default(rtc_material *out);   /* Material::default(), material.rs:83-96 */
void rtc_render_opts_default(rtc_render_opts *out);

/* ---- host math, bit-faithful to the reference (used by loaders / callers that build transforms) ----
 * matrix.rs:181-266 (builders), :302-326 (multiply), :74-179 (determinant / cofactor inverse),
 * transformation.rs:3-15 (view transform).  `out` may alias an input. */
void rtc_matrix_identity(double out[16]);
void rtc_matrix_translation(double x, double y, double z, double out[16]);
void rtc_matrix_scaling(double x, double y, double z, double out[16]);
void rtc_matrix_rotation_x(double radians, double out[16]);
void rtc_matrix_rotation_y(double radians, double out[16]);
void rtc_matrix_rotation_z(double radians, double out[16]);
void rtc_matrix_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double out[16]);
void rtc_matrix_multiply(const double a[16], const double b[16], double out[16]);
int rtc_matrix_inverse(const double a[16], double out[16]); /* RTC_ERR_NON_INVERTIBLE iff det == 0.0 */
void rtc_view_transform(const double from[3], const double to[3], const double up[3], double out[16]);

/* ---- scene construction: replaces World::new + the Shape constructors ---- */
rtc_scene *rtc_scene_create(void);
void rtc_scene_destroy(rtc_scene *scene);

/* Camera::new(hsize, vsize, field_of_view, transformation), camera.rs:17-37 */
int rtc_scene_set_camera(rtc_scene *scene, uint32_t hsize, uint32_t vsize, double field_of_view,
                         const double view[16]);
/* PointLight::new, light.rs:33-38 */
int rtc_scene_add_point_light(rtc_scene *scene, const double position[3], const double intensity[3]);
/* AreaLight::new, light.rs:80-104 */
int rtc_scene_add_area_light(rtc_scene *scene, const double corner[3], const double full_uvec[3], uint32_t usteps,
                             const double full_vvec[3], uint32_t vsteps, const double intensity[3]);
int rtc_scene_add_material(rtc_scene *scene, const rtc_material *material, uint32_t *out_id);
/* Pattern constructors, pattern.rs:80-87, 212-219, 286-293, 354-361, 437-444, 508-511, 568-575.
 * a/b: colours (GRADIENT/RING/RADIAL, SOLID uses a); sub_a/sub_b: pattern ids (STRIPE/CHECKER/BLENDED). */
int rtc_scene_add_pattern(rtc_scene *scene, int type, const double a[3], const double b[3], int32_t sub_a,
                          int32_t sub_b, const double transform[16], uint32_t *out_id);
/* Group::default().with_transformation(T) [.with_material(m)], shape.rs:1900-1928; children are whatever is
 * added until the matching rtc_scene_end_group (Group::add_child, shape.rs:1951-1958).  material_id < 0 means
 * `material: None`; the reference's own constructors always carry Some(..) (shape.rs:1909), which only
 * triangles consult (shape.rs:1520-1532). */
int rtc_scene_begin_group(rtc_scene *scene, const double transform[16], int32_t material_id);
int rtc_scene_end_group(rtc_scene *scene);
/* Csg::new(operation, left, right), shape.rs:2166-2184: the first object added after rtc_scene_begin_csg is `left`,
 * the second `right` (analytic shapes, groups or other Csg nodes); rtc_scene_end_csg closes the node and fails with
 * RTC_ERR_STATE unless exactly two were added.  Like the reference's Csg the node has no transformation and no
 * material of its own.  Triangles cannot be direct children (Triangle::material() panics under a Csg parent,
 * shape.rs:1521-1523); put them in a group.  Equal-t records of the two operands are ordered by depth-first shape
 * index (the reference sorts them with sort_unstable_by, shape.rs:2232: unspecified).
 * Limits the reference does not have: a tree is ONE entry of the intersection structures, so a ray that reaches the
 * tree's bounding box tests every leaf shape of the tree (no hierarchy inside a tree: cost linear in its leaves), and one
 * ray may collect at most 32 intersection records from one tree (a Csg of more than ~16 convex shapes along one ray):
 * beyond that the launch reports RTC_ERR_OVERFLOW. */
int rtc_scene_begin_csg(rtc_scene *scene, int operation);
int rtc_scene_end_csg(rtc_scene *scene);
/* Sphere/Plane/Cube/Cylinder/Cone::new, shape.rs:179, 498, 653, 873, 1190.  minimum/maximum/closed are used
 * by cylinders and cones only. */
int rtc_scene_add_shape(rtc_scene *scene, int type, const double transform[16], uint32_t material_id,
                        double minimum, double maximum, int closed);
/* Triangle::new (normals == NULL, shape.rs:1459) or SmoothTriangle::new (shape.rs:1682): n triangles,
 * points = 9n doubles (p1,p2,p3), normals = 9n doubles (n1,n2,n3).  material_id is the triangles' own
 * material, used only when the parent group has none. */
int rtc_scene_add_triangles(rtc_scene *scene, size_t n, const double *points, const double *normals,
                            uint32_t material_id);

/* Freeze the scene: cofactor inverses (matrix.rs:157-179), depth-first flattening of the shape tree into SoA
 * device tables, host-built BVH, upload to HIP device `device` (>= 0).  device == -1 performs the host-side
 * work only (for CPU-side inspection; such a scene cannot render).  Errors: RTC_ERR_NON_INVERTIBLE if any
 * camera/shape/group/pattern transform has det == 0.0 (the reference raises it lazily on the first ray that
 * touches the matrix, matrix.rs:157-161; every shape is touched by every ray). */
int rtc_scene_commit(rtc_scene *scene, int device);
/* The same for a single process that drives several GPUs (the reference's counterpart is the rayon
 * loop over rows, camera.rs:72-85, plus the serial copy into the Canvas, :87-91): the scene is replicated on every
 * listed device; rtc_render then shards a frame by interleaved 8-row tiles (tile t -> devices[t % n]), assembles the
 * canvas on devices[0] with one peer copy per tile over xGMI and hands it to the host once.  rtc_render_device,
 * rtc_render_device_static and explicit tile options keep using devices[0] only.  A device may be listed more than
 * once (a rehearsal of the sharded path on a one-GPU box). */
int rtc_scene_commit_devices(rtc_scene *scene, const int *devices, uint32_t n_devices);

/* ---- the hot path: replaces Camera::render (camera.rs:64-93) ----
 * rgb_out: vsize*hsize*3 doubles, row-major canvas; like the reference, the last row and the last column
 * are never rendered and stay black (camera.rs:72,78).  Synchronous.  With tiling options only this rank's
 * rows are written (the rest of rgb_out is zeroed). */
int rtc_render(rtc_scene *scene, const rtc_render_opts *opts, double *rgb_out, rtc_stats *stats);
/* Errors of the render entry points beyond argument checks: RTC_ERR_OVERFLOW when a lane exceeded one of the
 * kernel's per-lane bounds (max_rays_per_pixel; more than 8 waiting refracted rays; a traversal stack deeper than 96
 * entries; more than 32 records of one ray in one Csg tree; a full queue of the wavefront schedule): rtc_render
 * reports it itself; after the asynchronous rtc_render_device ask rtc_scene_check().  The canvas is still written,
 * but the pixels concerned are UNDEFINED: NaN where max_rays_per_pixel was exceeded, otherwise a finite colour that
 * lacks the contributions that were dropped -- discard the frame.  The calls leave the caller's current HIP device as
 * they found it.
 * Scenes the reference does not finish: refraction keeps the bounce budget (world.rs:142), so a ray caught between
 * refracting surfaces (the apex of a transparent double cone, a transparent smooth triangle whose normals disagree with
 * its face) recurses until the reference overflows its stack.  Here such a ray ends at max_rays_per_pixel
 * (RTC_ERR_OVERFLOW, NaN in that pixel) or, earlier, where its weight -- the product of the transparencies on its way --
 * has underflowed to exactly 0.0; all other pixels of the frame are the reference's. */

/* Synchronises the scene's device and reports the state of the most recent launch on this scene: RTC_OK,
 * RTC_ERR_OVERFLOW (see above) or RTC_ERR_HIP (the kernel faulted). */
int rtc_scene_check(rtc_scene *scene);

/* Device-resident variant for callers that own device memory and a stream (torch, RCCL pipelines):
 * d_rgb is a DEVICE pointer to rtc_local_rows(...)*hsize*3 doubles; this rank's rows are stored compactly in
 * ascending canvas-row order (every row, including an all-black last canvas row if it is owned).
 * Asynchronous on `hip_stream` (a hipStream_t, NULL = default stream).  Whole-frame calls allocate nothing; the FIRST
 * call with a given share of the frame (tile_rows / tile_index / tile_count) uploads that share's tile order (one small
 * hipMalloc + synchronous copy into a buffer of its own, kept until the scene is destroyed: launches of other shares
 * that are still in flight are not disturbed), later calls with the same share allocate nothing. */
int rtc_render_device(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream);
/* rtc_render_device for callers that cannot allocate and cannot rely on memory the library allocated for this
 * scene's launches: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has NO
 * counterpart in the reference.  Same arguments, same pixels.  The kernel's work buffer (tile counters, overflow flag,
 * mailboxes of the end-of-frame ray exchange) is a file-scope `static __device__` array of the library's code object
 * instead of the per-scene buffer that rtc_scene_commit allocated: RTC_STATIC_SCRATCH_BYTES (172 MB: per wave of the
 * persistent grid a mailbox and, per lane, the ring of waiting refracted rays, for RTC_STATIC_SCRATCH_WAVES waves) per
 * GPU, held until the process ends.  Limits: collect_stats is refused; frames beyond 2^26 8x8 tiles are refused (as by
 * every entry point); the persistent grid is capped at RTC_STATIC_SCRATCH_WAVES resident waves (MI355X: 3072).  Every
 * scene on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race -- serialise
 * them.  The overflow flag lives in that array too and names the launch that raised it: ask rtc_scene_check(scene) BEFORE
 * the next static launch of ANY scene on the same GPU, or a later launch's overflow replaces the word and this one's
 * goes unreported.  The call itself makes no hipMalloc / hipFree / allocating runtime call (a share of the frame whose
 * tile order has not been uploaded by an earlier rtc_render / rtc_render_device call is rendered in raster order; the
 * first launch of a kernel variant in a process still loads its code object, as any first launch does). */
#define RTC_STATIC_SCRATCH_WAVES 4096u
#define RTC_STATIC_SCRATCH_BYTES (36864u + RTC_STATIC_SCRATCH_WAVES * 64u * (80u + 576u))
int rtc_render_device_static(rtc_scene *scene, const rtc_render_opts *opts, double *d_rgb, void *hip_stream);
/* Number of canvas rows owned by (tile_index, tile_count, tile_rows) and their canvas row numbers. */
uint32_t rtc_local_rows(uint32_t vsize, uint32_t tile_rows, uint32_t tile_index, uint32_t tile_count);
int rtc_local_row_map(uint32_t vsize, uint32_t tile_rows, uint32_t tile_index, uint32_t tile_count,
                      uint32_t *rows_out, uint32_t cap);
/* Counters of the most recent collect_stats launch on this scene (synchronises the device). */
int rtc_scene_last_stats(rtc_scene *scene, rtc_stats *stats);

/* ---- introspection of the committed scene (host side) ---- */
typedef struct rtc_scene_info {
    uint32_t hsize, vsize;
    uint32_t n_prims, n_unbounded, n_bvh_nodes, n_xforms, n_materials, n_lights, n_patterns, n_triangles;
    uint32_t bvh_max_depth, bvh_max_leaf;
    uint32_t committed, device;
    uint64_t device_bytes; /* all replicas */
    uint32_t n_devices;    /* replicas (rtc_scene_commit_devices) */
    uint32_t n_heavy_tiles; /* 8x8 tiles of the whole frame that show a transparent object (the static schedule may render
                              them in parts) */
} rtc_scene_info;
int rtc_scene_get_info(rtc_scene *scene, rtc_scene_info *out);

/* The builder tree as the reference would hold it (World.objects and Group::children order), for loaders'
 * callers and tests.  Items are numbered in insertion order. */
typedef struct rtc_item_info {
    int32_t kind;        /* 0 = group, 1 = analytic shape, 2 = run of triangles, 3 = Csg node */
    int32_t parent;      /* enclosing group's / Csg's item index, -1 = member of World.objects */
    int32_t shape_type;  /* rtc_shape_type (kind 1), rtc_csg_op (kind 3) */
    int32_t material_id; /* -1 = none */
    uint32_t n_children; /* kind 0 */
    uint32_t n_triangles; /* kind 2 */
    int32_t closed;
    int32_t smooth;      /* kind 2: SmoothTriangle run */
    double minimum, maximum;
    double transform[16];
} rtc_item_info;
typedef struct rtc_light_info {
    int32_t is_area;
    uint32_t usteps, vsteps;
    double position[3], intensity[3];
    double corner[3], uvec[3], vvec[3]; /* area lights: per-cell vectors (light.rs:80-104) */
} rtc_light_info;
uint32_t rtc_scene_item_count(rtc_scene *scene);
int rtc_scene_get_item(rtc_scene *scene, uint32_t item, rtc_item_info *out);
uint32_t rtc_scene_child_count(rtc_scene *scene, int32_t parent); /* parent = -1: World.objects */
int rtc_scene_get_child(rtc_scene *scene, int32_t parent, uint32_t k, uint32_t *out_item);
int rtc_scene_get_material(rtc_scene *scene, uint32_t material_id, rtc_material *out);
int rtc_scene_get_triangles(rtc_scene *scene, uint32_t item, uint32_t first, uint32_t n, double *points,
                            double *normals); /* 9 doubles per triangle each; normals may be NULL */
int rtc_scene_get_light(rtc_scene *scene, uint32_t index, rtc_light_info *out);
int rtc_scene_get_camera(rtc_scene *scene, uint32_t *hsize, uint32_t *vsize, double *field_of_view, double view[16]);

/* ---- callers either side of the path (host only) ----
 * parse_scene(&str) -> Scene (parse_scene/src/lib.rs:735-784): builds an UNCOMMITTED scene from YAML text.
 * base_dir resolves relative `obj` file names (may be NULL); width/height != 0 override the camera's size (the
 * reference has no such override: BASELINE.json's configs render the sample scenes at other resolutions).
 * Errors: RTC_ERR_PARSE (ParseError / ParserError, text in rtc_last_error()), RTC_ERR_IO. */
int rtc_scene_load_yaml(const char *yaml_text, const char *base_dir, uint32_t width, uint32_t height, rtc_scene **out);
int rtc_scene_load_yaml_file(const char *path, uint32_t width, uint32_t height, rtc_scene **out);
/* parse_obj_string(text, Some(material), Some(transformation)) (obj_file/src/lib.rs:40-230): adds the OBJ's default
 * group (and its named child groups) to the scene.  material_id < 0 = Material::default(). */
typedef struct rtc_obj_info {
    uint32_t ignored_lines, vertices, normals, triangles, groups;
} rtc_obj_info;
int rtc_scene_add_obj(rtc_scene *scene, const char *obj_text, const double transform[16], int32_t material_id,
                      rtc_obj_info *info);
/* Canvas::to_ppm (canvas.rs:44-89): P3 text of a vsize x hsize x 3 f64 canvas.  out == NULL queries the size. */
int rtc_canvas_to_ppm(const double *rgb, uint32_t hsize, uint32_t vsize, char *out, size_t cap, size_t *needed);

#ifdef __cplusplus
}
#endif
#endif
