/*
 * rtc_oracle.h -- CPU ORACLE for the per-pixel render path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm
 * (anna-hope/ray-tracer-challenge, crate `ray_tracer`): recursive color_at, brute-force
 * intersection of every shape, full sorted intersection lists, literal n1/n2 container loop.
 * It is NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The shipped library (librtc_amd.so) never links or calls it.
 *
 * Parity pinning: the reference cannot be built here (no Rust toolchain), so this oracle is
 * pinned by the reference's own known-answer unit tests, re-stated as data in
 * tests/golden/reference_known_answers.json and checked by tests/test_oracle_known_answers.py.
 * End-to-end images are not pinned by any reference fixture ("image parity unpinned" beyond
 * camera.rs:156-165).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef RTC_ORACLE_H
#define RTC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shape kinds -- ray_tracer/src/shape.rs:27-38 (Csg excluded) */
enum {
    ORC_SPHERE = 0,
    ORC_PLANE = 1,
    ORC_CUBE = 2,
    ORC_CYLINDER = 3,
    ORC_CONE = 4,
    ORC_TRIANGLE = 5,
    ORC_SMOOTH_TRIANGLE = 6,
    ORC_GROUP = 7,
    ORC_TEST_SHAPE = 8,
    ORC_CSG = 9 /* shape.rs:2146-2291; `closed` carries the CsgOp, the first two children are left and right */
};

/* pattern kinds -- ray_tracer/src/pattern.rs:5-15 (Perturbed excluded: un-vendored perlin_noise) */
enum {
    ORC_PAT_STRIPE = 0,
    ORC_PAT_GRADIENT = 1,
    ORC_PAT_RING = 2,
    ORC_PAT_CHECKER = 3,
    ORC_PAT_RADIAL = 4,
    ORC_PAT_SOLID = 5,
    ORC_PAT_BLENDED = 6,
    ORC_PAT_TEST = 7
};

enum { ORC_OK = 0, ORC_ERR_NON_INVERTIBLE = 1, ORC_ERR_INVALID = 2 };
/* CsgOp, shape.rs:2149-2154 */
enum { ORC_CSG_UNION = 0, ORC_CSG_INTERSECTION = 1, ORC_CSG_DIFFERENCE = 2 };

typedef struct orc_world orc_world;

/* material.rs:4-15; pattern < 0 means None */
typedef struct orc_material {
    double color[3];
    double ambient, diffuse, specular, shininess;
    double reflectivity, transparency, refractive_index;
    int32_t casts_shadow;
    int32_t pattern;
} orc_material;

typedef struct orc_stats {
    uint64_t primary_rays;
    uint64_t shadow_rays;
    uint64_t secondary_rays; /* reference-equivalent count: subtrees re-traced once per light */
    double seconds;
    uint64_t diverged_pixels; /* pixels whose ray tree nested deeper than the oracle's guard (rtc_oracle.c ORC_MAX_NESTING):
                                 the reference overflows its stack on them; they are NaN in the canvas */
} orc_stats;

/* ---- matrix / transform helpers (matrix.rs, transformation.rs); all 4x4 row-major ---- */
void orc_matrix_identity(double out[16]);
void orc_matrix_mul(const double a[16], const double b[16], double out[16]);
double orc_matrix_determinant(const double a[16]);
int orc_matrix_inverse(const double a[16], double out[16]);
void orc_matrix_translation(double x, double y, double z, double out[16]);
void orc_matrix_scaling(double x, double y, double z, double out[16]);
void orc_matrix_rotation_x(double r, double out[16]);
void orc_matrix_rotation_y(double r, double out[16]);
void orc_matrix_rotation_z(double r, double out[16]);
void orc_matrix_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double out[16]);
void orc_matrix_mul_point(const double m[16], const double p[3], double out[3]);
void orc_matrix_mul_vector(const double m[16], const double v[3], double out[3]);
void orc_view_transform(const double from[3], const double to[3], const double up[3], double out[16]);

/* ---- world construction ---- */
orc_world *orc_world_new(void);
void orc_world_free(orc_world *w);
void orc_world_set_depth(orc_world *w, int depth); /* test hook; default = the reference's 5 (camera.rs:80) */

/* returns material index */
int orc_add_material(orc_world *w, const orc_material *m);
/* a,b: colours for GRADIENT/RING/RADIAL/SOLID(a only); sub_a/sub_b: pattern indices for
 * STRIPE/CHECKER/BLENDED. Returns pattern index. */
int orc_add_pattern(orc_world *w, int kind, const double a[3], const double b[3], int sub_a, int sub_b,
                    const double transform[16]);
/* parent: index of a GROUP shape or -1 (top level object of the World). For ORC_GROUP,
 * `material` is the group's `material: Option<Material>` (shape.rs:1896; -1 = None).
 * Returns shape index (>=0) or -(error). */
int orc_add_shape(orc_world *w, int kind, int parent, const double transform[16], int material, double minimum,
                  double maximum, int closed);
/* normals == NULL -> Triangle (shape.rs:1459), else SmoothTriangle (shape.rs:1682); 3 doubles each */
int orc_add_triangle(orc_world *w, int parent, const double p1[3], const double p2[3], const double p3[3],
                     const double n1[3], const double n2[3], const double n3[3], int material);
int orc_add_point_light(orc_world *w, const double position[3], const double intensity[3]);
int orc_add_area_light(orc_world *w, const double corner[3], const double full_uvec[3], uint32_t usteps,
                       const double full_vvec[3], uint32_t vsteps, const double intensity[3]);
int orc_set_camera(orc_world *w, uint32_t hsize, uint32_t vsize, double field_of_view, const double view[16]);
int orc_shape_set_material(orc_world *w, int shape, int material);
/* Csg::new(operation, left, right) (shape.rs:2166-2184): the node is added first, left and right are then added with
 * the node as their parent (first child = left, second = right; a third child is refused).  Returns the shape index. */
int orc_add_csg(orc_world *w, int parent, int operation);
/* Csg::intersection_allowed (shape.rs:2186-2197) and Csg::filter_intersections (shape.rs:2203-2225) on an explicit
 * list: xs given as n (t, shape) records; writes the indices of the records that pass, returns their number */
int orc_csg_intersection_allowed(int operation, int left_hit, int in_left, int in_right);
int orc_csg_filter(orc_world *w, int csg, int n, const double *ts, const int *shapes, int *kept);
/* Shape::includes (shape.rs:136-138, 2023-2030, 2278-2280) */
int orc_shape_includes(orc_world *w, int shape, int object);

/* ---- the hot path ---- */
/* Camera::render (camera.rs:64-93): rgb_out is vsize*hsize*3 doubles, row-major, last row and
 * last column stay black.  nthreads rows-parallel workers (mirrors rayon over rows).
 * row_begin/row_end/col_step/row_step restrict the rendered pixel set (for bounded CPU-baseline
 * samples): pixels (x,y) with y in [row_begin,row_end) stepping row_step, x stepping col_step.
 * Pass 0,vsize,1,1 for the full frame. */
int orc_render(orc_world *w, int nthreads, double *rgb_out, orc_stats *stats);
int orc_render_subset(orc_world *w, int nthreads, uint32_t row_begin, uint32_t row_end, uint32_t row_step,
                      uint32_t col_step, double *rgb_out, orc_stats *stats);
/* The loop body of Camera::render (camera.rs:78-84) for an explicit list of n pixels, xy = n (x, y) pairs;
 * rgb_out = 3 doubles per listed pixel (black for the never-rendered last row / column). */
int orc_render_pixels(orc_world *w, int nthreads, size_t n, const uint32_t *xy, double *rgb_out, orc_stats *stats);

/* ---- unit-level entry points used by the known-answer tests ---- */
int orc_camera_params(orc_world *w, double out_hw_hh_ps[3]);
int orc_ray_for_pixel(orc_world *w, uint32_t px, uint32_t py, double origin[3], double direction[3]);
/* Shape::intersect (or World::intersect when shape < 0): writes up to cap (t, shape index, u, v) */
int orc_intersect(orc_world *w, int shape, const double origin[3], const double direction[3], int cap, double *ts,
                  int *shapes, double *us, double *vs);
/* local_intersect: ray given in object space */
int orc_local_intersect(orc_world *w, int shape, const double origin[3], const double direction[3], int cap,
                        double *ts);
int orc_local_normal_at(orc_world *w, int shape, const double p[3], double u, double v, double out[3]);
int orc_normal_at(orc_world *w, int shape, const double p[3], double u, double v, double out[3]);
int orc_world_to_object(orc_world *w, int shape, const double p[3], double out[3]);
int orc_normal_to_world(orc_world *w, int shape, const double n[3], double out[3]);
int orc_color_at(orc_world *w, const double origin[3], const double direction[3], int remaining, double rgb[3]);
int orc_is_shadowed(orc_world *w, const double point[3], const double light_position[3], int *out);
int orc_light_intensity_at(orc_world *w, int light, const double point[3], double *out);
int orc_area_light_point(orc_world *w, int light, uint32_t u, uint32_t v, double out[3]);
int orc_light_info(orc_world *w, int light, double position[3], double uvec[3], double vvec[3], uint32_t *samples);
/* Material::lighting (material.rs:20-75) */
int orc_lighting(orc_world *w, int material, int shape, int light, const double point[3], const double eye[3],
                 const double normal[3], double intensity, double rgb[3]);
int orc_pattern_at(orc_world *w, int pattern, const double p[3], double rgb[3]);
int orc_pattern_at_shape(orc_world *w, int pattern, int shape, const double p[3], double rgb[3]);

/* prepare_computations over an explicit intersection list (intersection.rs:123-194):
 * xs given as n (t, shape, u, v) records, hit_index selects `self`.
 * comps_out layout (27 doubles): t, point[3], eye[3], normal[3], inside, over[3], under[3],
 * reflect[3], n1, n2, shape, u, v (n1/n2 are NaN when None). */
#define ORC_COMPS_LEN 27
int orc_prepare_computations(orc_world *w, const double origin[3], const double direction[3], int n,
                             const double *ts, const int *shapes, const double *us, const double *vs,
                             int hit_index, double *comps_out);
int orc_hit_index(int n, const double *ts);
int orc_shade_hit(orc_world *w, const double *comps, int remaining, double rgb[3]);
int orc_reflected_color(orc_world *w, const double *comps, int remaining, double rgb[3]);
int orc_refracted_color(orc_world *w, const double *comps, int remaining, double rgb[3]);
double orc_schlick(const double *comps);

/* canvas.rs:86-89 */
uint8_t orc_scale_and_clamp(double v);
/* canvas.rs:44-82; returns bytes needed (excluding NUL); writes at most cap bytes */
size_t orc_canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
