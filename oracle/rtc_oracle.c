/*
 * rtc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see rtc_oracle.h).
 *
 * Plain-C restatement of the reference's per-pixel render path, operation for operation:
 *   camera.rs:17-93, world.rs:30-147,176-186, intersection.rs:20-39,66-90,123-219,
 *   shape.rs:43-103 + every local_intersect/local_normal_at, material.rs:20-96,
 *   light.rs:49-134, pattern.rs:25-32 + kinds, tuple.rs:96-125, matrix.rs:74-179,181-266,302-365,
 *   transformation.rs:3-15, canvas.rs:44-89.
 * The reference's quirks are reproduced on purpose.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (see oracle/Makefile): rustc never
 * contracts a*b+c into an FMA, so neither may this file.
 *
 * Unspecified behaviour pinned here (documented in DESIGN.md): the reference sorts intersections
 * with sort_unstable_by(total_cmp) (world.rs:183, shape.rs:1940), which leaves the order of
 * equal-t records of different objects unspecified; this oracle uses a STABLE sort, i.e. equal-t
 * records keep depth-first object order ("lower flattened index first").
 */
#define _GNU_SOURCE
#include "rtc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define EPSILON 1e-5 /* lib.rs:24 */

typedef struct {
    double x, y, z;
} v3;

typedef struct {
    double m[4][4];
} mat4;

/* ------------------------------------------------------------------ tuple.rs / color.rs */

static inline v3 v3_new(double x, double y, double z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 v_add(v3 a, v3 b) { return v3_new(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return v3_new(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, double s) { return v3_new(a.x * s, a.y * s, a.z * s); }
static inline v3 v_div(v3 a, double s) { return v3_new(a.x / s, a.y / s, a.z / s); }
static inline v3 v_neg(v3 a) { return v3_new(-a.x, -a.y, -a.z); }
static inline v3 c_mul(v3 a, v3 b) { return v3_new(a.x * b.x, a.y * b.y, a.z * b.z); } /* color.rs:83-93 */

/* lib.rs:26-29 */
static inline int equal(double a, double b) {
    double c = a - b;
    return fabs(c) < EPSILON;
}

/* tuple.rs:96-99: x.powi(2) + y.powi(2) + z.powi(2), sqrt */
static inline double v_magnitude(v3 a) {
    double sum = a.x * a.x + a.y * a.y + a.z * a.z;
    return sqrt(sum);
}
/* tuple.rs:101-108: three divisions */
static inline v3 v_norm(v3 a) {
    double m = v_magnitude(a);
    return v3_new(a.x / m, a.y / m, a.z / m);
}
/* tuple.rs:110-112 */
static inline double v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* tuple.rs:114-120 */
static inline v3 v_cross(v3 a, v3 b) {
    return v3_new(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* tuple.rs:122-124: self - normal * 2. * self.dot(normal)  ==  self - ((normal*2.)*dot) */
static inline v3 v_reflect(v3 self, v3 normal) {
    double d = v_dot(self, normal);
    return v_sub(self, v_mul(v_mul(normal, 2.), d));
}

/* f64::total_cmp key */
static inline int64_t total_key(double x) {
    int64_t b;
    memcpy(&b, &x, 8);
    b ^= (int64_t)(((uint64_t)(b >> 63)) >> 1);
    return b;
}

/* ------------------------------------------------------------------ matrix.rs */

static mat4 m_identity(void) {
    mat4 r;
    memset(&r, 0, sizeof r);
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0;
    return r;
}

/* matrix.rs:302-326 */
static mat4 m_mul(const mat4 *a, const mat4 *b) {
    mat4 r;
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++)
            r.m[row][col] = a->m[row][0] * b->m[0][col] + a->m[row][1] * b->m[1][col] +
                            a->m[row][2] * b->m[2][col] + a->m[row][3] * b->m[3][col];
    return r;
}

/* matrix.rs:328-346 */
static inline v3 m_mul_point(const mat4 *m, v3 p) {
    double r[3];
    for (int row = 0; row < 3; row++)
        r[row] = m->m[row][0] * p.x + m->m[row][1] * p.y + m->m[row][2] * p.z + m->m[row][3];
    return v3_new(r[0], r[1], r[2]);
}
/* matrix.rs:348-365 */
static inline v3 m_mul_vector(const mat4 *m, v3 v) {
    double r[3];
    for (int row = 0; row < 3; row++) r[row] = m->m[row][0] * v.x + m->m[row][1] * v.y + m->m[row][2] * v.z;
    return v3_new(r[0], r[1], r[2]);
}

/* matrix.rs:57-72 */
static mat4 m_transpose(const mat4 *a) {
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[j][i] = a->m[i][j];
    return r;
}

/* matrix.rs:86-108 (closed-form 4x4 determinant) */
static double m_det4(const mat4 *s) {
    double a = s->m[0][0], b = s->m[0][1], c = s->m[0][2], d = s->m[0][3];
    double e = s->m[1][0], f = s->m[1][1], g = s->m[1][2], h = s->m[1][3];
    double i = s->m[2][0], j = s->m[2][1], k = s->m[2][2], l = s->m[2][3];
    double m = s->m[3][0], n = s->m[3][1], o = s->m[3][2], p = s->m[3][3];

    double q1 = a * (f * (k * p - l * o) - g * (j * p - l * n) + h * (j * o - k * n));
    double q2 = b * (e * (k * p - l * o) - g * (i * p - l * m) + h * (i * o - k * m));
    double q3 = c * (e * (j * p - l * n) - f * (i * p - l * m) + h * (i * n - j * m));
    double q4 = d * (e * (j * o - k * n) - f * (i * o - k * m) + g * (i * n - j * m));
    return q1 - q2 + q3 - q4;
}

/* matrix.rs:74-81: 2x2 determinant of the submatrix of a 3x3 `s` without row 0 / column j,
 * then matrix.rs:110-117: det = 0.0; det += row0[j] * cofactor(0, j) */
static double m_det3(const double s[3][3]) {
    double det = 0.0;
    for (int j = 0; j < 3; j++) {
        double sub[2][2];
        int cc = 0;
        for (int c = 0; c < 3; c++) {
            if (c == j) continue;
            sub[0][cc] = s[1][c];
            sub[1][cc] = s[2][c];
            cc++;
        }
        double minor = sub[0][0] * sub[1][1] - sub[0][1] * sub[1][0];
        double cof = ((0 + j) % 2 == 0) ? minor : -minor; /* matrix.rs:142-151 */
        det += s[0][j] * cof;
    }
    return det;
}

/* matrix.rs:120-151: cofactor(row, col) of a 4x4 */
static double m_cofactor4(const mat4 *a, int row, int col) {
    double s[3][3];
    int rr = 0;
    for (int i = 0; i < 4; i++) {
        if (i == row) continue;
        int cc = 0;
        for (int j = 0; j < 4; j++) {
            if (j == col) continue;
            s[rr][cc++] = a->m[i][j];
        }
        rr++;
    }
    double minor = m_det3(s);
    return ((row + col) % 2 == 0) ? minor : -minor;
}

/* matrix.rs:153-179.  Returns 0 when det == 0.0 exactly (NonInvertibleMatrix). */
static int m_inverse(const mat4 *a, mat4 *out) {
    double det = m_det4(a);
    if (!(det != 0.0)) return 0;
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) out->m[col][row] = m_cofactor4(a, row, col) / det;
    return 1;
}

static mat4 m_from16(const double a[16]) {
    mat4 r;
    memcpy(&r.m[0][0], a, 16 * sizeof(double));
    return r;
}
static void m_to16(const mat4 *a, double out[16]) { memcpy(out, &a->m[0][0], 16 * sizeof(double)); }

void orc_matrix_identity(double out[16]) {
    mat4 r = m_identity();
    m_to16(&r, out);
}
void orc_matrix_mul(const double a[16], const double b[16], double out[16]) {
    mat4 x = m_from16(a), y = m_from16(b), r = m_mul(&x, &y);
    m_to16(&r, out);
}
double orc_matrix_determinant(const double a[16]) {
    mat4 x = m_from16(a);
    return m_det4(&x);
}
int orc_matrix_inverse(const double a[16], double out[16]) {
    mat4 x = m_from16(a), r;
    if (!m_inverse(&x, &r)) return ORC_ERR_NON_INVERTIBLE;
    m_to16(&r, out);
    return ORC_OK;
}
/* matrix.rs:181-187 */
void orc_matrix_translation(double x, double y, double z, double out[16]) {
    mat4 r = m_identity();
    r.m[0][3] = x;
    r.m[1][3] = y;
    r.m[2][3] = z;
    m_to16(&r, out);
}
/* matrix.rs:189-195 */
void orc_matrix_scaling(double x, double y, double z, double out[16]) {
    mat4 r = m_identity();
    r.m[0][0] = x;
    r.m[1][1] = y;
    r.m[2][2] = z;
    m_to16(&r, out);
}
/* matrix.rs:197-205 */
void orc_matrix_rotation_x(double rad, double out[16]) {
    mat4 r = m_identity();
    double s = sin(rad), c = cos(rad);
    r.m[1][1] = c;
    r.m[2][1] = s;
    r.m[1][2] = -s;
    r.m[2][2] = c;
    m_to16(&r, out);
}
/* matrix.rs:207-215 */
void orc_matrix_rotation_y(double rad, double out[16]) {
    mat4 r = m_identity();
    double s = sin(rad), c = cos(rad);
    r.m[0][0] = c;
    r.m[2][0] = -s;
    r.m[0][2] = s;
    r.m[2][2] = c;
    m_to16(&r, out);
}
/* matrix.rs:217-225 */
void orc_matrix_rotation_z(double rad, double out[16]) {
    mat4 r = m_identity();
    double s = sin(rad), c = cos(rad);
    r.m[0][0] = c;
    r.m[1][0] = s;
    r.m[0][1] = -s;
    r.m[1][1] = c;
    m_to16(&r, out);
}
/* matrix.rs:227-236 */
void orc_matrix_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double out[16]) {
    mat4 r = m_identity();
    r.m[0][1] = xy;
    r.m[0][2] = xz;
    r.m[1][0] = yx;
    r.m[1][2] = yz;
    r.m[2][0] = zx;
    r.m[2][1] = zy;
    m_to16(&r, out);
}
void orc_matrix_mul_point(const double m[16], const double p[3], double out[3]) {
    mat4 x = m_from16(m);
    v3 r = m_mul_point(&x, v3_new(p[0], p[1], p[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_matrix_mul_vector(const double m[16], const double v[3], double out[3]) {
    mat4 x = m_from16(m);
    v3 r = m_mul_vector(&x, v3_new(v[0], v[1], v[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

/* transformation.rs:3-15 */
void orc_view_transform(const double from[3], const double to[3], const double up[3], double out[16]) {
    v3 f = v3_new(from[0], from[1], from[2]), t = v3_new(to[0], to[1], to[2]), u = v3_new(up[0], up[1], up[2]);
    v3 forward = v_norm(v_sub(t, f));
    v3 up_norm = v_norm(u);
    v3 left = v_cross(forward, up_norm);
    v3 true_up = v_cross(left, forward);
    mat4 o = m_identity();
    o.m[0][0] = left.x, o.m[0][1] = left.y, o.m[0][2] = left.z;
    o.m[1][0] = true_up.x, o.m[1][1] = true_up.y, o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x, o.m[2][1] = -forward.y, o.m[2][2] = -forward.z;
    mat4 tr = m_identity();
    tr.m[0][3] = -f.x;
    tr.m[1][3] = -f.y;
    tr.m[2][3] = -f.z;
    mat4 r = m_mul(&o, &tr);
    m_to16(&r, out);
}

/* ------------------------------------------------------------------ world model */

typedef struct {
    int kind;
    v3 a, b;
    int sub_a, sub_b;
    mat4 transform, inverse;
    int invertible;
} pattern_t;

typedef struct {
    int kind;
    int parent;
    mat4 transform, inverse;
    int invertible;
    int material; /* for groups: Option<Material>, -1 = None */
    double minimum, maximum;
    int closed;
    v3 p1, p2, p3, e1, e2, normal, n1, n2, n3;
    int *children;
    int nchildren, cap_children;
} shape_t;

typedef struct {
    int is_area;
    v3 position, intensity;
    v3 corner, uvec, vvec;
    uint32_t usteps, vsteps, samples;
} light_t;

struct orc_world {
    orc_material *materials;
    int nmaterials, cap_materials;
    pattern_t *patterns;
    int npatterns, cap_patterns;
    shape_t *shapes;
    int nshapes, cap_shapes;
    int *objects; /* World.objects: top-level shapes in insertion order */
    int nobjects, cap_objects;
    light_t *lights;
    int nlights, cap_lights;
    int depth_override; /* 0: the reference's literal 5 (camera.rs:80); n > 0: bounce budget n - 1 */
    /* camera.rs:7-14 */
    int has_camera;
    uint32_t hsize, vsize;
    mat4 cam_transform, cam_inverse;
    int cam_invertible;
    double half_width, half_height, pixel_size;
};

typedef struct {
    v3 origin, direction;
} ray_t;

/* intersection.rs:93-99 */
typedef struct {
    double t;
    int shape;
    double u, v;
} isect_t;

typedef struct {
    isect_t *data;
    int n, cap;
    isect_t inl[32];
} xs_t;

typedef struct {
    uint64_t primary, shadow, secondary, diverged;
} counters_t;

static __thread counters_t *tl_counters;

/* Nesting guard of color_at (not in the reference).  refracted_color keeps the bounce budget (world.rs:142, Q2), so a
 * ray that is caught between refracting surfaces -- a glass double cone around its apex, a glass smooth triangle whose
 * interpolated normal disagrees with its face -- recurses without end: the reference overflows its stack there (a
 * rayon worker has 2 MB) and aborts.  The oracle stops at ORC_MAX_NESTING nested color_at calls instead, marks the
 * pixel as DIVERGED (NaN in the canvas, counted in orc_stats.diverged_pixels) and carries on with the next pixel: a
 * scene with such pixels has no reference image, parity tests must not use it. */
#define ORC_MAX_NESTING 2048
#define ORC_WORKER_STACK ((size_t)64 << 20)
static __thread int tl_nesting, tl_diverged;

static void xs_init(xs_t *l) {
    l->data = l->inl;
    l->n = 0;
    l->cap = 32;
}
static void xs_free(xs_t *l) {
    if (l->data != l->inl) free(l->data);
}
static void xs_push(xs_t *l, double t, int shape, double u, double v) {
    if (l->n == l->cap) {
        int ncap = l->cap * 2;
        isect_t *nd = (isect_t *)malloc(sizeof(isect_t) * (size_t)ncap);
        memcpy(nd, l->data, sizeof(isect_t) * (size_t)l->n);
        if (l->data != l->inl) free(l->data);
        l->data = nd;
        l->cap = ncap;
    }
    isect_t r = {t, shape, u, v};
    l->data[l->n++] = r;
}
/* stable sort of l->data[from..n) by total_cmp(t) (see header comment on sort_unstable_by) */
static void xs_sort_from(xs_t *l, int from) {
    for (int i = from + 1; i < l->n; i++) {
        isect_t key = l->data[i];
        int64_t k = total_key(key.t);
        int j = i - 1;
        while (j >= from && total_key(l->data[j].t) > k) {
            l->data[j + 1] = l->data[j];
            j--;
        }
        l->data[j + 1] = key;
    }
}

#define GROW(arr, n, cap, type)                                            \
    do {                                                                   \
        if ((n) == (cap)) {                                                \
            (cap) = (cap) ? (cap)*2 : 16;                                  \
            (arr) = (type *)realloc((arr), sizeof(type) * (size_t)(cap)); \
        }                                                                  \
    } while (0)

orc_world *orc_world_new(void) { return (orc_world *)calloc(1, sizeof(orc_world)); }
/* Test hook: the reference hard-codes the bounce budget 5 (camera.rs:80); the product exposes it as an option, so
 * the oracle can be asked for other budgets too. */
void orc_world_set_depth(orc_world *w, int depth) { w->depth_override = depth + 1; }

void orc_world_free(orc_world *w) {
    if (!w) return;
    for (int i = 0; i < w->nshapes; i++) free(w->shapes[i].children);
    free(w->materials);
    free(w->patterns);
    free(w->shapes);
    free(w->objects);
    free(w->lights);
    free(w);
}

int orc_add_material(orc_world *w, const orc_material *m) {
    GROW(w->materials, w->nmaterials, w->cap_materials, orc_material);
    w->materials[w->nmaterials] = *m;
    return w->nmaterials++;
}

int orc_add_pattern(orc_world *w, int kind, const double a[3], const double b[3], int sub_a, int sub_b,
                    const double transform[16]) {
    GROW(w->patterns, w->npatterns, w->cap_patterns, pattern_t);
    pattern_t *p = &w->patterns[w->npatterns];
    memset(p, 0, sizeof *p);
    p->kind = kind;
    if (a) p->a = v3_new(a[0], a[1], a[2]);
    if (b) p->b = v3_new(b[0], b[1], b[2]);
    p->sub_a = sub_a;
    p->sub_b = sub_b;
    p->transform = transform ? m_from16(transform) : m_identity();
    p->invertible = m_inverse(&p->transform, &p->inverse);
    return w->npatterns++;
}

static int add_shape_common(orc_world *w, int kind, int parent, const mat4 *t, int material) {
    if (parent >= w->nshapes) return -ORC_ERR_INVALID;
    if (parent >= 0 && w->shapes[parent].kind != ORC_GROUP && w->shapes[parent].kind != ORC_CSG) return -ORC_ERR_INVALID;
    if (parent >= 0 && w->shapes[parent].kind == ORC_CSG) {
        /* Csg::new takes exactly a left and a right shape; a triangle under a Csg panics in Triangle::material()
         * (shape.rs:1521-1523: parent.as_group().unwrap()) */
        if (w->shapes[parent].nchildren >= 2 || kind == ORC_TRIANGLE || kind == ORC_SMOOTH_TRIANGLE)
            return -ORC_ERR_INVALID;
    }
    GROW(w->shapes, w->nshapes, w->cap_shapes, shape_t);
    shape_t *s = &w->shapes[w->nshapes];
    memset(s, 0, sizeof *s);
    s->kind = kind;
    s->parent = parent;
    s->transform = *t;
    s->invertible = m_inverse(&s->transform, &s->inverse);
    s->material = material;
    s->minimum = -INFINITY;
    s->maximum = INFINITY;
    int idx = w->nshapes++;
    if (parent >= 0) {
        shape_t *g = &w->shapes[parent];
        GROW(g->children, g->nchildren, g->cap_children, int);
        g->children[g->nchildren++] = idx;
    } else {
        GROW(w->objects, w->nobjects, w->cap_objects, int);
        w->objects[w->nobjects++] = idx;
    }
    return idx;
}

int orc_add_shape(orc_world *w, int kind, int parent, const double transform[16], int material, double minimum,
                  double maximum, int closed) {
    if (kind == ORC_TRIANGLE || kind == ORC_SMOOTH_TRIANGLE) return -ORC_ERR_INVALID;
    mat4 t = transform ? m_from16(transform) : m_identity();
    int idx = add_shape_common(w, kind, parent, &t, material);
    if (idx < 0) return idx;
    shape_t *s = &w->shapes[idx];
    if (kind == ORC_CYLINDER || kind == ORC_CONE) {
        s->minimum = minimum;
        s->maximum = maximum;
        s->closed = closed;
    }
    return idx;
}

/* shape.rs:2166-2184; Csg::transformation() is the identity (shape.rs:2250-2252) */
int orc_add_csg(orc_world *w, int parent, int operation) {
    if (operation < ORC_CSG_UNION || operation > ORC_CSG_DIFFERENCE) return -ORC_ERR_INVALID;
    mat4 t = m_identity();
    int idx = add_shape_common(w, ORC_CSG, parent, &t, -1);
    if (idx < 0) return idx;
    w->shapes[idx].closed = operation;
    return idx;
}

int orc_add_triangle(orc_world *w, int parent, const double p1[3], const double p2[3], const double p3[3],
                     const double n1[3], const double n2[3], const double n3[3], int material) {
    /* Triangle/SmoothTriangle::transformation() is the identity (shape.rs:1550-1552, 1786-1788) */
    mat4 t = m_identity();
    int smooth = (n1 && n2 && n3);
    int idx = add_shape_common(w, smooth ? ORC_SMOOTH_TRIANGLE : ORC_TRIANGLE, parent, &t, material);
    if (idx < 0) return idx;
    shape_t *s = &w->shapes[idx];
    s->p1 = v3_new(p1[0], p1[1], p1[2]);
    s->p2 = v3_new(p2[0], p2[1], p2[2]);
    s->p3 = v3_new(p3[0], p3[1], p3[2]);
    /* shape.rs:1463-1465 / 1693-1694 */
    s->e1 = v_sub(s->p2, s->p1);
    s->e2 = v_sub(s->p3, s->p1);
    if (smooth) {
        s->n1 = v3_new(n1[0], n1[1], n1[2]);
        s->n2 = v3_new(n2[0], n2[1], n2[2]);
        s->n3 = v3_new(n3[0], n3[1], n3[2]);
    } else {
        s->normal = v_norm(v_cross(s->e2, s->e1));
    }
    return idx;
}

int orc_shape_set_material(orc_world *w, int shape, int material) {
    if (shape < 0 || shape >= w->nshapes) return ORC_ERR_INVALID;
    w->shapes[shape].material = material;
    return ORC_OK;
}

int orc_add_point_light(orc_world *w, const double position[3], const double intensity[3]) {
    GROW(w->lights, w->nlights, w->cap_lights, light_t);
    light_t *l = &w->lights[w->nlights];
    memset(l, 0, sizeof *l);
    l->position = v3_new(position[0], position[1], position[2]);
    l->intensity = v3_new(intensity[0], intensity[1], intensity[2]);
    return w->nlights++;
}

/* light.rs:80-104 */
int orc_add_area_light(orc_world *w, const double corner[3], const double full_uvec[3], uint32_t usteps,
                       const double full_vvec[3], uint32_t vsteps, const double intensity[3]) {
    GROW(w->lights, w->nlights, w->cap_lights, light_t);
    light_t *l = &w->lights[w->nlights];
    memset(l, 0, sizeof *l);
    l->is_area = 1;
    v3 fu = v3_new(full_uvec[0], full_uvec[1], full_uvec[2]);
    v3 fv = v3_new(full_vvec[0], full_vvec[1], full_vvec[2]);
    l->corner = v3_new(corner[0], corner[1], corner[2]);
    l->uvec = v_div(fu, (double)usteps);
    l->vvec = v_div(fv, (double)vsteps);
    l->usteps = usteps;
    l->vsteps = vsteps;
    l->samples = usteps * vsteps;
    l->position = v_div(v_add(fu, fv), 2.); /* Q12: corner ignored (light.rs:91-92) */
    l->intensity = v3_new(intensity[0], intensity[1], intensity[2]);
    return w->nlights++;
}

/* camera.rs:17-37 */
int orc_set_camera(orc_world *w, uint32_t hsize, uint32_t vsize, double field_of_view, const double view[16]) {
    w->hsize = hsize;
    w->vsize = vsize;
    w->cam_transform = view ? m_from16(view) : m_identity();
    w->cam_invertible = m_inverse(&w->cam_transform, &w->cam_inverse);
    double half_view = tan(field_of_view / 2.);
    double aspect = (double)hsize / (double)vsize;
    if (aspect >= 1.) {
        w->half_width = half_view;
        w->half_height = half_view / aspect;
    } else {
        w->half_width = half_view * aspect;
        w->half_height = half_view;
    }
    w->pixel_size = w->half_width * 2. / (double)hsize;
    w->has_camera = 1;
    return ORC_OK;
}

int orc_camera_params(orc_world *w, double out[3]) {
    out[0] = w->half_width;
    out[1] = w->half_height;
    out[2] = w->pixel_size;
    return ORC_OK;
}

/* ------------------------------------------------------------------ intersection.rs: Ray */

/* intersection.rs:20-22 */
static inline v3 ray_position(const ray_t *r, double t) { return v_add(r->origin, v_mul(r->direction, t)); }
/* intersection.rs:24-28 */
static inline ray_t ray_transform(const ray_t *r, const mat4 *m) {
    ray_t o;
    o.origin = m_mul_point(m, r->origin);
    o.direction = m_mul_vector(m, r->direction);
    return o;
}
/* intersection.rs:34-39 (Q7: compares with radius, not radius^2) */
static inline int ray_check_cap(const ray_t *r, double t, double radius) {
    double x = r->origin.x + t * r->direction.x;
    double z = r->origin.z + t * r->direction.z;
    return (x * x + z * z) <= radius;
}

/* ------------------------------------------------------------------ shape.rs: local_intersect */

/* shape.rs:218-239 */
static void sphere_local_intersect(int self, const ray_t *ray, xs_t *out) {
    v3 sphere_to_ray = v_sub(ray->origin, v3_new(0., 0., 0.));
    double a = v_dot(ray->direction, ray->direction);
    double b = 2. * v_dot(ray->direction, sphere_to_ray);
    double c = v_dot(sphere_to_ray, sphere_to_ray) - 1.;
    double discriminant = b * b - 4. * a * c;
    if (discriminant < 0.) return;
    double t1 = (-b - sqrt(discriminant)) / (2. * a);
    double t2 = (-b + sqrt(discriminant)) / (2. * a);
    xs_push(out, t1, self, NAN, NAN);
    xs_push(out, t2, self, NAN, NAN);
}

/* shape.rs:510-517 */
static void plane_local_intersect(int self, const ray_t *ray, xs_t *out) {
    if (fabs(ray->direction.y) < EPSILON) return;
    double t = -ray->origin.y / ray->direction.y;
    xs_push(out, t, self, NAN, NAN);
}

/* shape.rs:708-720 (Q9: divides by zero on purpose) */
static void cube_check_axis(double origin, double direction, double *tmin, double *tmax) {
    double t_min_numerator = -1. - origin;
    double t_max_numerator = 1. - origin;
    double t_min = t_min_numerator / direction;
    double t_max = t_max_numerator / direction;
    if (t_min > t_max) {
        double tmp = t_min;
        t_min = t_max;
        t_max = tmp;
    }
    *tmin = t_min;
    *tmax = t_max;
}

/* Iterator::max_by(total_cmp): last maximum; Iterator::min_by(total_cmp): first minimum */
static double max3_total(double a, double b, double c) {
    double m = a;
    if (total_key(b) >= total_key(m)) m = b;
    if (total_key(c) >= total_key(m)) m = c;
    return m;
}
static double min3_total(double a, double b, double c) {
    double m = a;
    if (total_key(b) < total_key(m)) m = b;
    if (total_key(c) < total_key(m)) m = c;
    return m;
}

/* shape.rs:665-704 */
static void cube_local_intersect(int self, const ray_t *ray, xs_t *out) {
    double xt_min, xt_max, yt_min, yt_max, zt_min, zt_max;
    cube_check_axis(ray->origin.x, ray->direction.x, &xt_min, &xt_max);
    if (xt_min == INFINITY || xt_max == -INFINITY) return;
    cube_check_axis(ray->origin.y, ray->direction.y, &yt_min, &yt_max);
    if (yt_min == INFINITY || yt_max == -INFINITY) return;
    cube_check_axis(ray->origin.z, ray->direction.z, &zt_min, &zt_max);
    double t_min = max3_total(xt_min, yt_min, zt_min);
    double t_max = min3_total(xt_max, yt_max, zt_max);
    if (t_min > t_max) return;
    xs_push(out, t_min, self, NAN, NAN);
    xs_push(out, t_max, self, NAN, NAN);
}

/* shape.rs:938-959 */
static void cylinder_intersect_caps(const shape_t *s, int self, const ray_t *ray, xs_t *out) {
    if (!s->closed || equal(ray->direction.y, 0.)) return;
    double t = (s->minimum - ray->origin.y) / ray->direction.y;
    if (ray_check_cap(ray, t, 1.)) xs_push(out, t, self, NAN, NAN);
    t = (s->maximum - ray->origin.y) / ray->direction.y;
    if (ray_check_cap(ray, t, 1.)) xs_push(out, t, self, NAN, NAN);
}

/* shape.rs:895-936 */
static void cylinder_local_intersect(const shape_t *s, int self, const ray_t *ray, xs_t *out) {
    double a = ray->direction.x * ray->direction.x + ray->direction.z * ray->direction.z;
    if (equal(a, 0.)) {
        cylinder_intersect_caps(s, self, ray, out);
        return;
    }
    double b = 2. * ray->origin.x * ray->direction.x + 2. * ray->origin.z * ray->direction.z;
    double c = ray->origin.x * ray->origin.x + ray->origin.z * ray->origin.z - 1.;
    double discriminant = b * b - 4. * a * c;
    if (discriminant < 0.) return; /* Q10: no caps either */
    double t0 = (-b - sqrt(discriminant)) / (2. * a);
    double t1 = (-b + sqrt(discriminant)) / (2. * a);
    if (t0 > t1) {
        double tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    double y0 = ray->origin.y + t0 * ray->direction.y;
    if (s->minimum < y0 && y0 < s->maximum) xs_push(out, t0, self, NAN, NAN);
    double y1 = ray->origin.y + t1 * ray->direction.y;
    if (s->minimum < y1 && y1 < s->maximum) xs_push(out, t1, self, NAN, NAN);
    cylinder_intersect_caps(s, self, ray, out);
}

/* shape.rs:1260-1281 (Q8: |min|, |max| for plane height AND radius) */
static void cone_intersect_caps(const shape_t *s, int self, const ray_t *ray, xs_t *out) {
    if (!s->closed || equal(ray->direction.y, 0.)) return;
    double t = (fabs(s->minimum) - ray->origin.y) / ray->direction.y;
    if (ray_check_cap(ray, t, fabs(s->minimum))) xs_push(out, t, self, NAN, NAN);
    t = (fabs(s->maximum) - ray->origin.y) / ray->direction.y;
    if (ray_check_cap(ray, t, fabs(s->maximum))) xs_push(out, t, self, NAN, NAN);
}

/* shape.rs:1212-1258 */
static void cone_local_intersect(const shape_t *s, int self, const ray_t *ray, xs_t *out) {
    double a = ray->direction.x * ray->direction.x - ray->direction.y * ray->direction.y +
               ray->direction.z * ray->direction.z;
    double b = 2. * ray->origin.x * ray->direction.x - 2. * ray->origin.y * ray->direction.y +
               2. * ray->origin.z * ray->direction.z;
    double c = ray->origin.x * ray->origin.x - ray->origin.y * ray->origin.y + ray->origin.z * ray->origin.z;
    if (equal(a, 0.)) {
        if (s->closed) {
            cone_intersect_caps(s, self, ray, out);
        } else {
            double t = -c / (2. * b); /* Q8: no y-range check, no b != 0 check */
            xs_push(out, t, self, NAN, NAN);
        }
        return;
    }
    double discriminant = b * b - 4. * a * c;
    if (discriminant < 0.) return;
    double t0 = (-b - sqrt(discriminant)) / (2. * a);
    double t1 = (-b + sqrt(discriminant)) / (2. * a);
    if (t0 > t1) {
        double tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    double y0 = ray->origin.y + t0 * ray->direction.y;
    if (s->minimum < y0 && y0 < s->maximum) xs_push(out, t0, self, NAN, NAN);
    double y1 = ray->origin.y + t1 * ray->direction.y;
    if (s->minimum < y1 && y1 < s->maximum) xs_push(out, t1, self, NAN, NAN);
    cone_intersect_caps(s, self, ray, out);
}

/* shape.rs:1479-1502 and 1712-1735 (identical Moeller-Trumbore) */
static void triangle_local_intersect(const shape_t *s, int self, const ray_t *ray, xs_t *out) {
    v3 direction_cross_edge2 = v_cross(ray->direction, s->e2);
    double determinant = v_dot(s->e1, direction_cross_edge2);
    if (fabs(determinant) < EPSILON) return;
    double f = 1. / determinant;
    v3 point1_to_origin = v_sub(ray->origin, s->p1);
    double u = f * v_dot(point1_to_origin, direction_cross_edge2);
    if (!(0. <= u && u <= 1.)) return;
    v3 origin_cross_edge1 = v_cross(point1_to_origin, s->e1);
    double v = f * v_dot(ray->direction, origin_cross_edge1);
    if (v < 0. || (u + v) > 1.) return;
    double t = f * v_dot(s->e2, origin_cross_edge1);
    xs_push(out, t, self, u, v);
}

static void shape_local_intersect(const orc_world *w, int self, const ray_t *local, xs_t *out);

/* Shape::includes: a leaf includes itself (shape.rs:136-138: same id and type), a Group whatever one of its children
 * includes (shape.rs:2023-2030), a Csg whatever its left or right includes (shape.rs:2278-2280) */
static int shape_includes(const orc_world *w, int self, int object) {
    const shape_t *s = &w->shapes[self];
    if (s->kind == ORC_GROUP || s->kind == ORC_CSG) {
        for (int i = 0; i < s->nchildren; i++)
            if (shape_includes(w, s->children[i], object)) return 1;
        return 0;
    }
    return self == object;
}

/* shape.rs:2186-2197 */
static int csg_intersection_allowed(int operation, int left_hit, int in_left, int in_right) {
    switch (operation) {
    case ORC_CSG_UNION: return (left_hit && !in_right) || (!left_hit && !in_left);
    case ORC_CSG_INTERSECTION: return (left_hit && in_right) || (!left_hit && in_left);
    default: return (left_hit && !in_right) || (!left_hit && in_left); /* Difference */
    }
}

/* shape.rs:2203-2225 on out->data[from..n): the records that pass are moved to the front of that range */
static void csg_filter(const orc_world *w, int self, xs_t *out, int from) {
    const shape_t *s = &w->shapes[self];
    int in_left = 0, in_right = 0, keep = from;
    for (int i = from; i < out->n; i++) {
        const int left_hit = s->nchildren > 0 && shape_includes(w, s->children[0], out->data[i].shape);
        if (csg_intersection_allowed(s->closed, left_hit, in_left, in_right)) out->data[keep++] = out->data[i];
        if (left_hit)
            in_left = !in_left;
        else
            in_right = !in_right;
    }
    out->n = keep;
}

/* <Shape as Intersect>::intersect: shape.rs:255-258, 539-542, 781-784, 1014-1017, 1346-1349,
 * 1506-1508 (triangles: no ray.transform), 2033-2038 (groups) */
static void shape_intersect(const orc_world *w, int self, const ray_t *ray, xs_t *out) {
    const shape_t *s = &w->shapes[self];
    if (s->kind == ORC_TRIANGLE || s->kind == ORC_SMOOTH_TRIANGLE) {
        triangle_local_intersect(s, self, ray, out);
        return;
    }
    if (s->kind == ORC_CSG) {
        /* shape.rs:2238-2243: ray.transform(self.transformation()) -- the identity itself, not its inverse */
        ray_t local = ray_transform(ray, &s->transform);
        shape_local_intersect(w, self, &local, out);
        return;
    }
    ray_t local = ray_transform(ray, &s->inverse);
    shape_local_intersect(w, self, &local, out);
}

static void shape_local_intersect(const orc_world *w, int self, const ray_t *local, xs_t *out) {
    const shape_t *s = &w->shapes[self];
    switch (s->kind) {
    case ORC_SPHERE: sphere_local_intersect(self, local, out); break;
    case ORC_PLANE: plane_local_intersect(self, local, out); break;
    case ORC_CUBE: cube_local_intersect(self, local, out); break;
    case ORC_CYLINDER: cylinder_local_intersect(s, self, local, out); break;
    case ORC_CONE: cone_local_intersect(s, self, local, out); break;
    case ORC_TRIANGLE:
    case ORC_SMOOTH_TRIANGLE: triangle_local_intersect(s, self, local, out); break;
    case ORC_GROUP: {
        /* shape.rs:1930-1942 */
        int from = out->n;
        for (int i = 0; i < s->nchildren; i++) shape_intersect(w, s->children[i], local, out);
        xs_sort_from(out, from);
        break;
    }
    case ORC_CSG: {
        /* shape.rs:2227-2235: left's records, then right's, sorted (see the header comment on sort_unstable_by),
         * filtered */
        int from = out->n;
        for (int i = 0; i < s->nchildren && i < 2; i++) shape_intersect(w, s->children[i], local, out);
        xs_sort_from(out, from);
        csg_filter(w, self, out, from);
        break;
    }
    default: break; /* TestShape::intersect is unimplemented!() (shape.rs:2459-2461) */
    }
}

/* world.rs:176-186 */
static void world_intersect(const orc_world *w, const ray_t *ray, xs_t *out) {
    for (int i = 0; i < w->nobjects; i++) shape_intersect(w, w->objects[i], ray, out);
    xs_sort_from(out, 0);
}

/* intersection.rs:207-219 */
static int hit_index(const isect_t *xs, int n) {
    int lowest = -1;
    for (int i = 0; i < n; i++) {
        if (lowest >= 0) {
            if (xs[i].t > 0. && xs[i].t < xs[lowest].t) lowest = i;
        } else if (xs[i].t > 0.) {
            lowest = i;
        }
    }
    return lowest;
}

int orc_hit_index(int n, const double *ts) {
    int lowest = -1;
    for (int i = 0; i < n; i++) {
        if (lowest >= 0) {
            if (ts[i] > 0. && ts[i] < ts[lowest]) lowest = i;
        } else if (ts[i] > 0.) {
            lowest = i;
        }
    }
    return lowest;
}

/* ------------------------------------------------------------------ shape.rs: normals */

/* shape.rs:61-63 default + per-shape material(); triangles: shape.rs:1520-1532 / 1756-1768 */
static const orc_material *shape_material(const orc_world *w, int self) {
    const shape_t *s = &w->shapes[self];
    if ((s->kind == ORC_TRIANGLE || s->kind == ORC_SMOOTH_TRIANGLE) && s->parent >= 0) {
        const shape_t *g = &w->shapes[s->parent];
        if (g->material >= 0) return &w->materials[g->material];
    }
    return &w->materials[s->material];
}

static v3 shape_local_normal_at(const orc_world *w, int self, v3 p, double hu, double hv) {
    const shape_t *s = &w->shapes[self];
    switch (s->kind) {
    case ORC_SPHERE: return v_sub(p, v3_new(0., 0., 0.)); /* shape.rs:273-279 */
    case ORC_PLANE: return v3_new(0., 1., 0.);             /* shape.rs:550-553 */
    case ORC_CUBE: {                                       /* shape.rs:743-757 */
        double max_c = max3_total(fabs(p.x), fabs(p.y), fabs(p.z));
        if (max_c == fabs(p.x)) return v3_new(p.x, 0., 0.);
        if (max_c == fabs(p.y)) return v3_new(0., p.y, 0.);
        return v3_new(0., 0., p.z);
    }
    case ORC_CYLINDER: { /* shape.rs:983-998 */
        double distance = p.x * p.x + p.z * p.z;
        if (distance < 1. && p.y >= s->maximum - EPSILON) return v3_new(0., 1., 0.);
        if (distance < 1. && p.y <= s->minimum + EPSILON) return v3_new(0., -1., 0.);
        return v3_new(p.x, 0., p.z);
    }
    case ORC_CONE: { /* shape.rs:1302-1314 (Q8: dist < 1) */
        double distance = p.x * p.x + p.z * p.z;
        if (distance < 1. && p.y >= s->maximum - EPSILON) return v3_new(0., 1., 0.);
        if (distance < 1. && p.y <= s->minimum + EPSILON) return v3_new(0., -1., 0.);
        double y = sqrt(p.x * p.x + p.z * p.z);
        y = (p.y > 0.) ? -y : y;
        return v3_new(p.x, y, p.z);
    }
    case ORC_TRIANGLE: return s->normal; /* shape.rs:1516-1518 */
    case ORC_SMOOTH_TRIANGLE:            /* shape.rs:1749-1754 */
        return v_add(v_add(v_mul(s->n2, hu), v_mul(s->n3, hv)), v_mul(s->n1, 1. - hu - hv));
    case ORC_TEST_SHAPE: return v3_new(p.x, p.y, p.z); /* shape.rs:2485-2487 */
    default: return v3_new(NAN, NAN, NAN);             /* Group::local_normal_at is unimplemented!() */
    }
}

/* shape.rs:82-90: parent first, then own inverse */
static v3 shape_world_to_object(const orc_world *w, int self, v3 point) {
    const shape_t *s = &w->shapes[self];
    if (s->parent >= 0) point = shape_world_to_object(w, s->parent, point);
    return m_mul_point(&s->inverse, point);
}

/* shape.rs:94-103: transpose(inverse) * n, normalise, then the parent */
static v3 shape_normal_to_world(const orc_world *w, int self, v3 normal) {
    const shape_t *s = &w->shapes[self];
    mat4 tr = m_transpose(&s->inverse);
    normal = m_mul_vector(&tr, normal);
    normal = v_norm(normal);
    if (s->parent >= 0) normal = shape_normal_to_world(w, s->parent, normal);
    return normal;
}

/* shape.rs:43-47 */
static v3 shape_normal_at(const orc_world *w, int self, v3 point, double hu, double hv) {
    v3 local_point = shape_world_to_object(w, self, point);
    v3 local_normal = shape_local_normal_at(w, self, local_point, hu, hv);
    return shape_normal_to_world(w, self, local_normal);
}

/* ------------------------------------------------------------------ pattern.rs */

/* Rust `%` on f64 is C fmod */
static v3 pattern_at(const orc_world *w, int pi, v3 p) {
    const pattern_t *pt = &w->patterns[pi];
    switch (pt->kind) {
    case ORC_PAT_STRIPE: /* pattern.rs:114-121 */
        return pattern_at(w, (fmod(floor(p.x), 2.) == 0.) ? pt->sub_a : pt->sub_b, p);
    case ORC_PAT_GRADIENT: { /* pattern.rs:228-231 */
        v3 distance = v_sub(pt->b, pt->a);
        return v_add(pt->a, v_mul(distance, p.x));
    }
    case ORC_PAT_RING: /* pattern.rs:297-303 */
        return (fmod(floor(sqrt(p.x * p.x + p.z * p.z)), 2.) == 0.) ? pt->a : pt->b;
    case ORC_PAT_CHECKER: /* pattern.rs:380-386 */
        return pattern_at(w, (fmod(floor(p.x) + floor(p.y) + floor(p.z), 2.) == 0.) ? pt->sub_a : pt->sub_b, p);
    case ORC_PAT_RADIAL: { /* pattern.rs:472-476 */
        double distance = sqrt(p.x * p.x + p.z * p.z);
        double fraction = distance - floor(distance);
        return v_add(pt->a, v_mul(v_sub(pt->b, pt->a), fraction));
    }
    case ORC_PAT_SOLID: return pt->a; /* pattern.rs:533-535 */
    case ORC_PAT_BLENDED: {           /* pattern.rs:584-591 */
        v3 c1 = pattern_at(w, pt->sub_a, p), c2 = pattern_at(w, pt->sub_b, p);
        return v3_new((c1.x + c2.x) / 2., (c1.y + c2.y) / 2., (c1.z + c2.z) / 2.);
    }
    case ORC_PAT_TEST: return v3_new(p.x, p.y, p.z); /* pattern.rs:669-671 */
    default: return v3_new(NAN, NAN, NAN);
    }
}

/* pattern.rs:25-32 (Q13: only the shape's own transform); SolidPattern override pattern.rs:538-540 */
static v3 pattern_at_shape(const orc_world *w, int pi, int shape, v3 world_point) {
    const pattern_t *pt = &w->patterns[pi];
    if (pt->kind == ORC_PAT_SOLID) return pt->a;
    v3 object_point = m_mul_point(&w->shapes[shape].inverse, world_point);
    v3 pattern_point = m_mul_point(&pt->inverse, object_point);
    return pattern_at(w, pi, pattern_point);
}

/* ------------------------------------------------------------------ material.rs */

/* material.rs:20-75 */
static v3 material_lighting(const orc_world *w, const orc_material *m, int object, const light_t *light, v3 point,
                            v3 eye_vector, v3 normal_vector, double light_intensity) {
    v3 color = (m->pattern >= 0) ? pattern_at_shape(w, m->pattern, object, point)
                                 : v3_new(m->color[0], m->color[1], m->color[2]);
    v3 effective_color = c_mul(color, light->intensity);
    v3 light_vector = v_norm(v_sub(light->position, point));
    v3 ambient = v_mul(effective_color, m->ambient);
    double light_dot_normal = v_dot(light_vector, normal_vector);
    v3 diffuse, specular;
    if (light_dot_normal < 0.) {
        diffuse = v3_new(0., 0., 0.);
        specular = v3_new(0., 0., 0.);
    } else {
        diffuse = v_mul(v_mul(v_mul(effective_color, m->diffuse), light_dot_normal), light_intensity);
        v3 reflect_vector = v_reflect(v_neg(light_vector), normal_vector);
        double reflect_dot_eye = v_dot(reflect_vector, eye_vector);
        if (reflect_dot_eye <= 0.) {
            specular = v_mul(v3_new(0., 0., 0.), light_intensity);
        } else {
            double factor = pow(reflect_dot_eye, m->shininess);
            specular = v_mul(v_mul(v_mul(light->intensity, m->specular), factor), light_intensity);
        }
    }
    return v_add(v_add(ambient, specular), diffuse);
}

/* ------------------------------------------------------------------ intersection.rs: comps */

typedef struct {
    double t;
    int object;
    double u, v;
    v3 point, eye_vector, normal_vector;
    int inside;
    v3 over_point, reflect_vector;
    int has_n;
    double n1, n2;
    v3 under_point;
} comps_t;

/* PartialEq for Intersection: intersection.rs:197-203 (shape index == (shape_type, id)) */
static inline int isect_eq(const isect_t *a, const isect_t *b) { return a->t == b->t && a->shape == b->shape; }

/* intersection.rs:123-194 */
static comps_t prepare_computations(const orc_world *w, const isect_t *self, const ray_t *ray, const isect_t *xs,
                                    int n) {
    comps_t c;
    memset(&c, 0, sizeof c);
    c.t = self->t;
    c.object = self->shape;
    c.u = self->u;
    c.v = self->v;
    c.point = ray_position(ray, self->t);
    c.eye_vector = v_neg(ray->direction);
    c.normal_vector = shape_normal_at(w, self->shape, c.point, self->u, self->v);
    double normal_dot_eye = v_dot(c.normal_vector, c.eye_vector);
    if (normal_dot_eye < 0.) {
        c.inside = 1;
        c.normal_vector = v_neg(c.normal_vector);
    }
    c.reflect_vector = v_reflect(ray->direction, c.normal_vector);
    c.over_point = v_add(c.point, v_mul(c.normal_vector, EPSILON));
    c.under_point = v_sub(c.point, v_mul(c.normal_vector, EPSILON));

    /* containers loop, intersection.rs:150-179 */
    int inl[64];
    int *containers = inl, ncont = 0, capc = 64;
    int have1 = 0, have2 = 0;
    for (int i = 0; i < n; i++) {
        const isect_t *it = &xs[i];
        int is_self = isect_eq(it, self);
        if (is_self) {
            c.n1 = (ncont == 0) ? 1. : shape_material(w, containers[ncont - 1])->refractive_index;
            have1 = 1;
        }
        int pos = -1;
        for (int k = 0; k < ncont; k++)
            if (containers[k] == it->shape) {
                pos = k;
                break;
            }
        if (pos >= 0) {
            memmove(&containers[pos], &containers[pos + 1], sizeof(int) * (size_t)(ncont - pos - 1));
            ncont--;
        } else {
            if (ncont == capc) {
                int *nc = (int *)malloc(sizeof(int) * (size_t)capc * 2);
                memcpy(nc, containers, sizeof(int) * (size_t)ncont);
                if (containers != inl) free(containers);
                containers = nc;
                capc *= 2;
            }
            containers[ncont++] = it->shape;
        }
        if (is_self) {
            c.n2 = (ncont == 0) ? 1. : shape_material(w, containers[ncont - 1])->refractive_index;
            have2 = 1;
            break;
        }
    }
    if (containers != inl) free(containers);
    c.has_n = have1 && have2;
    if (!c.has_n) c.n1 = c.n2 = NAN;
    return c;
}

/* intersection.rs:66-90; x.powi(5) == x * ((x*x)*(x*x)) */
static double comps_schlick(const comps_t *c) {
    if (!c->has_n) return 0.;
    double n1 = c->n1, n2 = c->n2;
    double cosv = v_dot(c->eye_vector, c->normal_vector);
    if (n1 > n2) {
        double n = n1 / n2;
        double sin2_t = (n * n) * (1. - cosv * cosv);
        if (sin2_t > 1.) return 1.;
        double cos_t = sqrt(1. - sin2_t);
        cosv = cos_t;
    }
    double q = (n1 - n2) / (n1 + n2);
    double r0 = q * q;
    double x = 1. - cosv;
    double x2 = x * x;
    double x5 = x * (x2 * x2);
    return r0 + (1. - r0) * x5;
}

/* ------------------------------------------------------------------ world.rs */

static v3 world_color_at(const orc_world *w, const ray_t *ray, int remaining);
static ray_t camera_ray_for_pixel_fwd(const orc_world *w, uint32_t px, uint32_t py);

/* world.rs:78-91 */
static int world_is_shadowed(const orc_world *w, v3 point, v3 light_position) {
    v3 distance_vector = v_sub(light_position, point);
    double distance = v_magnitude(distance_vector);
    v3 direction = v_norm(distance_vector);
    ray_t ray = {point, direction};
    if (tl_counters) tl_counters->shadow++;
    xs_t xs;
    xs_init(&xs);
    world_intersect(w, &ray, &xs);
    int h = hit_index(xs.data, xs.n);
    int result = 0;
    if (h >= 0) result = xs.data[h].t < distance && shape_material(w, xs.data[h].shape)->casts_shadow;
    xs_free(&xs);
    return result;
}

/* light.rs:106-108 */
static v3 area_point_on_light(const light_t *l, uint32_t u, uint32_t v) {
    return v_add(v_add(l->corner, v_mul(l->uvec, (double)u + 0.5)), v_mul(l->vvec, (double)v + 0.5));
}

/* light.rs:50-56 and 112-125 */
static double light_intensity_at(const orc_world *w, const light_t *l, v3 point) {
    if (!l->is_area) return world_is_shadowed(w, point, l->position) ? 0. : 1.;
    double total = 0.;
    for (uint32_t v = 0; v < l->vsteps; v++)
        for (uint32_t u = 0; u < l->usteps; u++) {
            v3 lp = area_point_on_light(l, u, v);
            if (!world_is_shadowed(w, point, lp)) total += 1.;
        }
    return total / (double)l->samples;
}

/* world.rs:93-106 */
static v3 world_reflected_color(const orc_world *w, const comps_t *c, int remaining) {
    if (remaining == 0) return v3_new(0., 0., 0.);
    double material_reflective = shape_material(w, c->object)->reflectivity;
    if (material_reflective == 0.) return v3_new(0., 0., 0.);
    ray_t reflect_ray = {c->over_point, c->reflect_vector};
    if (tl_counters) tl_counters->secondary++;
    v3 color = world_color_at(w, &reflect_ray, remaining - 1);
    return v_mul(color, material_reflective);
}

/* world.rs:108-147 (Q2: remaining is NOT decremented) */
static v3 world_refracted_color(const orc_world *w, const comps_t *c, int remaining) {
    const orc_material *m = shape_material(w, c->object);
    if (m->transparency == 0. || remaining == 0) return v3_new(0., 0., 0.);
    if (c->has_n) {
        double n_ratio = c->n1 / c->n2;
        double cos_i = v_dot(c->eye_vector, c->normal_vector);
        double sin2_t = (n_ratio * n_ratio) * (1. - cos_i * cos_i);
        if (sin2_t > 1.) return v3_new(0., 0., 0.);
        double cos_t = sqrt(1. - sin2_t);
        v3 direction = v_sub(v_mul(c->normal_vector, n_ratio * cos_i - cos_t), v_mul(c->eye_vector, n_ratio));
        ray_t refracted_ray = {c->under_point, direction};
        if (tl_counters) tl_counters->secondary++;
        v3 color = v_mul(world_color_at(w, &refracted_ray, remaining), m->transparency);
        return color;
    }
    return v3_new(1., 1., 1.);
}

/* world.rs:42-76 (Q3: reflected and refracted are evaluated inside the per-light loop) */
static v3 world_shade_hit(const orc_world *w, const comps_t *c, int remaining) {
    v3 acc = v3_new(0., 0., 0.);
    int have = 0;
    for (int li = 0; li < w->nlights; li++) {
        const light_t *light = &w->lights[li];
        double light_intensity = light_intensity_at(w, light, c->over_point);
        const orc_material *material = shape_material(w, c->object);
        v3 surface_color = material_lighting(w, material, c->object, light, c->over_point, c->eye_vector,
                                             c->normal_vector, light_intensity);
        v3 reflected_color = world_reflected_color(w, c, remaining);
        v3 refracted_color = world_refracted_color(w, c, remaining);
        v3 color;
        if (material->reflectivity > 0. && material->transparency > 0.) {
            double reflectance = comps_schlick(c);
            color = v_add(v_add(surface_color, v_mul(reflected_color, reflectance)),
                          v_mul(refracted_color, 1. - reflectance));
        } else {
            color = v_add(v_add(surface_color, reflected_color), refracted_color);
        }
        if (have) {
            acc = v_add(acc, color);
        } else {
            acc = color;
            have = 1;
        }
    }
    return have ? acc : v3_new(0., 0., 0.);
}

/* world.rs:30-38 */
static v3 world_color_at(const orc_world *w, const ray_t *ray, int remaining) {
    xs_t xs;
    v3 color = v3_new(0., 0., 0.);
    if (tl_diverged) return color; /* unwinding a diverged pixel: no further work */
    if (tl_nesting >= ORC_MAX_NESTING) {
        tl_diverged = 1;
        return color;
    }
    xs_init(&xs);
    world_intersect(w, ray, &xs);
    int h = hit_index(xs.data, xs.n);
    if (h >= 0) {
        comps_t c = prepare_computations(w, &xs.data[h], ray, xs.data, xs.n);
        xs_free(&xs); /* the list is not needed below; frees memory before recursing */
        tl_nesting++;
        color = world_shade_hit(w, &c, remaining);
        tl_nesting--;
        return color;
    }
    xs_free(&xs);
    return color;
}

/* one pixel of Camera::render (camera.rs:78-84) under the nesting guard */
static v3 pixel_color(const orc_world *w, uint32_t x, uint32_t y, counters_t *counters) {
    ray_t ray = camera_ray_for_pixel_fwd(w, x, y);
    counters->primary++;
    tl_nesting = 0, tl_diverged = 0;
    v3 c = world_color_at(w, &ray, w->depth_override ? w->depth_override - 1 : 5); /* camera.rs:80 */
    if (tl_diverged) {
        counters->diverged++;
        c = v3_new(NAN, NAN, NAN);
        tl_diverged = 0;
    }
    return c;
}

/* ------------------------------------------------------------------ camera.rs */

/* camera.rs:41-59 */
static ray_t camera_ray_for_pixel(const orc_world *w, uint32_t px, uint32_t py);
static ray_t camera_ray_for_pixel_fwd(const orc_world *w, uint32_t px, uint32_t py) {
    return camera_ray_for_pixel(w, px, py);
}
static ray_t camera_ray_for_pixel(const orc_world *w, uint32_t px, uint32_t py) {
    double x_offset = ((double)px + 0.5) * w->pixel_size;
    double y_offset = ((double)py + 0.5) * w->pixel_size;
    double world_x = w->half_width - x_offset;
    double world_y = w->half_height - y_offset;
    v3 pixel = m_mul_point(&w->cam_inverse, v3_new(world_x, world_y, -1.));
    v3 origin = m_mul_point(&w->cam_inverse, v3_new(0., 0., 0.));
    v3 direction = v_norm(v_sub(pixel, origin));
    ray_t r = {origin, direction};
    return r;
}

/* Err(NonInvertibleMatrix) surfaces lazily in the reference (matrix.rs:157-161) as soon as any
 * ray touches the matrix; every shape's intersect() inverts its own transform for every ray, so a
 * world containing a singular shape/group/camera transform fails on its first pixel. */
static int world_check_invertible(const orc_world *w) {
    if (w->has_camera && !w->cam_invertible) return 0;
    for (int i = 0; i < w->nshapes; i++)
        if (!w->shapes[i].invertible) return 0;
    for (int i = 0; i < w->npatterns; i++)
        if (!w->patterns[i].invertible) return 0;
    return 1;
}

typedef struct {
    orc_world *w;
    double *out;
    uint32_t row_begin, row_end, row_step, col_step;
    volatile uint32_t *next_row;
    pthread_mutex_t *mu;
    counters_t counters;
} job_t;

static void *render_worker(void *arg) {
    job_t *j = (job_t *)arg;
    orc_world *w = j->w;
    tl_counters = &j->counters;
    for (;;) {
        pthread_mutex_lock(j->mu);
        uint32_t y = *j->next_row;
        *j->next_row = y + j->row_step;
        pthread_mutex_unlock(j->mu);
        if (y >= j->row_end) break;
        /* camera.rs:78: x in 0..hsize-1 (Q1) */
        for (uint32_t x = 0; x + 1 < w->hsize; x += j->col_step) {
            v3 c = pixel_color(w, x, y, &j->counters);
            double *px = j->out + ((size_t)y * w->hsize + x) * 3;
            px[0] = c.x, px[1] = c.y, px[2] = c.z;
        }
    }
    tl_counters = NULL;
    return NULL;
}

int orc_render_subset(orc_world *w, int nthreads, uint32_t row_begin, uint32_t row_end, uint32_t row_step,
                      uint32_t col_step, double *rgb_out, orc_stats *stats) {
    if (!w->has_camera || w->hsize == 0 || w->vsize == 0 || row_step == 0 || col_step == 0) return ORC_ERR_INVALID;
    if (!world_check_invertible(w)) return ORC_ERR_NON_INVERTIBLE;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    /* camera.rs:72: y in 0..vsize-1 (Q1) */
    if (row_end > w->vsize - 1) row_end = w->vsize - 1;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    memset(rgb_out, 0, sizeof(double) * 3 * (size_t)w->hsize * w->vsize); /* Canvas::new: all black */
    volatile uint32_t next_row = row_begin;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_t th[256];
    job_t jobs[256];
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, ORC_WORKER_STACK);
    for (int i = 0; i < nthreads; i++) {
        memset(&jobs[i], 0, sizeof(job_t));
        jobs[i].w = w;
        jobs[i].out = rgb_out;
        jobs[i].row_begin = row_begin;
        jobs[i].row_end = row_end;
        jobs[i].row_step = row_step;
        jobs[i].col_step = col_step;
        jobs[i].next_row = &next_row;
        jobs[i].mu = &mu;
        pthread_create(&th[i], &attr, render_worker, &jobs[i]);
    }
    pthread_attr_destroy(&attr);
    counters_t total = {0, 0, 0, 0};
    for (int i = 0; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        total.primary += jobs[i].counters.primary;
        total.shadow += jobs[i].counters.shadow;
        total.secondary += jobs[i].counters.secondary;
        total.diverged += jobs[i].counters.diverged;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (stats) {
        stats->primary_rays = total.primary;
        stats->shadow_rays = total.shadow;
        stats->secondary_rays = total.secondary;
        stats->diverged_pixels = total.diverged;
        stats->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    }
    return ORC_OK;
}

/* The same loop body (camera.rs:78-84: ray_for_pixel, color_at(ray, 5)) for an explicit list of pixels: test
 * infrastructure for parity checks of large frames on seeded-random or hand-picked pixels.  xy holds n (x, y) pairs;
 * pixels of the last row / column (never rendered, Q1) come back black.  rgb_out: 3 doubles per listed pixel. */
typedef struct {
    orc_world *w;
    const uint32_t *xy;
    double *out;
    size_t n;
    volatile size_t *next;
    pthread_mutex_t *mu;
    counters_t counters;
} pjob_t;

static void *pixels_worker(void *arg) {
    pjob_t *j = (pjob_t *)arg;
    orc_world *w = j->w;
    tl_counters = &j->counters;
    for (;;) {
        pthread_mutex_lock(j->mu);
        size_t i = *j->next;
        *j->next = i + 1;
        pthread_mutex_unlock(j->mu);
        if (i >= j->n) break;
        const uint32_t x = j->xy[2 * i], y = j->xy[2 * i + 1];
        double *px = j->out + 3 * i;
        px[0] = px[1] = px[2] = 0.;
        if (x + 1 >= w->hsize || y + 1 >= w->vsize) continue; /* camera.rs:72,78 (Q1) */
        v3 c = pixel_color(w, x, y, &j->counters);
        px[0] = c.x, px[1] = c.y, px[2] = c.z;
    }
    tl_counters = NULL;
    return NULL;
}

int orc_render_pixels(orc_world *w, int nthreads, size_t n, const uint32_t *xy, double *rgb_out, orc_stats *stats) {
    if (!w->has_camera || w->hsize == 0 || w->vsize == 0 || (n && (!xy || !rgb_out))) return ORC_ERR_INVALID;
    if (!world_check_invertible(w)) return ORC_ERR_NON_INVERTIBLE;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    volatile size_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_t th[256];
    pjob_t jobs[256];
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, ORC_WORKER_STACK);
    for (int i = 0; i < nthreads; i++) {
        memset(&jobs[i], 0, sizeof(pjob_t));
        jobs[i].w = w, jobs[i].xy = xy, jobs[i].out = rgb_out, jobs[i].n = n;
        jobs[i].next = &next, jobs[i].mu = &mu;
        pthread_create(&th[i], &attr, pixels_worker, &jobs[i]);
    }
    pthread_attr_destroy(&attr);
    counters_t total = {0, 0, 0, 0};
    for (int i = 0; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        total.primary += jobs[i].counters.primary;
        total.shadow += jobs[i].counters.shadow;
        total.secondary += jobs[i].counters.secondary;
        total.diverged += jobs[i].counters.diverged;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (stats) {
        stats->primary_rays = total.primary;
        stats->shadow_rays = total.shadow;
        stats->secondary_rays = total.secondary;
        stats->diverged_pixels = total.diverged;
        stats->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    }
    return ORC_OK;
}

/* camera.rs:64-93 */
int orc_render(orc_world *w, int nthreads, double *rgb_out, orc_stats *stats) {
    return orc_render_subset(w, nthreads, 0, w->vsize, 1, 1, rgb_out, stats);
}

/* ------------------------------------------------------------------ unit-level entry points */

static v3 in3(const double p[3]) { return v3_new(p[0], p[1], p[2]); }
/* a unit-level call whose ray tree tripped the nesting guard answers NaN, like a pixel */
static v3 unit_result(v3 c) {
    if (tl_diverged) c = v3_new(NAN, NAN, NAN);
    tl_diverged = 0;
    return c;
}
static void out3(v3 a, double o[3]) { o[0] = a.x, o[1] = a.y, o[2] = a.z; }

int orc_ray_for_pixel(orc_world *w, uint32_t px, uint32_t py, double origin[3], double direction[3]) {
    if (!w->has_camera) return ORC_ERR_INVALID;
    if (!w->cam_invertible) return ORC_ERR_NON_INVERTIBLE;
    ray_t r = camera_ray_for_pixel(w, px, py);
    out3(r.origin, origin);
    out3(r.direction, direction);
    return ORC_OK;
}

static int copy_out(const xs_t *xs, int cap, double *ts, int *shapes, double *us, double *vs) {
    for (int i = 0; i < xs->n && i < cap; i++) {
        if (ts) ts[i] = xs->data[i].t;
        if (shapes) shapes[i] = xs->data[i].shape;
        if (us) us[i] = xs->data[i].u;
        if (vs) vs[i] = xs->data[i].v;
    }
    return xs->n;
}

int orc_intersect(orc_world *w, int shape, const double origin[3], const double direction[3], int cap, double *ts,
                  int *shapes, double *us, double *vs) {
    if (!world_check_invertible(w)) return -ORC_ERR_NON_INVERTIBLE;
    ray_t r = {in3(origin), in3(direction)};
    xs_t xs;
    xs_init(&xs);
    if (shape < 0)
        world_intersect(w, &r, &xs);
    else
        shape_intersect(w, shape, &r, &xs);
    int n = copy_out(&xs, cap, ts, shapes, us, vs);
    xs_free(&xs);
    return n;
}

int orc_local_intersect(orc_world *w, int shape, const double origin[3], const double direction[3], int cap,
                        double *ts) {
    ray_t r = {in3(origin), in3(direction)};
    xs_t xs;
    xs_init(&xs);
    shape_local_intersect(w, shape, &r, &xs);
    int n = copy_out(&xs, cap, ts, NULL, NULL, NULL);
    xs_free(&xs);
    return n;
}

int orc_local_normal_at(orc_world *w, int shape, const double p[3], double u, double v, double out[3]) {
    out3(shape_local_normal_at(w, shape, in3(p), u, v), out);
    return ORC_OK;
}
int orc_normal_at(orc_world *w, int shape, const double p[3], double u, double v, double out[3]) {
    if (!world_check_invertible(w)) return ORC_ERR_NON_INVERTIBLE;
    out3(shape_normal_at(w, shape, in3(p), u, v), out);
    return ORC_OK;
}
int orc_world_to_object(orc_world *w, int shape, const double p[3], double out[3]) {
    out3(shape_world_to_object(w, shape, in3(p)), out);
    return ORC_OK;
}
int orc_normal_to_world(orc_world *w, int shape, const double n[3], double out[3]) {
    out3(shape_normal_to_world(w, shape, in3(n)), out);
    return ORC_OK;
}
int orc_csg_intersection_allowed(int operation, int left_hit, int in_left, int in_right) {
    return csg_intersection_allowed(operation, left_hit, in_left, in_right);
}
int orc_csg_filter(orc_world *w, int csg, int n, const double *ts, const int *shapes, int *kept) {
    if (csg < 0 || csg >= w->nshapes || w->shapes[csg].kind != ORC_CSG) return -ORC_ERR_INVALID;
    xs_t xs;
    xs_init(&xs);
    for (int i = 0; i < n; i++) xs_push(&xs, ts[i], shapes[i], (double)i, 0.); /* u carries the input position */
    csg_filter(w, csg, &xs, 0);
    const int m = xs.n;
    for (int i = 0; i < m; i++) kept[i] = (int)xs.data[i].u;
    xs_free(&xs);
    return m;
}
int orc_shape_includes(orc_world *w, int shape, int object) {
    if (shape < 0 || shape >= w->nshapes || object < 0 || object >= w->nshapes) return -ORC_ERR_INVALID;
    return shape_includes(w, shape, object);
}

int orc_color_at(orc_world *w, const double origin[3], const double direction[3], int remaining, double rgb[3]) {
    if (!world_check_invertible(w)) return ORC_ERR_NON_INVERTIBLE;
    ray_t r = {in3(origin), in3(direction)};
    tl_nesting = 0, tl_diverged = 0;
    out3(unit_result(world_color_at(w, &r, remaining)), rgb);
    return ORC_OK;
}
int orc_is_shadowed(orc_world *w, const double point[3], const double light_position[3], int *out) {
    if (!world_check_invertible(w)) return ORC_ERR_NON_INVERTIBLE;
    *out = world_is_shadowed(w, in3(point), in3(light_position));
    return ORC_OK;
}
int orc_light_intensity_at(orc_world *w, int light, const double point[3], double *out) {
    if (light < 0 || light >= w->nlights) return ORC_ERR_INVALID;
    *out = light_intensity_at(w, &w->lights[light], in3(point));
    return ORC_OK;
}
int orc_area_light_point(orc_world *w, int light, uint32_t u, uint32_t v, double out[3]) {
    if (light < 0 || light >= w->nlights || !w->lights[light].is_area) return ORC_ERR_INVALID;
    out3(area_point_on_light(&w->lights[light], u, v), out);
    return ORC_OK;
}
int orc_light_info(orc_world *w, int light, double position[3], double uvec[3], double vvec[3], uint32_t *samples) {
    if (light < 0 || light >= w->nlights) return ORC_ERR_INVALID;
    const light_t *l = &w->lights[light];
    out3(l->position, position);
    out3(l->uvec, uvec);
    out3(l->vvec, vvec);
    *samples = l->samples;
    return ORC_OK;
}
int orc_lighting(orc_world *w, int material, int shape, int light, const double point[3], const double eye[3],
                 const double normal[3], double intensity, double rgb[3]) {
    if (material < 0 || material >= w->nmaterials || light < 0 || light >= w->nlights) return ORC_ERR_INVALID;
    out3(material_lighting(w, &w->materials[material], shape, &w->lights[light], in3(point), in3(eye), in3(normal),
                           intensity),
         rgb);
    return ORC_OK;
}
int orc_pattern_at(orc_world *w, int pattern, const double p[3], double rgb[3]) {
    if (pattern < 0 || pattern >= w->npatterns) return ORC_ERR_INVALID;
    out3(pattern_at(w, pattern, in3(p)), rgb);
    return ORC_OK;
}
int orc_pattern_at_shape(orc_world *w, int pattern, int shape, const double p[3], double rgb[3]) {
    if (pattern < 0 || pattern >= w->npatterns || shape < 0 || shape >= w->nshapes) return ORC_ERR_INVALID;
    out3(pattern_at_shape(w, pattern, shape, in3(p)), rgb);
    return ORC_OK;
}

static void comps_pack(const comps_t *c, double *o) {
    o[0] = c->t;
    out3(c->point, o + 1);
    out3(c->eye_vector, o + 4);
    out3(c->normal_vector, o + 7);
    o[10] = c->inside ? 1. : 0.;
    out3(c->over_point, o + 11);
    out3(c->under_point, o + 14);
    out3(c->reflect_vector, o + 17);
    o[20] = c->has_n ? c->n1 : NAN;
    o[21] = c->has_n ? c->n2 : NAN;
    o[22] = (double)c->object;
    o[23] = c->u;
    o[24] = c->v;
    o[25] = 0.;
    o[26] = 0.;
}
static comps_t comps_unpack(const double *o) {
    comps_t c;
    memset(&c, 0, sizeof c);
    c.t = o[0];
    c.point = in3(o + 1);
    c.eye_vector = in3(o + 4);
    c.normal_vector = in3(o + 7);
    c.inside = o[10] != 0.;
    c.over_point = in3(o + 11);
    c.under_point = in3(o + 14);
    c.reflect_vector = in3(o + 17);
    c.has_n = !(isnan(o[20]) || isnan(o[21]));
    c.n1 = o[20];
    c.n2 = o[21];
    c.object = (int)o[22];
    c.u = o[23];
    c.v = o[24];
    return c;
}

int orc_prepare_computations(orc_world *w, const double origin[3], const double direction[3], int n,
                             const double *ts, const int *shapes, const double *us, const double *vs,
                             int hit_idx, double *comps_out) {
    if (hit_idx < 0 || hit_idx >= n) return ORC_ERR_INVALID;
    if (!world_check_invertible(w)) return ORC_ERR_NON_INVERTIBLE;
    isect_t *xs = (isect_t *)malloc(sizeof(isect_t) * (size_t)n);
    for (int i = 0; i < n; i++) {
        xs[i].t = ts[i];
        xs[i].shape = shapes[i];
        xs[i].u = us ? us[i] : NAN;
        xs[i].v = vs ? vs[i] : NAN;
    }
    ray_t r = {in3(origin), in3(direction)};
    comps_t c = prepare_computations(w, &xs[hit_idx], &r, xs, n);
    free(xs);
    comps_pack(&c, comps_out);
    return ORC_OK;
}
int orc_shade_hit(orc_world *w, const double *comps, int remaining, double rgb[3]) {
    comps_t c = comps_unpack(comps);
    tl_nesting = 0, tl_diverged = 0;
    out3(unit_result(world_shade_hit(w, &c, remaining)), rgb);
    return ORC_OK;
}
int orc_reflected_color(orc_world *w, const double *comps, int remaining, double rgb[3]) {
    comps_t c = comps_unpack(comps);
    tl_nesting = 0, tl_diverged = 0;
    out3(unit_result(world_reflected_color(w, &c, remaining)), rgb);
    return ORC_OK;
}
int orc_refracted_color(orc_world *w, const double *comps, int remaining, double rgb[3]) {
    comps_t c = comps_unpack(comps);
    tl_nesting = 0, tl_diverged = 0;
    out3(unit_result(world_refracted_color(w, &c, remaining)), rgb);
    return ORC_OK;
}
double orc_schlick(const double *comps) {
    comps_t c = comps_unpack(comps);
    return comps_schlick(&c);
}

/* ------------------------------------------------------------------ canvas.rs */

/* canvas.rs:86-89: (val * 255).round() as u8 -- round half away from zero; `as u8` saturates,
 * NaN -> 0 */
uint8_t orc_scale_and_clamp(double val) {
    double s = round(val * 255.);
    if (isnan(s)) return 0;
    if (s <= 0.) return 0;
    if (s >= 255.) return 255;
    return (uint8_t)s;
}

/* canvas.rs:44-82 */
size_t orc_canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap) {
    size_t pos = 0;
    char tmp[64];
#define EMIT(str, len)                                        \
    do {                                                      \
        size_t l_ = (len);                                    \
        if (out && pos + l_ <= cap) memcpy(out + pos, (str), l_); \
        pos += l_;                                            \
    } while (0)
    int n = snprintf(tmp, sizeof tmp, "P3\n%u %u\n255", width, height);
    EMIT(tmp, (size_t)n);
    for (uint32_t y = 0; y < height; y++) {
        EMIT("\n", 1);
        size_t num_chars_line = 0;
        int line_empty = 1, last_was_newline = 0;
        for (uint32_t i = 0; i < width * 3; i++) {
            uint8_t val = orc_scale_and_clamp(rgb[((size_t)y * width) * 3 + i]);
            if (num_chars_line + 4 > 70) {
                EMIT("\n", 1);
                num_chars_line = 0;
                last_was_newline = 1;
                line_empty = 0;
            }
            if (!line_empty && !last_was_newline) {
                EMIT(" ", 1);
                num_chars_line += 1;
            }
            n = snprintf(tmp, sizeof tmp, "%u", (unsigned)val);
            EMIT(tmp, (size_t)n);
            line_empty = 0;
            last_was_newline = 0;
            num_chars_line += 3;
        }
    }
    EMIT("\n", 1);
#undef EMIT
    return pos;
}
