// rtc_kernels.hip -- the per-pixel render path as one iterative HIP megakernel for gfx950 (MI355X).
//
// Replaces (reference, ray_tracer/src): camera.rs:41-93 (ray_for_pixel, render), world.rs:30-147 (color_at,
// shade_hit, is_shadowed, reflected_color, refracted_color), world.rs:176-186 + shape.rs local_intersect
// functions, intersection.rs:66-90,123-219 (schlick, prepare_computations, hit), material.rs:20-75 (lighting),
// light.rs:49-125.
//
// Design:
//   * one lane = one pixel, one wave = one 8x8 pixel tile, FP64 throughout, compiled with -ffp-contract=off so
//     every a*b+c rounds twice exactly like the reference's (never contracted) Rust arithmetic;
//   * the recursion of color_at is an iterative loop: a ray carries a scalar throughput weight, the reflect
//     child (remaining-1) is followed immediately, the refract child (remaining unchanged, world.rs:142) is
//     pushed on a 5-entry per-lane stack -- at most one pending entry per distinct `remaining` value;
//   * closest hit = linear scan of the unbounded primitives + ordered traversal of a host-built binary BVH
//     whose per-lane stack lives in LDS ([entry][lane] layout, bank-conflict free);
//   * per lane the kernel is a state machine (closest hit -> container pass -> shadow rays -> shade) around a
//     single inlined copy of that intersection loop, which keeps the code inside the instruction cache and
//     lets lanes that trace different kinds of rays share instructions;
//   * n1/n2 (intersection.rs:150-179) are computed by a second pass over the whole line t in (-inf, t_hit)
//     in streaming form: the container list's last element is the primitive with an odd number of crossings
//     before the hit whose last crossing is latest (see container_visit()).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <cstdlib>

#include "rtc_scene.h"
#include "rtc_types.h"

// Compiled with -ffp-contract=off (Makefile): the reference's Rust arithmetic is never contracted to FMA.  (Rounds 1-2
// also built a `fast` flavour with contraction and shared reciprocals: 2 % faster, and pixels behind glass or mirrors
// differed by up to 95/255 -- a grazing ray that lands on the other side of an edge -- so it was removed in round 3.)

namespace rtc {

#define RTC_EPS 1e-5  // lib.rs:24
#define RTC_WAVE 64
#define RTC_BVH_STACK 96  // a four-wide node pushes up to three references: 3 x depth in the worst case
#define RTC_MAX_DEPTH_STACK 8
#ifndef RTC_MAX_RAYS_PER_PIXEL
#define RTC_MAX_RAYS_PER_PIXEL 65536
#endif
#define RTC_NO_NODE ((int)0x80000000)  // "no node in hand": the next step pops the stack
#define RTC_DONE ((int)0x80000001)     // deferred leaves: the stack is empty, the run in hand is all that is left of the pass
#ifndef RTC_DEFER_LEAVES
#define RTC_DEFER_LEAVES 1
#endif
#ifndef RTC_STEPS_PER_CHECK
#define RTC_STEPS_PER_CHECK 8         // traversal steps between two looks at the wave's occupancy
#endif
#ifndef RTC_NODES_PER_STEP
#define RTC_NODES_PER_STEP 2          // BVH nodes one step may descend through
#endif
#ifndef RTC_NODES4_PER_STEP
#define RTC_NODES4_PER_STEP 1         // the same for the four-wide nodes of large hierarchies
#endif
#define RTC_SHARDS 8u  // tile counters of the persistent grid
#ifndef RTC_STAY_MIN
#define RTC_STAY_MIN 24  // waiting lanes (finished pass / free) at which a wave leaves the traversal loop to serve them
#endif
#ifndef RTC_REFILL_MIN
#define RTC_REFILL_MIN 1  // free lanes needed before a wave assigns new pixels
#endif
#ifndef RTC_LEAF_DEN
#define RTC_LEAF_DEN 3  // quantised-node scenes; 64 = primitive tests never wait
#endif
#ifndef RTC_KILL_DEAD
#define RTC_KILL_DEAD 1  // end the live ranges of the traversal's per-pass state across the state machine (see there)
#endif

struct V3 {
    double x, y, z;
};
struct Ray {
    V3 o, d;
};

#define DEV __device__ __forceinline__

DEV V3 mk(double x, double y, double z) {
    V3 r;
    r.x = x, r.y = y, r.z = z;
    return r;
}
DEV V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
DEV V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }  // color.rs:83-93
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // tuple.rs:110-112
DEV V3 cross(V3 a, V3 b) {                                               // tuple.rs:114-120
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DEV double magnitude(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // tuple.rs:96-99
DEV V3 normalize(V3 a) {  // tuple.rs:101-108: three divisions
    const double m = magnitude(a);
    return mk(a.x / m, a.y / m, a.z / m);
}
// the two roots (-b -+ sq) / (2a) of the quadrics (shape.rs:231-234, 917-918, 1238-1239)
DEV void quadric_roots(double a, double b, double sq, double &t0, double &t1) {
    t0 = (-b - sq) / (2. * a), t1 = (-b + sq) / (2. * a);
}
DEV V3 reflect(V3 v, V3 n) {  // tuple.rs:122-124: self - (normal * 2.) * self.dot(normal)
    const double d = dot(v, n);
    return v - (n * 2.) * d;
}
DEV bool approx_zero(double a) { return fabs(a - 0.) < RTC_EPS; }  // lib.rs:26-29 equal(a, 0.)

// matrix.rs:328-346 / 348-365 on a 3x4 row-major inverse
template <class M>
DEV V3 xf_point(M m, V3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
template <class M>
DEV V3 xf_vector(M m, V3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// transpose(inverse) * n, shape.rs:95
DEV V3 xf_normal(const double *m, V3 n) {
    return mk(m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
              m[2] * n.x + m[6] * n.y + m[10] * n.z);
}

// f64::total_cmp key.  A NaN produced by 0/0 on the reference's x86 host is the negative "default NaN";
// gfx950 produces the positive one, so NaNs are keyed as x86 would order them (below -inf).
DEV long long total_key(double x) {
    if (x != x) return (long long)0x8000000000000001LL;
    long long b = __double_as_longlong(x);
    b ^= (long long)(((unsigned long long)(b >> 63)) >> 1);
    return b;
}
DEV double max3_total(double a, double b, double c) {  // Iterator::max_by(total_cmp): last maximum
    double m = a;
    if (total_key(b) >= total_key(m)) m = b;
    if (total_key(c) >= total_key(m)) m = c;
    return m;
}
DEV double min3_total(double a, double b, double c) {  // Iterator::min_by(total_cmp): first minimum
    double m = a;
    if (total_key(b) < total_key(m)) m = b;
    if (total_key(c) < total_key(m)) m = c;
    return m;
}

// ---------------------------------------------------------------------------------------------
// per-lane trace state and the record visitors
// ---------------------------------------------------------------------------------------------

enum Mode { MODE_CLOSEST = 0, MODE_SHADOW = 1, MODE_CONTAINER = 2 };

// What one pass of the intersection loop accumulates.  The reference materialises and sorts the full
// intersection list (world.rs:176-186); both consumers of that list are order statistics that can be folded
// record by record:
//   hit() (intersection.rs:207-219): smallest t > 0; first record wins on ties within a primitive, the lower
//     flattened index between primitives (stable order, Q15);
//   the containers loop (intersection.rs:150-179), streaming form: per primitive, count the records strictly
//     before the hit and remember the latest of them.  A primitive is "in containers" just before the hit iff
//     that count is odd, and the list's last element is the open primitive whose last record is latest in the
//     (stably) sorted order, i.e. max over (last_t, primitive index).
// Records whose t is NaN fail every comparison below and are dropped (DESIGN.md "NaN records").
struct Trace {
    int mode;
    // MODE_CLOSEST / MODE_SHADOW: smallest qualifying t so far (initially +inf / the light distance) and its
    // primitive (-1: none).  MODE_CONTAINER: t and primitive of the hit (fixed during the pass).
    double lim;
    int prim;
    // MODE_CLOSEST: u, v of the best record.  MODE_CONTAINER: latest crossing among OTHER open primitives /
    // last crossing of the hit primitive.  (The modes are exclusive, so the registers are shared.)
    double a, b;
    int c_best_prim;  // MODE_CONTAINER: open primitive with the latest crossing, -1: none
    int c_hit_open;   // MODE_CONTAINER: the hit primitive has an odd number of records before the hit
    // per-primitive scratch of the container pass
    int n;
    double last;
};

DEV void visit(Trace &T, int pi, double t) {
    if (T.mode == MODE_CONTAINER) {
        if (t < T.lim) {
            if (T.n == 0 || t >= T.last) T.last = t;
            T.n++;
        }
    } else if (t > 0. && (t < T.lim || (t == T.lim && pi < T.prim))) {
        T.lim = t;
        T.prim = pi;
    }
}
DEV void visit_uv(Trace &T, int pi, double t, double u, double v) {
    if (T.mode == MODE_CONTAINER) {
        if (t < T.lim) {
            if (T.n == 0 || t >= T.last) T.last = t;
            T.n++;
        }
    } else if (t > 0. && (t < T.lim || (t == T.lim && pi < T.prim))) {
        T.lim = t;
        T.prim = pi;
        T.a = u;
        T.b = v;
    }
}
// What one primitive test yields when its records are wanted as a list (Csg, below): at most four (cylinder / cone:
// two on the side, two on the caps), in the order the reference pushes them.
struct Collect {
    double t0, t1, t2, t3;
    double u, v;
    int n;
};
DEV void visit(Collect &C, int, double t) {
    if (C.n == 0) C.t0 = t;
    else if (C.n == 1) C.t1 = t;
    else if (C.n == 2) C.t2 = t;
    else C.t3 = t;
    C.n++;
}
DEV void visit_uv(Collect &C, int pi, double t, double u, double v) {
    C.u = u, C.v = v;
    visit(C, pi, t);
}
DEV void end_prim(Collect &, int) {}

DEV void end_prim(Trace &T, int pi) {
    if (T.mode == MODE_CONTAINER) {
        if (T.n & 1) {
            if (pi == T.prim) {
                T.c_hit_open = 1;
                T.b = T.last;
            } else if (T.c_best_prim < 0 || T.last > T.a || (T.last == T.a && pi > T.c_best_prim)) {
                T.a = T.last;
                T.c_best_prim = pi;
            }
        }
        T.n = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// local_intersect functions (object-space ray); records are visited in the order the reference pushes them
// ---------------------------------------------------------------------------------------------

// shape.rs:218-239
template <class TR>
DEV void isect_sphere(const Ray &r, TR &T, int pi) {
    const V3 s2r = r.o - mk(0., 0., 0.);
    const double a = dot(r.d, r.d);
    const double b = 2. * dot(r.d, s2r);
    const double c = dot(s2r, s2r) - 1.;
    const double disc = b * b - 4. * a * c;
    if (disc < 0.) return;
    const double sq = sqrt(disc);
    double t0, t1;
    quadric_roots(a, b, sq, t0, t1);
    visit(T, pi, t0);
    visit(T, pi, t1);
}

// shape.rs:708-720 (Q9: divides by a zero direction on purpose)
DEV void cube_axis(double origin, double direction, double &tmin, double &tmax) {
    double a = (-1. - origin) / direction;
    double b = (1. - origin) / direction;
    if (a > b) {
        const double t = a;
        a = b;
        b = t;
    }
    tmin = a, tmax = b;
}
// shape.rs:665-704
template <class TR>
DEV void isect_cube(const Ray &r, TR &T, int pi) {
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double xmin, xmax, ymin, ymax, zmin, zmax;
    cube_axis(r.o.x, r.d.x, xmin, xmax);
    if (xmin == inf || xmax == -inf) return;
    cube_axis(r.o.y, r.d.y, ymin, ymax);
    if (ymin == inf || ymax == -inf) return;
    cube_axis(r.o.z, r.d.z, zmin, zmax);
    const double tmin = max3_total(xmin, ymin, zmin);
    const double tmax = min3_total(xmax, ymax, zmax);
    if (tmin > tmax) return;
    visit(T, pi, tmin);
    visit(T, pi, tmax);
}

// intersection.rs:34-39 (Q7: `<= radius`, not radius^2)
DEV bool check_cap(const Ray &r, double t, double radius) {
    const double x = r.o.x + t * r.d.x;
    const double z = r.o.z + t * r.d.z;
    return (x * x + z * z) <= radius;
}

// shape.rs:895-959
template <class TR>
DEV void isect_cylinder(const Ray &r, double minimum, double maximum, int closed, TR &T, int pi) {
    const double a = r.d.x * r.d.x + r.d.z * r.d.z;
    if (!approx_zero(a)) {
        const double b = 2. * r.o.x * r.d.x + 2. * r.o.z * r.d.z;
        const double c = r.o.x * r.o.x + r.o.z * r.o.z - 1.;
        const double disc = b * b - 4. * a * c;
        if (disc < 0.) return;  // Q10: caps are not tested either
        const double sq = sqrt(disc);
        double t0, t1;
        quadric_roots(a, b, sq, t0, t1);
        if (t0 > t1) {
            const double t = t0;
            t0 = t1;
            t1 = t;
        }
        const double y0 = r.o.y + t0 * r.d.y;
        if (minimum < y0 && y0 < maximum) visit(T, pi, t0);
        const double y1 = r.o.y + t1 * r.d.y;
        if (minimum < y1 && y1 < maximum) visit(T, pi, t1);
    }
    // intersect_caps, shape.rs:938-959
    if (!closed || approx_zero(r.d.y)) return;
    double t = (minimum - r.o.y) / r.d.y;
    if (check_cap(r, t, 1.)) visit(T, pi, t);
    t = (maximum - r.o.y) / r.d.y;
    if (check_cap(r, t, 1.)) visit(T, pi, t);
}

// shape.rs:1212-1281 (Q8)
template <class TR>
DEV void isect_cone(const Ray &r, double minimum, double maximum, int closed, TR &T, int pi) {
    const double a = r.d.x * r.d.x - r.d.y * r.d.y + r.d.z * r.d.z;
    const double b = 2. * r.o.x * r.d.x - 2. * r.o.y * r.d.y + 2. * r.o.z * r.d.z;
    const double c = r.o.x * r.o.x - r.o.y * r.o.y + r.o.z * r.o.z;
    if (approx_zero(a)) {
        if (!closed) {
            visit(T, pi, -c / (2. * b));  // no y-range check, no b != 0 check
            return;
        }
    } else {
        const double disc = b * b - 4. * a * c;
        if (disc < 0.) return;
        const double sq = sqrt(disc);
        double t0, t1;
        quadric_roots(a, b, sq, t0, t1);
        if (t0 > t1) {
            const double t = t0;
            t0 = t1;
            t1 = t;
        }
        const double y0 = r.o.y + t0 * r.d.y;
        if (minimum < y0 && y0 < maximum) visit(T, pi, t0);
        const double y1 = r.o.y + t1 * r.d.y;
        if (minimum < y1 && y1 < maximum) visit(T, pi, t1);
    }
    // intersect_caps, shape.rs:1260-1281: |min| / |max| for both the plane height and the radius
    if (!closed || approx_zero(r.d.y)) return;
    const double amin = fabs(minimum), amax = fabs(maximum);
    double t = (amin - r.o.y) / r.d.y;
    if (check_cap(r, t, amin)) visit(T, pi, t);
    t = (amax - r.o.y) / r.d.y;
    if (check_cap(r, t, amax)) visit(T, pi, t);
}

// shape.rs:1479-1502 == 1712-1735; a = p1, e1, e2
template <class P, class TR>
DEV void isect_triangle(const Ray &r, P a, TR &T, int pi) {
    const V3 p1 = mk(a[0], a[1], a[2]), e1 = mk(a[3], a[4], a[5]), e2 = mk(a[6], a[7], a[8]);
    const V3 dce2 = cross(r.d, e2);
    const double det = dot(e1, dce2);
    if (fabs(det) < RTC_EPS) return;
    const double f = 1. / det;
    const V3 p1o = r.o - p1;
    const double u = f * dot(p1o, dce2);
    if (!(0. <= u && u <= 1.)) return;
    const V3 oce1 = cross(p1o, e1);
    const double v = f * dot(r.d, oce1);
    if (v < 0. || (u + v) > 1.) return;
    visit_uv(T, pi, f * dot(e2, oce1), u, v);
}

// One primitive against the ray `r` given in the space of the primitive's parent (world space for top-level
// shapes): Shape::intersect = ray.transform(inverse(T_own)) + local_intersect (e.g. shape.rs:254-258).  A plane
// only ever reads the y components of the object-space ray, so only row 1 of the product is formed (the
// arithmetic of that row is unchanged).
template <class Leaf, class TR>
DEV void isect_leaf(const Leaf &L, const Ray &r, TR &T) {
    const int pi = L.prim;
    const auto *m = L.a;  // keeps the record's address space (constant for the lockstep list)
    switch (L.type) {
    case PT_PLANE: {  // shape.rs:510-517
        const double oy = m[4] * r.o.x + m[5] * r.o.y + m[6] * r.o.z + m[7];
        const double dy = m[4] * r.d.x + m[5] * r.d.y + m[6] * r.d.z;
        if (!(fabs(dy) < RTC_EPS)) visit(T, pi, -oy / dy);
        break;
    }
    case PT_SPHERE: {
        Ray q;
        q.o = xf_point(m, r.o), q.d = xf_vector(m, r.d);
        isect_sphere(q, T, pi);
        break;
    }
    case PT_CUBE: {
        Ray q;
        q.o = xf_point(m, r.o), q.d = xf_vector(m, r.d);
        isect_cube(q, T, pi);
        break;
    }
    case PT_CYLINDER: {
        Ray q;
        q.o = xf_point(m, r.o), q.d = xf_vector(m, r.d);
        isect_cylinder(q, L.minimum, L.maximum, L.closed & 1, T, pi);
        break;
    }
    case PT_CONE: {
        Ray q;
        q.o = xf_point(m, r.o), q.d = xf_vector(m, r.d);
        isect_cone(q, L.minimum, L.maximum, L.closed & 1, T, pi);
        break;
    }
    default: isect_triangle(r, m, T, pi); break;
    }
    end_prim(T, pi);
}

// Rust's `%` on f64 is fmod; pattern.rs kinds
DEV V3 pattern_eval(const DScene &s, int pi, V3 p) {
    // Stripe / Checker select a sub-pattern and evaluate it at the same point (pattern.rs:114-121, 380-386);
    // Blended needs both children (pattern.rs:584-591) -> small explicit stack with halving weights.
    V3 acc = mk(0., 0., 0.);
    int st_pat[RTC_MAX_DEPTH_STACK];
    double st_w[RTC_MAX_DEPTH_STACK];
    int sp = 0;
    double w = 1.;
    bool weighted = false;
    for (;;) {
        const DPattern &pt = s.patterns[pi];
        V3 c;
        bool leaf = true;
        switch (pt.kind) {
        case RTC_PAT_STRIPE:
            pi = (fmod(floor(p.x), 2.) == 0.) ? pt.sub_a : pt.sub_b;
            leaf = false;
            break;
        case RTC_PAT_CHECKER:
            pi = (fmod(floor(p.x) + floor(p.y) + floor(p.z), 2.) == 0.) ? pt.sub_a : pt.sub_b;
            leaf = false;
            break;
        case RTC_PAT_BLENDED:
            weighted = true;
            w = w / 2.;
            if (sp < RTC_MAX_DEPTH_STACK) st_pat[sp] = pt.sub_b, st_w[sp] = w, sp++;
            pi = pt.sub_a;
            leaf = false;
            break;
        case RTC_PAT_GRADIENT: {  // pattern.rs:228-231
            const V3 d = ld3(pt.b) - ld3(pt.a);
            c = ld3(pt.a) + d * p.x;
            break;
        }
        case RTC_PAT_RING:  // pattern.rs:297-303
            c = (fmod(floor(sqrt(p.x * p.x + p.z * p.z)), 2.) == 0.) ? ld3(pt.a) : ld3(pt.b);
            break;
        case RTC_PAT_RADIAL: {  // pattern.rs:472-476
            const double dist = sqrt(p.x * p.x + p.z * p.z);
            const double frac = dist - floor(dist);
            c = ld3(pt.a) + (ld3(pt.b) - ld3(pt.a)) * frac;
            break;
        }
        case RTC_PAT_SOLID: c = ld3(pt.a); break;  // pattern.rs:533-535
        default: c = p; break;                     // RTC_PAT_TEST, pattern.rs:669-671
        }
        if (!leaf) continue;
        if (!weighted) return c;
        acc = acc + c * w;
        if (sp == 0) return acc;
        sp--;
        pi = st_pat[sp];
        w = st_w[sp];
    }
}

// Pattern::pattern_at_shape (pattern.rs:25-32; Q13: only the shape's own transform; Solid override :538-540)
DEV V3 pattern_at_shape(const DScene &s, int pat, const DPrim &p, V3 world_point) {
    const DPattern &pt = s.patterns[pat];
    if (pt.kind == RTC_PAT_SOLID) return ld3(pt.a);
    V3 op = world_point;
    if (p.own_xform >= 0) op = xf_point(s.xforms[p.own_xform].inv, op);
    const V3 pp = xf_point(pt.inv, op);
    return pattern_eval(s, pat, pp);
}

// f64::powf (material.rs:67).  Kept out of line on purpose: inlined, the compiler hoists pow's ~13 f64 constants out of
// the ray loop, runs out of registers and parks them in scratch memory at the start of EVERY wave -- 6.6 KB per wave,
// 0.65 GB of write-back per 4K frame of 129 600 short-lived waves (profiles/r1_v5_teapot4k.txt, WRITE_SIZE).
__device__ __attribute__((noinline)) double specular_pow(double base, double exponent) { return pow(base, exponent); }

// Material::lighting, material.rs:20-75
// `lightv` = normalize(light.position - point) (material.rs:37): for a point light that is, bit for bit, the
// direction of the shadow ray just traced (world.rs:79-83), so the caller passes it instead of normalising again.
DEV V3 lighting(const DMaterial &m, V3 color, V3 lightv, V3 light_int, V3 eye, V3 normal, double intensity) {
    const V3 effective = cmul(color, light_int);
    const V3 ambient = effective * m.ambient;
    const double ldn = dot(lightv, normal);
    V3 diffuse = mk(0., 0., 0.), specular = mk(0., 0., 0.);
    if (!(ldn < 0.)) {
        diffuse = effective * m.diffuse * ldn * intensity;
        const V3 rv = reflect(-lightv, normal);
        const double rde = dot(rv, eye);
        if (rde <= 0.) {
            specular = mk(0., 0., 0.) * intensity;
        } else if (rde < m.spec_skip_rde && fabs(light_int.x * m.specular * intensity) <= 1024. * fabs(ambient.x) &&
                   fabs(light_int.y * m.specular * intensity) <= 1024. * fabs(ambient.y) &&
                   fabs(light_int.z * m.specular * intensity) <= 1024. * fabs(ambient.z)) {
            // far from the highlight: rde^shininess < 2^-66, so |specular| < 2^-56 |ambient| in every channel and
            // `ambient + specular` rounds to `ambient` -- the result below is bit-identical without the pow
        } else {
            const double factor = specular_pow(rde, m.shininess);
            specular = light_int * m.specular * factor * intensity;
        }
    }
    return ambient + specular + diffuse;
}

// Computations::schlick, intersection.rs:66-90
DEV double schlick(double n1, double n2, V3 eye, V3 normal) {
    double c = dot(eye, normal);
    if (n1 > n2) {
        const double n = n1 / n2;
        const double sin2_t = (n * n) * (1. - c * c);
        if (sin2_t > 1.) return 1.;
        c = sqrt(1. - sin2_t);
    }
    const double q = (n1 - n2) / (n1 + n2);
    const double r0 = q * q;
    const double x = 1. - c;
    const double x2 = x * x;
    return r0 + (1. - r0) * (x * (x2 * x2));  // powi(5)
}

// ---------------------------------------------------------------------------------------------
// BVH helpers
// ---------------------------------------------------------------------------------------------

// Slab test of a padded world-space box.  This is the build's own culling arithmetic, not reference
// arithmetic: it uses FMA and an approximate reciprocal; the 1e-6 relative padding of the boxes (host) is ten
// orders of magnitude above its rounding error.  fmin/fmax drop NaNs, which keeps the test conservative.
template <class F>
DEV bool slab(const F *bmin, const F *bmax, const V3 &inv_d, const V3 &oi, double lo, double hi, double &tnear) {
    double t1 = fma((double)bmin[0], inv_d.x, oi.x), t2 = fma((double)bmax[0], inv_d.x, oi.x);
    double tn = fmin(t1, t2), tf = fmax(t1, t2);
    t1 = fma((double)bmin[1], inv_d.y, oi.y), t2 = fma((double)bmax[1], inv_d.y, oi.y);
    tn = fmax(tn, fmin(t1, t2)), tf = fmin(tf, fmax(t1, t2));
    t1 = fma((double)bmin[2], inv_d.z, oi.z), t2 = fma((double)bmax[2], inv_d.z, oi.z);
    tn = fmax(tn, fmin(t1, t2)), tf = fmin(tf, fmax(t1, t2));
    // [tn, tf] clipped to the ray segment [lo, hi] (lo <= hi always): one comparison instead of three
    tn = fmax(tn, lo), tf = fmin(tf, hi);
    tnear = tn;
    return tn <= tf;
}
// ---- the traversal's own slab test, in f32 ----
// Box planes are exact in f32 (16-bit cell indices, or f32 planes); what is rounded is the ray's slope a = 1/d (times
// the cell size) and intercept b = (origin of the frame - ray origin)/d, each once from their f64 values, and the
// fma itself: |error of t| <= 2^-23 (qmax |a| + |b|) for every plane of the scene.  That bound e is folded into the
// intercepts once per ray -- the plane that is NEAR for this ray's direction gets b - e, the far one b + e -- so the
// test below is conservative without a single extra instruction: it may visit a box the exact test would skip (by
// ~1e-6 relative, like the boxes' own padding), never the other way round.  The f64 pipe takes 4 cycles per wave
// instruction and ONE wave saturates it; these f32 instructions take 2, the two planes of an axis share one
// v_pk_fma_f32, and max3 / min3 fold three axes at once (tools/micro/valu_rates.hip has the measured rates).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef RTC_SEG_CACHE
#define RTC_SEG_CACHE 1
#endif
#ifndef RTC_SLAB_PERM
#define RTC_SLAB_PERM 1  // quantised nodes: near / far plane of an axis picked by the ray's sign (v_perm_b32), no min / max
#endif
struct SlabRay {
    float ax, ay, az;    // slope per axis
    f32x2 bx, by, bz;    // intercepts of the (low, high) plane of an axis, widened by the error bound
                         // (RTC_SLAB_PERM, quantised nodes: of the (near, far) plane)
#if RTC_SLAB_PERM
    unsigned sx, sy, sz; // v_perm_b32 selectors that pair an axis' (near, far) cell indices by the sign of its slope
#endif
#if RTC_SEG_CACHE
    float hi, lim_lo;    // T.lim as the end of the f32 ray segment, rounded outward / inward: kept up to date where T.lim
                         // changes (pass start, leaf tests) instead of being formed from the f64 value in every step
#endif
};
DEV void slab_axis(double a64, double b64, double qmax, float &a, f32x2 &b) {
    a = (float)a64;
    const float bc = (float)b64;
    const float e = ((float)qmax * fabsf(a) + fabsf(bc)) * 0x1p-21f;  // 4x the bound above
    const float lo = bc - e, hi = bc + e;
    b.x = (a >= 0.f) ? lo : hi;  // the low plane is the near one for a >= 0
    b.y = (a >= 0.f) ? hi : lo;
}
// one box: (lo, hi) planes per axis; [lo_t, hi_t] is the ray segment of interest (already rounded outward)
DEV bool slab32(f32x2 px, f32x2 py, f32x2 pz, const SlabRay &R, float lo_t, float hi_t, float &tnear) {
    const f32x2 ax = {R.ax, R.ax}, ay = {R.ay, R.ay}, az = {R.az, R.az};
    const f32x2 tx = __builtin_elementwise_fma(px, ax, R.bx);
    const f32x2 ty = __builtin_elementwise_fma(py, ay, R.by);
    const f32x2 tz = __builtin_elementwise_fma(pz, az, R.bz);
    const float nx = __builtin_fminf(tx.x, tx.y), fx = __builtin_fmaxf(tx.x, tx.y);
    const float ny = __builtin_fminf(ty.x, ty.y), fy = __builtin_fmaxf(ty.x, ty.y);
    const float nz = __builtin_fminf(tz.x, tz.y), fz = __builtin_fmaxf(tz.x, tz.y);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx, ny), nz), lo_t);
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx, fy), fz), hi_t);
    tnear = tn;
    return !(tn > tf);  // (a NaN -- coordinates beyond the range of f32 -- visits the box)
}
// the six planes of a quantised box: d0 = min.x | min.y << 16, d1 = min.z | max.x << 16, d2 = max.y | max.z << 16
DEV bool slab32_q(unsigned d0, unsigned d1, unsigned d2, const SlabRay &R, float lo_t, float hi_t, float &tnear) {
    const f32x2 px = {(float)(d0 & 0xffffu), (float)(d1 >> 16)};
    const f32x2 py = {(float)(d0 >> 16), (float)(d2 & 0xffffu)};
    const f32x2 pz = {(float)(d1 & 0xffffu), (float)(d2 >> 16)};
    return slab32(px, py, pz, R, lo_t, hi_t, tnear);
}
#if RTC_SLAB_PERM
// Quantised nodes, second form.  Which plane of an axis is the near one follows from the sign of the ray's slope alone,
// and a correctly rounded fma is monotonic in the plane's index, so picking by sign gives exactly what sorting the two
// products gave: per axis ONE v_perm_b32 pairs (near, far) out of the node's dwords before the conversions, and the six
// min / max of a box are gone (20 -> 17 instructions).  The intercepts are then those of the (near, far) plane: the error
// bound goes b - e / b + e whatever the sign.
DEV void slab_axis_nf(double a64, double b64, double qmax, float &a, f32x2 &b, unsigned &sel, unsigned pos, unsigned neg) {
    a = (float)a64;
    const float bc = (float)b64;
    const float e = ((float)qmax * fabsf(a) + fabsf(bc)) * 0x1p-21f;
    b.x = bc - e, b.y = bc + e;
    sel = (a >= 0.f) ? pos : neg;  // the low plane is the near one for a >= 0
}
// min.x = d0[15:0], max.x = d1[31:16]; min.y = d0[31:16], max.y = d2[15:0]; min.z = d1[15:0], max.z = d2[31:16]
DEV bool slab32_qs(unsigned d0, unsigned d1, unsigned d2, const SlabRay &R, float lo_t, float hi_t, float &tnear) {
    const unsigned qx = __builtin_amdgcn_perm(d1, d0, R.sx);  // near.x | far.x << 16
    const unsigned qy = __builtin_amdgcn_perm(d2, d0, R.sy);
    const unsigned qz = __builtin_amdgcn_perm(d2, d1, R.sz);
    const f32x2 px = {(float)(qx & 0xffffu), (float)(qx >> 16)};
    const f32x2 py = {(float)(qy & 0xffffu), (float)(qy >> 16)};
    const f32x2 pz = {(float)(qz & 0xffffu), (float)(qz >> 16)};
    const f32x2 ax = {R.ax, R.ax}, ay = {R.ay, R.ay}, az = {R.az, R.az};
    const f32x2 tx = __builtin_elementwise_fma(px, ax, R.bx);
    const f32x2 ty = __builtin_elementwise_fma(py, ay, R.by);
    const f32x2 tz = __builtin_elementwise_fma(pz, az, R.bz);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tx.x, ty.x), tz.x), lo_t);
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(tx.y, ty.y), tz.y), hi_t);
    tnear = tn;
    return !(tn > tf);
}
#define SLABQ slab32_qs
#else
#define SLABQ slab32_q
#endif
// T.lim (> 0 or +inf) as the end of the f32 ray segment: rounded away from / towards zero
DEV float f32_above(double x) { return (float)(x * (1. + 0x1p-22)); }
DEV float f32_below(double x) { return (float)(x * (1. - 0x1p-22)); }

DEV double safe_rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double a = fmin(fabs(r), 1e30);  // d == 0 (or denormal): a huge finite slope keeps fma() NaN-free
    return copysign(a, d);
}

#ifndef RTC_PREFETCH
#define RTC_PREFETCH 0
#endif
#ifndef RTC_RING_PAD
#define RTC_RING_PAD 0
#endif
#ifndef RTC_BRANCHFREE
#define RTC_BRANCHFREE 1  // large hierarchies (incoherent rays): the traversal's exchange with its stack as selects
                          // around LDS accesses instead of branches (stress 4K -4 %; coherent small scenes +3 %: not there)
#endif
// first line of the record a child reference of a quantised binary node points at: another node, or the first
// primitive record of a leaf run
DEV const char *child_line(const DScene &s, int ref) {
    if (ref >= 0) return (const char *)(s.bvh_q + ref);
#if RTC_PREFETCH >= 2
    const int first = (~ref) >> 3;
    return (first & RTC_LEAF_COMPACT) ? (const char *)(s.leaves_s + (first & (RTC_LEAF_COMPACT - 1)))
                                      : (const char *)(s.leaves + first);
#else
    return (const char *)s.bvh_q;  // (leaf runs are not requested ahead: the root node's line is always cached)
#endif
}

struct Counters {
    unsigned long long c[ST_COUNT];
};

struct Pending {
    V3 o, d;
    double w;
    int remaining;
};

// canvas row of this rank's local row l (interleaved row tiles)
DEV unsigned canvas_row(const DRenderParams &p, unsigned l) {
    if (p.tile_count <= 1) return l;
    const unsigned k = l / p.tile_rows;
    return (k * p.tile_count + p.tile_index) * p.tile_rows + (l - k * p.tile_rows);
}

// RTC_TIMING_ONLY (diagnostic build, tools/build_variant.sh): the counting kernel keeps only its per-stage clocks
// (lane 0, one s_memtime per stage) so that the stage shares it reports are those of the plain kernel
#ifdef RTC_TIMING_ONLY
#define CNT(idx, n) do { } while (0)
#else
#define CNT(idx, n)                                   \
    do {                                              \
        if (STATS) cnt.c[idx] += (unsigned long long)(n); \
    } while (0)
#endif

// Ray::transform through a run of ancestor (group) inverses, root first (shape.rs:2035)
DEV Ray through_ancestors(const DScene &s, int off, int len, Ray r) {
    for (int k = 0; k < len; k++) {
        const double *m = s.xforms[s.chains[off + k]].inv;
        Ray q;
        q.o = xf_point(m, r.o);
        q.d = xf_vector(m, r.d);
        r = q;
    }
    return r;
}

// ---- Csg (shape.rs:2146-2291) ----------------------------------------------------------------------------------------
// Csg::intersection_allowed, shape.rs:2186-2197
DEV bool csg_allowed(int op, bool left_hit, bool in_left, bool in_right) {
    if (op == RTC_CSG_UNION) return left_hit ? !in_right : !in_left;
    if (op == RTC_CSG_INTERSECTION) return left_hit ? in_right : in_left;
    return left_hit ? !in_right : in_left;  // Difference
}
struct CsgRec {
    double t, u, v;
    int leaf;  // relative to the tree's first primitive
};
// One whole Csg tree against the world-space ray.  Csg::local_intersect (shape.rs:2227-2235) concatenates the records
// of its operands, sorts them and keeps what filter_intersections (shape.rs:2203-2225) lets through; a parent Csg sees
// only what its operands kept.  Here: every leaf of the tree is tested once (its DLeaf record carries the whole chain
// of transforms from world space: enclosing groups, the Csg nodes' identities, inner groups), the records go into one
// list sorted by (t in total_cmp order, leaf) -- a sub-list of a sorted list is sorted, so one sort serves every node --
// and the nodes' filters run bottom-up over a mask of surviving records.  The survivors then enter the pass's fold
// exactly like the records of any other primitive (they keep their own primitive's identity: containers, tie order).
// Returns false when the ray collected more than RTC_CSG_MAX_RECORDS records.
DEV bool isect_csg(const DScene &s, int tree, const Ray &ray, Trace &T) {
    const DCsgRoot root = s.csg_roots[tree];
    CsgRec recs[RTC_CSG_MAX_RECORDS];
    int n = 0;
    bool ok = true;
    for (int k = 0; k < root.n_prims; k++) {
        const DLeaf &L = s.leaves_all[root.prim_first + k];
        Ray r = ray;
        if (L.anc_len > 0) r = through_ancestors(s, L.anc_off, L.anc_len, ray);
        Collect C;
        C.t0 = C.t1 = C.t2 = C.t3 = 0., C.u = C.v = 0., C.n = 0;
        isect_leaf(L, r, C);
        for (int j = 0; j < C.n && j < 4; j++) {
            const double t = j == 0 ? C.t0 : j == 1 ? C.t1 : j == 2 ? C.t2 : C.t3;
            if (n == RTC_CSG_MAX_RECORDS) {
                ok = false;
                break;
            }
            // insertion into the sorted list: records of one leaf arrive in the reference's push order and equal keys
            // keep their arrival order (leaves are visited in depth-first order, so ties fall by leaf index)
            const long long key = total_key(t);
            int i = n;
            while (i > 0 && total_key(recs[i - 1].t) > key) {
                recs[i] = recs[i - 1];
                i--;
            }
            recs[i].t = t, recs[i].u = C.u, recs[i].v = C.v, recs[i].leaf = k;
            n++;
        }
    }
    unsigned long long alive = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    for (int q = 0; q < root.n_nodes; q++) {
        const DCsgNode nd = s.csg_nodes[root.node_first + q];
        bool in_left = false, in_right = false;
        for (int i = 0; i < n; i++) {
            if (!((alive >> i) & 1ull)) continue;
            const int leaf = recs[i].leaf;
            const bool left_hit = leaf >= nd.l_first && leaf < nd.l_first + nd.l_count;  // self.left.includes(object)
            const bool right_hit = leaf >= nd.r_first && leaf < nd.r_first + nd.r_count;
            if (!left_hit && !right_hit) continue;  // a record of another part of the tree
            if (!csg_allowed(nd.op, left_hit, in_left, in_right)) alive &= ~(1ull << i);
            if (left_hit)
                in_left = !in_left;
            else
                in_right = !in_right;
        }
    }
    for (int k = 0; k < root.n_prims; k++) {
        for (int i = 0; i < n; i++)
            if (((alive >> i) & 1ull) && recs[i].leaf == k) visit_uv(T, root.prim_first + k, recs[i].t, recs[i].u, recs[i].v);
        end_prim(T, root.prim_first + k);
    }
    return ok;
}

// Values that are only touched between intersection passes (the shading context of the current hit and the
// pixel's accumulated colour) are parked in LDS, [slot][lane] layout, instead of occupying VGPRs during the
// register-hungry pass: that is what lets RTC_MIN_WAVES waves share a SIMD.  PARK selects how much is parked.
#ifndef RTC_LDS_STACK
#define RTC_LDS_STACK 16
#endif
#ifndef RTC_ONE_LIGHT_LDS
#define RTC_ONE_LIGHT_LDS 1
#endif
#ifndef RTC_CANVAS_MODE
#define RTC_CANVAS_MODE 0  // persistent grid, where a pixel's colour is accumulated: 0 = the lane's side line, 1 = in the
                           // canvas with atomics (A/B, rejected), 2 = in the canvas, plain while one lane owns the pixel
#endif
#ifndef RTC_STATIC_REGS
#define RTC_STATIC_REGS 0  // A/B only (rejected): static schedule keeps colour / weight / Schlick factor in registers
#endif
#ifndef RTC_LDS_STACK_STATIC
#define RTC_LDS_STACK_STATIC 6  // static schedule: see k_render
#endif
#ifndef RTC_LDS_TOP
#define RTC_LDS_TOP 0
#endif
#ifndef RTC_LDS_TABLES
#define RTC_LDS_TABLES 0
#endif
#define RTC_LDS_TAB_DOUBLES 288  // 2304 bytes: cover.yaml's 19 materials and 2 lights fit
//  // BVH stack entries kept in LDS; deeper entries spill to a per-lane scratch array
enum ParkSlot {
    PK_POINT = 0,   // h_point  (3)
    PK_NORMAL = 3,  // h_normal (3)
    PK_EYE = 6,     // h_eye    (3)
    PK_SURFACE = 9,  // sum of Material::lighting over the lights evaluated so far (3)
    PK_COUNT_SIDE = 12,  // (persistent grid: the three below live in the lane's line of the work buffer, RTC_SIDE_SLOT)
    PK_COLOR = 12,  // accumulated pixel colour (3)
    PK_W = 15,
    PK_KR = 16,  // Schlick reflectance of the hit (1 unless the material is both reflective and transparent)
    PK_COUNT = 17
};
// The persistent grid keeps colour, weight and Schlick factor -- the five parked values that are touched least often -- in
// a line of the work buffer: 12 LDS slots are 10 KB per wave with the traversal stack, which lets 16 waves share a CU
// instead of 12, and the traversal is latency bound (profiles/r2_ab/r2_wf_grid.txt, r2_ab_occ.txt).  The register
// allocation follows: 128 VGPRs, the state machine spills ~50 of them.
#ifndef RTC_MIN_WAVES_PERSISTENT
#define RTC_MIN_WAVES_PERSISTENT (RTC_SIDE_SLOT && RTC_RING_GLOBAL ? 4 : RTC_MIN_WAVES)
#endif
typedef __attribute__((address_space(3))) double lds_f64;  // an LDS pointer the compiler can not mistake for flat
DEV void park3(lds_f64 *base, int slot, V3 v) {
    base[(slot + 0) * RTC_WAVE] = v.x;
    base[(slot + 1) * RTC_WAVE] = v.y;
    base[(slot + 2) * RTC_WAVE] = v.z;
}
DEV V3 unpark3(const lds_f64 *base, int slot) {
    return mk(base[(slot + 0) * RTC_WAVE], base[(slot + 1) * RTC_WAVE], base[(slot + 2) * RTC_WAVE]);
}

// Shape::normal_at (shape.rs:43-47): world_to_object down the chain (shape.rs:82-90), local normal,
// normal_to_world back up with a normalisation per level (shape.rs:94-103).
template <bool GROUPS>
DEV V3 normal_at(const DScene &s, const DPrim &p, V3 world_point, double u, double v) {
    V3 n;
    if (p.type == PT_PLANE) return ld3(p.plane_normal);  // constant per plane: evaluated at commit (rtc_scene.cpp)
    if (p.type >= PT_TRIANGLE) {
        const DTri &tr = s.tris[p.tri];
        if (p.type == PT_TRIANGLE)
            n = ld3(tr.n1);  // shape.rs:1516-1518
        else                 // shape.rs:1749-1754
            n = ld3(tr.n2) * u + ld3(tr.n3) * v + ld3(tr.n1) * (1. - u - v);
        // the triangle's own transformation() is the identity (shape.rs:1550-1552): only that level's
        // normalisation remains
        n = normalize(n);
    } else {
        const double *own = s.xforms[p.own_xform].inv;  // == the chain's last entry, without the look-up
        V3 lp = world_point;
        if (GROUPS)
            for (int k = 0; k < p.chain_len - 1; k++) lp = xf_point(s.xforms[s.chains[p.chain_off + k]].inv, lp);
        switch (p.type) {
        case PT_SPHERE:                            // shape.rs:273-279
            lp = xf_point(own, lp);
            n = lp - mk(0., 0., 0.);
            break;
        case PT_CUBE: {  // shape.rs:743-757
            lp = xf_point(own, lp);
            const double ax = fabs(lp.x), ay = fabs(lp.y), az = fabs(lp.z);
            const double mc = max3_total(ax, ay, az);
            if (mc == ax)
                n = mk(lp.x, 0., 0.);
            else if (mc == ay)
                n = mk(0., lp.y, 0.);
            else
                n = mk(0., 0., lp.z);
            break;
        }
        default: {  // PT_CYLINDER shape.rs:983-998, PT_CONE shape.rs:1302-1314 (Q8: dist < 1)
            lp = xf_point(own, lp);
            const double dist = lp.x * lp.x + lp.z * lp.z;
            if (dist < 1. && lp.y >= p.maximum - RTC_EPS)
                n = mk(0., 1., 0.);
            else if (dist < 1. && lp.y <= p.minimum + RTC_EPS)
                n = mk(0., -1., 0.);
            else if (p.type == PT_CYLINDER)
                n = mk(lp.x, 0., lp.z);
            else {
                double yy = sqrt(lp.x * lp.x + lp.z * lp.z);
                yy = (lp.y > 0.) ? -yy : yy;
                n = mk(lp.x, yy, lp.z);
            }
            break;
        }
        }
        n = normalize(xf_normal(own, n));
    }
    if (GROUPS) {
        const int first_parent = (p.type >= PT_TRIANGLE) ? p.chain_len - 1 : p.chain_len - 2;
        for (int k = first_parent; k >= 0; k--) n = normalize(xf_normal(s.xforms[s.chains[p.chain_off + k]].inv, n));
    }
    return n;
}

// device-scope accesses to the work buffer (coherent between the XCDs' L2s)
DEV unsigned wb_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void wb_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV bool wb_cas(unsigned *p, unsigned expect, unsigned desired) {
    return __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
}
DEV void wb_store64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEV unsigned long long wb_load64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEV unsigned long long f64_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
DEV double bits_f64(unsigned long long v) { return __longlong_as_double((long long)v); }

// value of lane `src` (wave-uniform index) for every lane
DEV unsigned lane_u32(unsigned v, int src) { return (unsigned)__builtin_amdgcn_readlane((int)v, src); }
DEV double lane_f64(double v, int src) {
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    return __hiloint2double(hi, lo);
}

// index of the k-th (0-based) set bit of a wave-uniform mask, k per lane (k < popcount(m))
DEV unsigned kth_set_bit(unsigned long long m, unsigned k) {
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    const unsigned c = (unsigned)__popc(lo);
    unsigned w = k < c ? lo : hi, base = k < c ? 0u : 32u;
    k = k < c ? k : k - c;
#pragma unroll
    for (unsigned sh = 16u; sh != 0u; sh >>= 1) {
        const unsigned n = (unsigned)__popc(w & ((1u << sh) - 1u));
        const bool up = k >= n;
        w = up ? w >> sh : w;
        base += up ? sh : 0u;
        k -= up ? n : 0u;
    }
    return base;
}

// The megakernel.  Per lane it is a small state machine around ONE intersection pass:
//   MODE_CLOSEST   World::color_at's closest hit for a primary / reflect / refract ray
//   MODE_CONTAINER second pass over the same ray for n1/n2 (only for transparent hits)
//   MODE_SHADOW    World::is_shadowed for one light sample
// so lanes tracing different kinds of rays still execute the same instructions.  A pass has two phases:
//   1. the always-tested list (unbounded primitives, or every primitive in brute-force mode): every lane
//      walks it in lockstep, so the records come through the scalar cache into SGPRs and the per-type
//      switch is a scalar branch;
//   2. ordered traversal of the binary BVH with the per-lane stack in LDS; leaf records come through the
//      vector cache, one 128-byte record (one cache line) per test.
// Template flags remove state a scene does not need: GROUPS (any primitive has an ancestor group: ray cache
// in group space, chain walks), PATTERNS (any material has a pattern); QNODES selects the 32-byte quantised nodes
// and the 64-byte sphere leaves of large hierarchies; PERSISTENT the schedule: a grid of resident waves that draw
// tiles from device counters, refill free lanes pixel by pixel and, once the tiles have run out, pass waiting rays
// to free lanes of the same wave and -- through per-wave mailboxes in `work` -- of other waves.
template <bool STATS, bool GROUPS, bool PATTERNS, bool PERSISTENT, bool QNODES>
__global__ __launch_bounds__(RTC_WAVE, PERSISTENT ? RTC_MIN_WAVES_PERSISTENT : RTC_MIN_WAVES) void k_render(DScene s, DRenderParams prm,
                                                                     double *__restrict__ out,
                                                                     unsigned long long *__restrict__ stats,
                                                                     unsigned *__restrict__ work) {
    // BVH stack entries kept in LDS; deeper entries go to a per-lane scratch array.  The static schedule keeps 6: with
    // near-child-first traversal few far children wait at a time even in hierarchies 18 levels deep, and the SIZE of a
    // workgroup's LDS is what counts -- at 12 x 12.8 KB the CU's 160 KB are full to the last 6 KB, waves wait for a hole
    // that fits and the device averages 10.3 instead of 12 resident waves per CU (SQ_WAVE_CYCLES; the same kernel with 4
    // entries and 12 unused ones is as slow as with 16).  10.2 KB: cover 4K 3.68 -> 3.26 ms, cover 1080p 1.58 -> 1.41 ms,
    // 30 000 spheres unchanged (profiles/r3_ab/r3_ab_sstack*.txt, r3_mid_probe.txt).
    constexpr int LDS_STACK = PERSISTENT ? RTC_LDS_STACK : RTC_LDS_STACK_STATIC;
    __shared__ int bvh_stack[LDS_STACK * RTC_WAVE];
#ifdef RTC_LDS_PAD  // A/B only: unused LDS, to tell the effect of a workgroup's LDS SIZE from that of its stack's length
    __shared__ int lds_pad[RTC_LDS_PAD * RTC_WAVE];
    if (threadIdx.x == 0 && prm.depth == 0xffffffffu) lds_pad[0] = 1;
    asm volatile("" ::"v"(&lds_pad[threadIdx.x]) : "memory");
#endif
    constexpr bool SIDE = PERSISTENT && RTC_RING_GLOBAL && RTC_SIDE_SLOT;
    // A/B only (rejected): the static schedule with colour, weight and Schlick factor in registers -- LDS 10.2 -> 7.7 KB, but
    // the ten registers live across the traversal loop cost more than the LDS was worth (cover 4K 3.26 -> 3.67 ms,
    // profiles/r3_ab/r3_ab_regs.txt)
    constexpr bool REGS = !PERSISTENT && RTC_STATIC_REGS;
    // A/B only (rejected, profiles/r3_ab/r3_ab_canvas.txt): the persistent grid ADDS every hit's share `weight x surface`
    // to the canvas with a no-return f64 atomic instead of carrying the pixel's colour in the lane's side line.  Same
    // pixels (one lane's additions reach a pixel in program order), same frame time (995 vs 997 ms) -- and MORE memory
    // traffic, not less: device-scope atomics are carried out on the memory side and each one is counted (and paid for)
    // as a write request of its own (WRITE_SIZE 430 -> 517 GB per frame).
    // Mode 2 (A/B, rejected: 981 vs 972 ms): the pixel's colour is accumulated in the canvas itself -- zeroed when the lane takes the pixel,
    // `canvas += weight x surface` at every hit as an ordinary load / add / store while one lane owns the pixel (the same
    // additions in the same order as an accumulator that travels with the lane), and as atomics once the pixel's ray tree
    // has been split between lanes at the end of the frame (the owner's plain stores are written back before the first
    // hand-over, see the donors' fence).  A tile row's lanes share canvas lines; the side line was one 64-byte line per
    // LANE that every hit read, dirtied and -- L2 being what it is under this kernel -- wrote back.
    constexpr bool CANVAS_ACC = PERSISTENT && RTC_CANVAS_MODE != 0;
    constexpr bool CANVAS_RMW = PERSISTENT && RTC_CANVAS_MODE == 2;
    __shared__ double parked[((SIDE || REGS) ? PK_COUNT_SIDE : PK_COUNT) * RTC_WAVE];
#if RTC_LDS_TOP
    // A/B only ("LDS staging"): the first RTC_LDS_TOP nodes of the breadth-first hierarchy -- its
    // top levels -- copied into LDS by every wave.  LDS is what bounds the occupancy (12.8 KB per wave x 12 waves of the
    // CU's 160 KB), so the copy has to be paid for with traversal-stack entries (build with a smaller RTC_LDS_STACK).
    __shared__ uint4 top_nodes[RTC_LDS_TOP * 2];
    const int n_top = (QNODES && !RTC_BVH4) ? (s.n_bvh_nodes < RTC_LDS_TOP ? s.n_bvh_nodes : RTC_LDS_TOP) : 0;
    if (QNODES)
        for (int i = (int)threadIdx.x; i < n_top * 2; i += RTC_WAVE) top_nodes[i] = ((const uint4 *)s.bvh_q)[i];
#endif
    const unsigned lane = threadIdx.x;
#if RTC_LDS_TABLES
    // A/B only ("light / material tables in LDS"; north_star's staging clause): the material and light
    // tables of a SMALL scene copied into LDS by every wave of the static schedule and read from there by the state
    // machine.  The default reads them through the vector L1 (per-lane indices) -- see the record for why that stays.
    __shared__ double lds_tab[RTC_LDS_TAB_DOUBLES];
    static_assert(sizeof(DMaterial) == 96 && sizeof(DLight) == 136, "table strides of the LDS copy");
    const bool tab_ok = !PERSISTENT && s.n_materials * 12 + s.n_lights * 17 <= RTC_LDS_TAB_DOUBLES;
    const int tab_lights = s.n_materials * 12;
    if (tab_ok) {
        for (int i = (int)lane; i < s.n_materials * 12; i += RTC_WAVE) lds_tab[i] = ((const double *)s.materials)[i];
        for (int i = (int)lane; i < s.n_lights * 17; i += RTC_WAVE) lds_tab[tab_lights + i] = ((const double *)s.lights)[i];
        __syncthreads();
    }
    auto mat_f64 = [&](int i, int k) { return tab_ok ? lds_tab[i * 12 + k] : ((const double *)(s.materials + i))[k]; };
    auto load_material = [&](int i) {
        DMaterial m = DMaterial();
        if (tab_ok) {
            double *md = (double *)&m;
            for (int k = 0; k < 12; k++) md[k] = lds_tab[i * 12 + k];
        } else {
            m = s.materials[i];
        }
        return m;
    };
    auto load_light = [&](int i) {
        DLight l = DLight();
        if (tab_ok) {
            double *ld = (double *)&l;
            for (int k = 0; k < 17; k++) ld[k] = lds_tab[tab_lights + i * 17 + k];
        } else {
            l = s.lights[i];
        }
        return l;
    };
#define RTC_MAT_F64(i, k, field) mat_f64((i), (k))
#define RTC_MAT_CASTS(i) (__double2loint(mat_f64((i), 10)) != 0)
#define RTC_LOAD_MATERIAL(i) load_material(i)
#define RTC_LOAD_LIGHT(i) load_light(i)
#else
#define RTC_MAT_F64(i, k, field) (s.materials[(i)].field)
#define RTC_MAT_CASTS(i) (s.materials[(i)].casts_shadow != 0)
#define RTC_LOAD_MATERIAL(i) (s.materials[(i)])
#define RTC_LOAD_LIGHT(i) (s.lights[(i)])
#endif
    int *const stack = &bvh_stack[lane];  // [entry][lane]: conflict free
    lds_f64 *const pk = (lds_f64 *)parked + lane;
    int deep_stack[RTC_BVH_STACK - LDS_STACK];  // scratch memory: entries beyond the LDS part (rare)
#ifdef RTC_POP_PROBE
    // measurement only (build with -DRTC_BRANCHFREE=0, read with collect_stats): entry distance of every waiting subtree,
    // to count the stack entries that are popped after the ray's limit has moved in front of them.  The counters of the
    // cube / cylinder tests carry the result (the probe is meant for scenes without those).
    float probe_t[RTC_BVH_STACK];
    float probe_push = 0.f;
#endif
    const bool use_bvh = prm.use_bvh != 0;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const unsigned n_tiles = prm.tiles_x * prm.tiles_y;

#ifdef RTC_TIMING_ONLY
    const unsigned long long wave_start_rt = __builtin_amdgcn_s_memrealtime();
#endif
    Counters cnt;
    if (STATS)
        for (int i = 0; i < ST_COUNT; i++) cnt.c[i] = 0;

    // ---- lane <-> pixel binding.  The grid is persistent: a wave draws 8x8 pixel tiles from a device-wide
    // counter and hands the tile's pixels to lanes AS THEY BECOME FREE, so a lane whose pixel needs a deep ray
    // tree does not idle the other 63 (camera.rs:72-85 parallelises over rows with the same dynamic flavour).
    bool alive = false;
    unsigned pix = 0;               // tile * 64 + position in tile
    // static schedule, wave-uniform: this workgroup renders a PART of a heavy tile (32 or 16 pixels); its other lanes take
    // over waiting refracted rays from the first round on (rtc_types.h, RTC_HEAVY_TILE_SCORE)
    bool in_parts = false;
    // wave-uniform: unassigned pixels of the tile in hand.  Wave b starts with tile b (no counter traffic while
    // the grid starts up); tiles from prm.dyn_base (= the grid size) on are drawn from the counters.
    unsigned r_next = (PERSISTENT && prm.tile_perm ? prm.tile_perm[blockIdx.x] : blockIdx.x) * 64u, r_end = r_next + 64u;
    const unsigned n_dynamic = PERSISTENT ? n_tiles - prm.dyn_base : 0u;
    unsigned shard = blockIdx.x % RTC_SHARDS, shards_tried = 0;
    bool exhausted = n_dynamic == 0u;  // wave-uniform: every counter ran past its last tile

    // ---- path state ----
    double mult = 1.;  // STATS only: how many times the reference traces this ray (Q3)
    int remaining = 0;
    // waiting refracted rays: a ring (first entry at pbase) so that the OLDEST one -- the root of the largest subtree
    // still to be traced -- can be handed to a free lane at the end of the frame (see "tail" below)
    Pending pend_private[RTC_MAX_DEPTH_STACK];
    // (persistent grid: the ring lives in the work buffer, one 64-byte line per entry, [wave][lane][entry])
    // (persistent grid: the ring lives in the work buffer, one 64-byte line per entry, [wave][lane][entry])
    Pending *const pend_global =
        (Pending *)((char *)work + RTC_WORK_RING_OFFSET) + ((size_t)blockIdx.x * RTC_WAVE + threadIdx.x) * RTC_LANE_LINES;
    // the lane's side line behind its ring: colour[3], weight, Schlick factor (SIDE), else their LDS slots.  (A/B, round
    // 3: the same values as [value][lane] arrays per wave, 2.5 KB instead of 64 lines -- slower and MORE write traffic,
    // 987 vs 971 ms, 318 vs 269 GB: a lane's three colour values then sit in three lines instead of one.)
    double *const side_base = (double *)(pend_global + RTC_RING_SLOTS);
#define side(k) side_base[(k)]
    V3 reg_color = mk(0., 0., 0.);
    double reg_w = 0., reg_kr = 0.;
    auto get_color = [&]() { return REGS ? reg_color : SIDE ? mk(side(0), side(1), side(2)) : unpark3(pk, PK_COLOR); };
    auto set_color = [&](V3 v) {
        if (REGS)
            reg_color = v;
        else if (SIDE)
            side(0) = v.x, side(1) = v.y, side(2) = v.z;
        else
            park3(pk, PK_COLOR, v);
    };
    // Persistent grid with ONE light (wave-uniform): the sum over the lights never waits in LDS -- Material::lighting and
    // shade_hit's tail run in the same visit of the state machine -- so its three slots are free, and the ray's weight and
    // the hit's Schlick factor live there instead of in the side line: one global line per lane less to read, dirty and
    // write back at every hit (the stress scene has one light; WRITE_SIZE 430 -> see profiles/r3_stress_4k.txt).
    const bool lds_w = SIDE && RTC_ONE_LIGHT_LDS && s.n_lights == 1;
    auto get_w = [&]() {
        return REGS ? reg_w : SIDE ? (lds_w ? (double)pk[(PK_SURFACE + 0) * RTC_WAVE] : side(3)) : (double)pk[PK_W * RTC_WAVE];
    };
    auto set_w = [&](double v) {
        if (REGS)
            reg_w = v;
        else if (SIDE) {
            if (lds_w)
                pk[(PK_SURFACE + 0) * RTC_WAVE] = v;
            else
                side(3) = v;
        } else
            pk[PK_W * RTC_WAVE] = v;
    };
    auto get_kr = [&]() {
        return REGS ? reg_kr : SIDE ? (lds_w ? (double)pk[(PK_SURFACE + 1) * RTC_WAVE] : side(4)) : (double)pk[PK_KR * RTC_WAVE];
    };
    auto set_kr = [&](double v) {
        if (REGS)
            reg_kr = v;
        else if (SIDE) {
            if (lds_w)
                pk[(PK_SURFACE + 1) * RTC_WAVE] = v;
            else
                side(4) = v;
        } else
            pk[PK_KR * RTC_WAVE] = v;
    };
#define pend_at(i) (*((PERSISTENT && RTC_RING_GLOBAL) ? &pend_global[i] : &pend_private[i]))
#if RTC_RING_PAD
    // A/B only ("the ring's write traffic"): every push also stores a second, unused entry, i.e.
    // the ring's store traffic doubles while nothing else changes
    Pending pend_pad[RTC_MAX_DEPTH_STACK];
#endif
    double pend_mult[STATS ? RTC_MAX_DEPTH_STACK : 1];  // STATS only: `mult` of the waiting rays
    int sp = 0, pbase = 0;
    bool shared = false;  // this lane's pixel is being traced by more than one lane: results are ADDED to the canvas
#define RTC_PEND(i) (((i) + pbase) & (RTC_MAX_DEPTH_STACK - 1))
    int rays = 0;
    const double nl = (double)s.n_lights;
    // ---- current ray ----
    Ray ray;
    ray.o = ray.d = mk(0., 0., 0.);
    Trace T;
    T.mode = MODE_CLOSEST;
    T.lim = inf, T.prim = -1, T.a = T.b = 0.;
    T.c_best_prim = -1, T.c_hit_open = 0;
    T.n = 0, T.last = 0.;
    // ---- BVH traversal state of the pass in flight (resumable) ----
    bool in_pass = false;
    int pos = 0, end = 0, sb = 0, cur = RTC_NO_NODE;
#if RTC_PREFETCH
    unsigned pf0 = 0, pf1 = 0;  // words of the records requested ahead of time (never used, see the node visit)
#endif
    SlabRay slab_ray;  // this pass's ray in the slab test's f32 form (persists with the traversal)
    slab_ray.ax = slab_ray.ay = slab_ray.az = 0.f;
    slab_ray.bx = slab_ray.by = slab_ray.bz = (f32x2){0.f, 0.f};
#if RTC_SLAB_PERM
    slab_ray.sx = slab_ray.sy = slab_ray.sz = 0u;
#endif
#if RTC_SEG_CACHE
    slab_ray.hi = slab_ray.lim_lo = 0.f;
#endif
    // The ray in group space is cached per lane across the leaves of a pass when primitives sit in groups.  The
    // persistent schedule is chosen for glass-heavy scenes, where the 13 registers this costs between passes are
    // worth more than the transform it saves: there the ray is transformed per grouped leaf.
    constexpr bool RAY_CACHE = GROUPS && !PERSISTENT;
    int cache_id = -1;  // ancestor chain the cached ray is expressed in
    Ray cache_ray = ray;
    // ---- shading context (valid from the closest hit until the last shadow ray of that hit) ----
    int h_prim = -1, h_mat = 0;  // hit primitive and its material index (read once per hit)
    int li = 0;
    unsigned su = 0, sv = 0, unshadowed = 0;

    // The scene tables are immutable during the launch: reading the lockstep list through the constant address
    // space keeps its loads on the scalar cache even though this kernel also stores (pixels, tile counter).
    typedef const __attribute__((address_space(4))) DLeaf *const_leaf_ptr;
    const const_leaf_ptr lin = (const_leaf_ptr)(use_bvh ? s.leaves : s.leaves_all);
    const int n_lin = use_bvh ? s.n_unbounded : s.n_prims;

    auto pixel_ptr = [&]() {
        const unsigned tile = prm.tile_order ? n_tiles - 1u - (pix >> 6) : (pix >> 6), q = pix & 63u;
        const unsigned x = (tile % prm.tiles_x) * 8u + (q & 7u);
        const unsigned l = (tile / prm.tiles_x) * 8u + (q >> 3);
        return out + ((size_t)l * s.hsize + x) * 3u;
    };
    // binds this lane to pixel `pix`: primary ray and a fresh path state, or a black pixel (Q1 / tile padding)
    auto init_pixel = [&]() {
        const unsigned tile = prm.tile_order ? n_tiles - 1u - (pix >> 6) : (pix >> 6), q = pix & 63u;
        const unsigned x = (tile % prm.tiles_x) * 8u + (q & 7u);
        const unsigned l = (tile / prm.tiles_x) * 8u + (q >> 3);
        const unsigned y = canvas_row(prm, l);
        if (x < s.hsize && l < prm.local_rows) {
            // camera.rs:72,78: rows 0..vsize-1 and columns 0..hsize-1 (exclusive) -- Q1
            if ((x + 1u < s.hsize) && (y + 1u < s.vsize)) {
                // Camera::ray_for_pixel, camera.rs:41-59
                // (the pointer is made opaque here so that the camera is fetched -- through the scalar cache -- when a
                // pixel starts, not hoisted into registers for the whole kernel)
                typedef const __attribute__((address_space(4))) DCamera *const_cam_ptr;
                const DCamera *cam_raw = s.cam;
                asm volatile("" : "+s"(cam_raw));
                const const_cam_ptr cam = (const_cam_ptr)cam_raw;
                const double xo = ((double)x + 0.5) * cam->pixel_size;
                const double yo = ((double)y + 0.5) * cam->pixel_size;
                const double wx = cam->half_width - xo;
                const double wy = cam->half_height - yo;
                const V3 pixel = xf_point(cam->inv, mk(wx, wy, -1.));
                ray.o = xf_point(cam->inv, mk(0., 0., 0.));
                ray.d = normalize(pixel - ray.o);
                if (CANVAS_ACC) {
                    double *px = out + ((size_t)l * s.hsize + x) * 3u;
                    px[0] = 0., px[1] = 0., px[2] = 0.;  // the accumulator of this pixel
                } else {
                    set_color(mk(0., 0., 0.));
                }
                set_w(1.);
                mult = 1.;
                remaining = (int)prm.depth;
                sp = 0, pbase = 0;
                shared = false;
                rays = 1;
                T.mode = MODE_CLOSEST;
                T.lim = inf, T.prim = -1;
                CNT(ST_PRIMARY, 1);
                alive = true;
            } else {
                double *px = out + ((size_t)l * s.hsize + x) * 3u;
                px[0] = 0., px[1] = 0., px[2] = 0.;  // Canvas::new: black
            }
        }
    };
    // the pixel is finished: Canvas::write_pixel (camera.rs:87-91)
    auto write_pixel = [&]() {
        if (CANVAS_ACC) return;  // (every hit's share is in the canvas already)
        const unsigned tile = prm.tile_order ? n_tiles - 1u - (pix >> 6) : (pix >> 6), q = pix & 63u;
        const unsigned x = (tile % prm.tiles_x) * 8u + (q & 7u);
        const unsigned l = (tile / prm.tiles_x) * 8u + (q >> 3);
        const V3 color = get_color();
        double *px = out + ((size_t)l * s.hsize + x) * 3u;
        if (!PERSISTENT && in_parts) {
            // every part of this pixel's ray tree is traced by lanes of THIS wave (the pixel was zeroed when the wave
            // started): the parts are added where the wave's stores went, in the L2 of its own XCD -- workgroup scope,
            // no fence, no write-back
            __hip_atomic_fetch_add(px + 0, color.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(px + 1, color.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(px + 2, color.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (shared) {
            // one of several lanes that traced parts of this pixel's ray tree (the pixel was zeroed when the first
            // part was handed over): the order of these few additions is the only run-to-run freedom of a frame
            atomicAdd(px + 0, color.x), atomicAdd(px + 1, color.y), atomicAdd(px + 2, color.z);
        } else {
            px[0] = color.x, px[1] = color.y, px[2] = color.z;
        }
    };
    // a free lane takes over a waiting ray of another lane (of this wave or of another one): it traces that subtree
    // of the pixel's ray tree and adds its part to the canvas
    auto adopt = [&](unsigned qpix, V3 qo, V3 qd, double qw, double qmult, int qrem) {
        pix = qpix;
        shared = true;
        alive = true;
        ray.o = qo, ray.d = qd, remaining = qrem;
        if (!CANVAS_ACC) set_color(mk(0., 0., 0.));
        set_w(qw);
        if (STATS) mult = qmult;
        sp = 0, pbase = 0;
        rays = 1;
        in_pass = false;
        T.mode = MODE_CLOSEST;
        T.lim = inf, T.prim = -1;
        CNT(ST_SECONDARY, 1);
        CNT(ST_SECONDARY_REF, mult);
    };
    // tail only: the pixel of this lane is about to be shared -- its canvas entry becomes an accumulator
    auto zero_pixel = [&]() {
        const unsigned tile = prm.tile_order ? n_tiles - 1u - (pix >> 6) : (pix >> 6), q = pix & 63u;
        const unsigned x = (tile % prm.tiles_x) * 8u + (q & 7u);
        const unsigned l = (tile / prm.tiles_x) * 8u + (q >> 3);
        double *px = out + ((size_t)l * s.hsize + x) * 3u;
        px[0] = 0., px[1] = 0., px[2] = 0.;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (the stores are in L2 before any lane adds to them)
    };
    auto zero_pixel_local = [&]() {
        const unsigned tile = prm.tile_order ? n_tiles - 1u - (pix >> 6) : (pix >> 6), q = pix & 63u;
        const unsigned x = (tile % prm.tiles_x) * 8u + (q & 7u);
        const unsigned l = (tile / prm.tiles_x) * 8u + (q >> 3);
        double *px = out + ((size_t)l * s.hsize + x) * 3u;
        px[0] = 0., px[1] = 0., px[2] = 0.;
    };
    // a per-lane bound was exceeded: the host finds this launch's id in the flag word and reports RTC_ERR_OVERFLOW
    auto flag_overflow = [&]() { wb_store(work + RTC_WORK_FLAG, prm.launch_id); };
    if (PERSISTENT && prm.share && lane == 0) atomicAdd(work + RTC_WORK_BUSY, 1u);
    unsigned long long *const my_box =
        (unsigned long long *)((char *)work + RTC_WORK_MBOX_BYTES) + (size_t)blockIdx.x * 64u * RTC_MBOX_ITEM_U64;
    unsigned share_round = 0;  // wave-uniform: varies the waves probed for hunger
    bool rendered = false;
    if (!PERSISTENT) {
        // static schedule: workgroup b renders one tile of the list, or one part of one of its first n_heavy tiles
        const unsigned parts = 1u << prm.part_shift, b = blockIdx.x, head = prm.n_heavy << prm.part_shift;
        in_parts = b < head;
        const unsigned entry = in_parts ? b >> prm.part_shift : b - head + prm.n_heavy;
        const unsigned part_lanes = in_parts ? RTC_WAVE / parts : RTC_WAVE;
        pix = (prm.tile_perm ? prm.tile_perm[entry] : entry) * 64u + (in_parts ? (b & (parts - 1u)) * part_lanes : 0u) + lane;
        if (lane < part_lanes) {
            init_pixel();
            if (in_parts && alive) {
                zero_pixel_local();  // (the parts of its ray tree are ADDED to it, see write_pixel)
            }
        }
        rendered = alive;
        if (in_parts) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the zeros are in L2 before any addition)
    }

    // STATS only: stage timing.  WCNT adds a wave-level quantity once (lane 0 carries it).
#define WCNT(idx, n)                                                    \
    do {                                                                \
        if (STATS && lane == 0) cnt.c[idx] += (unsigned long long)(n);  \
    } while (0)
    unsigned long long stamp = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
#define STAGE_END(idx)                                                   \
    do {                                                                 \
        if (STATS) {                                                     \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
            WCNT(idx, now__ - stamp);                                    \
            stamp = now__;                                               \
        }                                                                \
    } while (0)
    // RTC_TIMING_ONLY builds also split the state machine's clock by block; the (otherwise unused) ray counters
    // carry the result: primary = closest-hit block, container = containers block, shadow = light block,
    // secondary = shade_hit tail, box = next-ray / shadow set-up
#ifdef RTC_TIMING_ONLY
    unsigned long long sub_stamp = 0;
#define SUB_BEGIN() do { if (STATS) sub_stamp = __builtin_amdgcn_s_memtime(); } while (0)
    // lanes that take part in a block of the state machine, and how often the block runs (free counters of this build)
#define BLOCK_PROBE(cond, idx_lanes, idx_runs)                                                       \
    do {                                                                                             \
        if (STATS) {                                                                                 \
            const unsigned long long m__ = __ballot(cond);                                           \
            if (m__ && lane == (unsigned)__ffsll((long long)__ballot(true)) - 1u)                    \
                cnt.c[idx_lanes] += __popcll(m__), cnt.c[idx_runs] += 1;                             \
        }                                                                                            \
    } while (0)
#define SUB_END(idx)                                                         \
    do {                                                                     \
        if (STATS) {                                                         \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime();   \
            WCNT(idx, now__ - sub_stamp);                                    \
            sub_stamp = now__;                                               \
        }                                                                    \
    } while (0)
#else
#define SUB_BEGIN() do { } while (0)
#define SUB_END(idx) do { } while (0)
#define BLOCK_PROBE(cond, idx_lanes, idx_runs) do { } while (0)
#endif
    for (;;) {
        // =====================================================================================
        // refill: free lanes take the next pixels (wave-uniform control flow)
        // =====================================================================================
        {
            unsigned long long dead = __ballot(!alive);
            if (PERSISTENT && ((unsigned)__popcll(dead) >= prm.refill_min || dead == ~0ull)) {
                for (;;) {
                    if (r_next == r_end) {
                        if (exhausted) break;
                        // one tile per request from RTC_SHARDS counters (one 128-byte line each): dynamic tile d
                        // belongs to shard d % RTC_SHARDS, a wave starts at its own shard and moves on to the
                        // next one when a shard runs dry, so no word sees more than 1/RTC_SHARDS of the traffic
                        unsigned t = n_tiles;
                        while (shards_tried < RTC_SHARDS) {
                            unsigned k = 0;
                            if (lane == 0) k = atomicAdd(work + shard * 32u, 1u);
                            k = __builtin_amdgcn_readfirstlane(k);
                            const unsigned d = k * RTC_SHARDS + shard;
                            if (d < n_dynamic) {
                                t = prm.dyn_base + d;
                                break;
                            }
                            shard = (shard + 1u) % RTC_SHARDS;
                            shards_tried++;
                        }
                        if (t >= n_tiles) {
                            exhausted = true;
                            break;
                        }
                        if (prm.tile_perm) t = prm.tile_perm[t];  // (heaviest tiles first, see the static schedule)
                        r_next = t * 64u, r_end = r_next + 64u;
                    }
                    dead = __ballot(!alive);
                    if (!dead) break;
                    const unsigned avail = r_end - r_next;
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(dead >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((unsigned)dead, 0u));
                    if (!alive && rank < avail) {
                        pix = r_next + rank;
                        init_pixel();
                    }
                    const unsigned taken = (unsigned)__popcll(dead);
                    r_next += taken < avail ? taken : avail;
                }
            }
            // ---- tail: no tile is left.  A few pixels of the stress scene grow ray trees of 10^4 rays, which one lane
            // traces in ~0.3 s -- a sixth of the whole frame -- while its neighbours sit idle.  Free lanes therefore take
            // over the oldest waiting ray (the largest untraced subtree) of a busy lane of the same wave. ----
            // (prm.share == 0 turns BOTH hand-overs off: every pixel is then traced by one lane from start to end and the
            // frame is bit-reproducible, like the static schedule's)
            if ((PERSISTENT && prm.share && exhausted && r_next == r_end) || (!PERSISTENT && in_parts)) {
                // every free lane is matched with a lane that has a ray waiting -- the k-th free lane with the k-th such
                // lane -- and ALL pairs move at once: the receiver pulls the donor's oldest waiting ray across the wave
                // (ds_bpermute), the donor drops it from its ring
                const unsigned long long recv = __ballot(!alive);
                const bool may_give = alive && sp > 0;
                const unsigned long long don = __ballot(may_give);
                const unsigned n_recv = (unsigned)__popcll(recv), n_don = (unsigned)__popcll(don);
                const unsigned n_pairs = n_recv < n_don ? n_recv : n_don;
                if (PERSISTENT) {
                    // (persistent grid, end of frame: pairs are few and the kernel is held to 128 registers -- one pair
                    // at a time through v_readlane, which keeps the ray in scalar registers on its way)
                    unsigned long long rm = recv, dm = don;
                    while (rm != 0ull && dm != 0ull) {
                        const int dl = __ffsll((long long)dm) - 1, rl = __ffsll((long long)rm) - 1;  // wave-uniform
                        dm &= dm - 1ull, rm &= rm - 1ull;
                        Pending q;
                        q.o = q.d = mk(0., 0., 0.), q.w = 0., q.remaining = 0;
                        double qm = 0.;
                        if ((int)lane == dl) {
                            q = pend_at(pbase);
                            if (STATS) qm = pend_mult[pbase];
                            pbase = (pbase + 1) & (RTC_MAX_DEPTH_STACK - 1);
                            sp--;
                            if (!shared) {
                                if (!CANVAS_ACC) zero_pixel();
                                // (what this lane has stored into the pixel so far is in memory before anyone adds to it)
                                if (CANVAS_RMW) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                                shared = true;
                            }
                        }
                        const unsigned qpix = lane_u32(pix, dl);
                        const int qrem = (int)lane_u32((unsigned)q.remaining, dl);
                        const V3 qo = mk(lane_f64(q.o.x, dl), lane_f64(q.o.y, dl), lane_f64(q.o.z, dl));
                        const V3 qd = mk(lane_f64(q.d.x, dl), lane_f64(q.d.y, dl), lane_f64(q.d.z, dl));
                        const double qw = lane_f64(q.w, dl);
                        const double qmult = STATS ? lane_f64(qm, dl) : 0.;
                        if ((int)lane == rl) adopt(qpix, qo, qd, qw, qmult, qrem);
                    }
                } else if (n_pairs != 0u) {
                    const unsigned rank_r = __builtin_amdgcn_mbcnt_hi((unsigned)(recv >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)recv, 0u));
                    const unsigned rank_d = __builtin_amdgcn_mbcnt_hi((unsigned)(don >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)don, 0u));
                    const bool is_don = may_give && rank_d < n_pairs;
                    const bool is_recv = !alive && rank_r < n_pairs;
                    const int src = is_recv ? (int)kth_set_bit(don, rank_r) : (int)lane;
                    Pending q;
                    q.o = q.d = mk(0., 0., 0.), q.w = 0., q.remaining = 0;
                    double qm = 0.;
                    if (is_don) {
                        q = pend_at(pbase);
                        if (STATS) qm = pend_mult[pbase];
                        pbase = (pbase + 1) & (RTC_MAX_DEPTH_STACK - 1);
                        sp--;
                    }
                    const unsigned qpix = (unsigned)__shfl((int)pix, src, RTC_WAVE);
                    const int qrem = __shfl(q.remaining, src, RTC_WAVE);
                    const V3 qo = mk(__shfl(q.o.x, src, RTC_WAVE), __shfl(q.o.y, src, RTC_WAVE), __shfl(q.o.z, src, RTC_WAVE));
                    const V3 qd = mk(__shfl(q.d.x, src, RTC_WAVE), __shfl(q.d.y, src, RTC_WAVE), __shfl(q.d.z, src, RTC_WAVE));
                    const double qw = __shfl(q.w, src, RTC_WAVE);
                    const double qmult = STATS ? __shfl(qm, src, RTC_WAVE) : 0.;
                    if (is_recv) adopt(qpix, qo, qd, qw, qmult, qrem);
                }
                // ---- and the waves that ran out of rays altogether take over waiting rays of other WAVES: a wave
                // without work publishes its hunger, a wave with waiting rays claims one hungry wave per round
                // (compare-and-swap on that wave's mailbox state) and moves its lanes' oldest waiting rays there. ----
                const unsigned long long don2 = PERSISTENT ? __ballot(alive && sp > 0) : 0ull;
                if (don2 != 0ull) {
                    int target = -1;
                    if (lane == 0 && wb_load(work + RTC_WORK_HUNGRY) != 0u) {
                        const unsigned ht = wb_load(work + RTC_WORK_HINT_TAIL);
                        for (unsigned k = 0; k < 6u && target < 0; k++) {
                            unsigned cand;
                            if (k < 4u) {  // the most recent hints first
                                if (ht <= k) continue;
                                cand = wb_load(work + RTC_WORK_HINTS + ((ht - 1u - k) & 255u));
                            } else {  // hints are a ring of 256: older hungry waves are found by chance
                                cand = (blockIdx.x * 2654435761u + share_round * 40503u + k) % prm.n_waves;
                            }
                            if (cand < prm.n_waves && cand != blockIdx.x &&
                                wb_cas(work + RTC_WORK_STATES + cand, RTC_MB_HUNGRY, RTC_MB_FILLING))
                                target = (int)cand;
                        }
                    }
                    share_round++;
                    target = __builtin_amdgcn_readfirstlane(target);
                    if (target >= 0) {
                        unsigned long long *box = (unsigned long long *)((char *)work + RTC_WORK_MBOX_BYTES) +
                                                  (size_t)target * 64u * RTC_MBOX_ITEM_U64;
                        if (alive && sp > 0) {
                            const unsigned rank = __builtin_amdgcn_mbcnt_hi(
                                (unsigned)(don2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)don2, 0u));
                            const Pending q = pend_at(pbase);
                            const double qm = STATS ? pend_mult[pbase] : 0.;
                            pbase = (pbase + 1) & (RTC_MAX_DEPTH_STACK - 1);
                            sp--;
                            if (!shared) {
                                if (!CANVAS_ACC) zero_pixel();
                                // (what this lane has stored into the pixel so far is in memory before anyone adds to it)
                                if (CANVAS_RMW) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                                shared = true;
                            }
                            unsigned long long *it = box + rank * RTC_MBOX_ITEM_U64;
                            wb_store64(it + 0, f64_bits(q.o.x)), wb_store64(it + 1, f64_bits(q.o.y));
                            wb_store64(it + 2, f64_bits(q.o.z)), wb_store64(it + 3, f64_bits(q.d.x));
                            wb_store64(it + 4, f64_bits(q.d.y)), wb_store64(it + 5, f64_bits(q.d.z));
                            wb_store64(it + 6, f64_bits(q.w)), wb_store64(it + 7, f64_bits(qm));
                            wb_store64(it + 8, ((unsigned long long)pix << 32) | (unsigned)q.remaining);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the rays are visible before the count
                        if (lane == 0) {
                            wb_store(work + RTC_WORK_COUNTS + target, (unsigned)__popcll(don2));
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                            wb_store(work + RTC_WORK_STATES + target, RTC_MB_FULL);
                            atomicSub(work + RTC_WORK_HUNGRY, 1u);
                        }
                    }
                }
            }
            if (__ballot(alive) == 0ull) {
                // no pixel in flight and none left
                if (!(PERSISTENT && prm.share)) break;  // the wave is done
                int got = 0;
                if (lane == 0) {
                    unsigned *my_state = work + RTC_WORK_STATES + blockIdx.x;
                    atomicSub(work + RTC_WORK_BUSY, 1u);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    wb_store(my_state, RTC_MB_HUNGRY);
                    atomicAdd(work + RTC_WORK_HUNGRY, 1u);
                    const unsigned slot = atomicAdd(work + RTC_WORK_HINT_TAIL, 1u);
                    wb_store(work + RTC_WORK_HINTS + (slot & 255u), blockIdx.x);
                    // waits for rays, or for every wave to have run out of its own -- or, with frames in flight
                    // (rtc_render_opts.tail_polls), until its patience ends: prm.share is the number of polls (2^20 when
                    // the wave is to stay until the frame ends: bounded anyway -- a donor between its claim and
                    // RTC_MB_FULL runs straight-line code; the cap only guards against the unforeseen)
                    for (unsigned spin = 0;; spin++) {
                        const unsigned stt = wb_load(my_state);
                        if (stt == RTC_MB_FULL) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                            got = (int)wb_load(work + RTC_WORK_COUNTS + blockIdx.x);
                            break;
                        }
                        if (stt == RTC_MB_HUNGRY && (wb_load(work + RTC_WORK_BUSY) == 0u || spin > prm.share) &&
                            wb_cas(my_state, RTC_MB_HUNGRY, RTC_MB_CLOSED)) {
                            atomicSub(work + RTC_WORK_HUNGRY, 1u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(127);
                    }
                    if (got > 0) atomicAdd(work + RTC_WORK_BUSY, 1u);
                }
                got = __builtin_amdgcn_readfirstlane(got);
                if (got <= 0) break;  // the frame is done
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if ((int)lane < got) {
                    const unsigned long long *it = my_box + lane * RTC_MBOX_ITEM_U64;
                    const V3 qo = mk(bits_f64(wb_load64(it + 0)), bits_f64(wb_load64(it + 1)), bits_f64(wb_load64(it + 2)));
                    const V3 qd = mk(bits_f64(wb_load64(it + 3)), bits_f64(wb_load64(it + 4)), bits_f64(wb_load64(it + 5)));
                    const double qw = bits_f64(wb_load64(it + 6)), qm = bits_f64(wb_load64(it + 7));
                    const unsigned long long pr = wb_load64(it + 8);
                    adopt((unsigned)(pr >> 32), qo, qd, qw, qm, (int)(unsigned)pr);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (the mailbox has been read)
                if (lane == 0) wb_store(work + RTC_WORK_STATES + blockIdx.x, RTC_MB_IDLE);
            }
        }
        STAGE_END(ST_CYC_REFILL);
        if (STATS) {
            const unsigned long long m = __ballot(alive && !in_pass);
            if (m) {
                WCNT(ST_ROUNDS_START, 1);
                WCNT(ST_LANES_START, __popcll(m));
            }
        }
        // =====================================================================================
        // World::intersect (world.rs:176-186) without materialising the list.
        // (A) lanes holding a fresh ray start its pass; (B) the BVH traversal advances in steps for every lane
        // that is inside a pass; (C) lanes whose pass is complete run the state machine.  A traversal is
        // resumable: its state (node in hand, leaf run, stack) lives in registers / LDS across (C), so the wave
        // can leave (B) to serve finished lanes while long rays are still on their way, and re-enter.
        // =====================================================================================
#include "rtc_traverse.inc"
        if (STATS) {
            const unsigned long long m = __ballot(alive && !in_pass);
            if (m) {
                WCNT(ST_ROUNDS_SHADE, 1);
                WCNT(ST_LANES_SHADE, __popcll(m));
            }
        }
        if (alive && !in_pass) {
        // (the parked values are re-read from LDS after a pass; the compiler does not carry them across it --
        // checked in the ISA: no launder is needed, and an asm barrier here would cost the scalar loads above)

        // =====================================================================================
        // state machine
        // =====================================================================================
        bool begin_lights = false, finish = false, next_ray = false;
        double f_refl = 0., f_transp = 0.;  // the hit material's reflectivity / transparency when `finish` is set
        SUB_BEGIN();
        BLOCK_PROBE(T.mode == MODE_CLOSEST && T.prim >= 0, ST_XFORM, ST_SCHLICK);
        BLOCK_PROBE(T.mode == MODE_CONTAINER, ST_REFRACT, ST_SHADOW_SKIPPED);
        if (T.mode == MODE_CLOSEST) {
            // World::color_at, world.rs:30-38
            if (T.prim < 0 || s.n_lights == 0) {
                next_ray = true;
            } else {
                // the whole primitive record is requested at once, and the one material field this block needs
                // before the normal's own look-ups: three dependent round trips (record; transform + material;
                // parents) instead of one per field
                const DPrim p = s.prims[T.prim];
                __builtin_amdgcn_sched_barrier(0);
                h_mat = p.material;
                const double m_transparency = RTC_MAT_F64(h_mat, 8, transparency);
                // Intersection::prepare_computations, intersection.rs:123-148
                h_prim = T.prim;
                const V3 h_point = ray.o + ray.d * T.lim;
                const V3 h_eye = -ray.d;
                V3 n = normal_at<GROUPS>(s, p, h_point, T.a, T.b);
                if (dot(n, h_eye) < 0.) n = -n;  // intersection.rs:134-139
                park3(pk, PK_POINT, h_point);
                park3(pk, PK_EYE, h_eye);
                park3(pk, PK_NORMAL, n);
                CNT(ST_SHADED, 1);
                if (remaining > 0 && m_transparency != 0.) {
                    // containers loop (intersection.rs:150-179) in streaming form: second pass over the same
                    // ray with T.lim / T.prim (the hit) held fixed
                    T.a = 0., T.b = 0., T.c_best_prim = -1, T.c_hit_open = 0;
                    T.mode = MODE_CONTAINER;
                    CNT(ST_CONTAINER, 1);
                    continue;
                }
                begin_lights = true;
            }
        }
        SUB_END(ST_PRIMARY);
        if (T.mode == MODE_CLOSEST) {
        } else if (T.mode == MODE_CONTAINER) {
            const double ri_hit = RTC_MAT_F64(h_mat, 9, refractive_index);
            const double ri_other =
                (T.c_best_prim >= 0) ? RTC_MAT_F64(s.prims[T.c_best_prim].material, 9, refractive_index) : 1.;
            double n1, n2;
            if (T.c_hit_open) {
                // the hit primitive leaves: n1 = last of (others + hit), n2 = last of the others
                const bool hit_is_last =
                    T.c_best_prim < 0 || T.b > T.a || (T.b == T.a && h_prim > T.c_best_prim);
                n1 = hit_is_last ? ri_hit : ri_other;
                n2 = ri_other;
            } else {
                // the hit primitive enters and becomes the last container
                n1 = ri_other;
                n2 = ri_hit;
            }
            // refracted_color (world.rs:108-147) is known as soon as n1/n2 are, and its ray goes to the ring NOW, before
            // the shadow rays of this hit: what the end of a frame waits for is a chain of refractions through glass, and
            // this way a free lane can start on the next link while this one still lights the hit.  (The containers pass
            // ran because remaining > 0 and the material is transparent; ring order and arithmetic are unchanged.)
            {
                const double m_refl = RTC_MAT_F64(h_mat, 7, reflectivity), m_transp = RTC_MAT_F64(h_mat, 8, transparency);
                const V3 h_normal = unpark3(pk, PK_NORMAL);
                const V3 h_eye = unpark3(pk, PK_EYE);
                double kr = 1., kt = 1.;
                if (m_refl > 0. && m_transp > 0.) {
                    const double r = schlick(n1, n2, h_eye, h_normal);
                    CNT(ST_SCHLICK, 1);
                    kr = r;
                    kt = 1. - r;
                }
                // reflected_color's share, used by shade_hit's tail below -- stored only where it is not 1: the tail
                // asks for it under the same condition (containers pass ran, material reflective and transparent)
                if (m_refl > 0. && m_transp > 0.) set_kr(kr);
                const double n_ratio = n1 / n2;
                const double cos_i = dot(h_eye, h_normal);
                const double sin2_t = (n_ratio * n_ratio) * (1. - cos_i * cos_i);
                CNT(ST_REFRACT, 1);
                if (!(sin2_t > 1.)) {
                    if (sp < RTC_MAX_DEPTH_STACK) {
                        const double cos_t = sqrt(1. - sin2_t);
                        Pending &q = pend_at(RTC_PEND(sp));
                        q.o = unpark3(pk, PK_POINT) - h_normal * RTC_EPS;  // under_point
                        q.d = h_normal * (n_ratio * cos_i - cos_t) - h_eye * n_ratio;
                        q.w = get_w() * (nl * (m_transp * kt));
                        q.remaining = remaining;  // Q2: not decremented (world.rs:142)
#if RTC_RING_PAD
                        {
                            Pending &z = pend_pad[RTC_PEND(sp)];
                            z = q;
                            asm volatile("" ::"v"(&z) : "memory");
                        }
#endif
                        if (STATS) pend_mult[RTC_PEND(sp)] = mult * nl;
                        sp++;
                    } else {
                        CNT(ST_OVERFLOW, 1);
                        flag_overflow();
                    }
                }
            }
            begin_lights = true;
        }
        SUB_END(ST_CONTAINER);
        // ---- the lights of this hit (world.rs:45-55).  A lane comes here with the result of a shadow pass
        // (T.mode == MODE_SHADOW) or to begin with the first light (begin_lights).  One round of the loop either
        // consumes a result -- Material::lighting, next sample / next light / finish -- or sets up the next shadow ray.
        // A ray whose light lies BEHIND the surface is not traced: material.rs:49-52 sets diffuse and specular to
        // black when light_dot_normal < 0 whatever light.intensity_at() returned, so the answer of is_shadowed cannot
        // reach the pixel; such a sample goes straight back into the loop as a result (bit-identical colours, and
        // for the mirror / glass spheres of the stress scene about half of all shadow passes).
        bool have_result = T.mode == MODE_SHADOW, need_setup = begin_lights, behind = false;
        V3 surface_now = mk(0., 0., 0.);  // one light: Material::lighting's result on its way to shade_hit's tail
        if (begin_lights) li = 0, su = sv = 0, unshadowed = 0;
        // (the set-up comes first in program order: a lane whose light is behind the surface falls through into the SAME
        // execution of the lighting block that serves the lanes arriving with a shadow pass's result)
#pragma clang loop unroll(disable)
        for (;;) {
            if (need_setup) {
                need_setup = false;
                const auto &lt = RTC_LOAD_LIGHT(li);
                const int lt_area = lt.is_area;
                V3 lp = ld3(lt.position);
                const V3 lt_pos = lp;
                if (lt_area)  // AreaLight::point_on_light, light.rs:106-108
                    lp = ld3(lt.corner) + ld3(lt.uvec) * ((double)su + 0.5) + ld3(lt.vvec) * ((double)sv + 0.5);
                const V3 h_normal = unpark3(pk, PK_NORMAL);
                const V3 over = unpark3(pk, PK_POINT) + h_normal * RTC_EPS;
                const V3 dv = lp - over;  // world.rs:79-83
                ray.o = over;
                ray.d = normalize(dv);
                T.lim = magnitude(dv);
                T.prim = -1;
                T.mode = MODE_SHADOW;
#ifndef RTC_NO_SHADOW_SKIP
                // Material::lighting's light vector (material.rs:39): towards light.position(), which for a point
                // light is the direction just formed
                const V3 lightv = lt_area ? normalize(lt_pos - over) : ray.d;
                if (dot(lightv, h_normal) < 0.) {
                    behind = true, have_result = true;
                    CNT(ST_SHADOW_SKIPPED, lt_area ? lt.samples : 1u);
                } else
#endif
                    CNT(ST_SHADOW, 1);
            }
            if (!have_result) break;  // a shadow pass starts (or the lane has nothing to do with lights this round)
            have_result = false;
            BLOCK_PROBE(true, ST_SECONDARY_REF, ST_OVERFLOW);
            {
                // World::is_shadowed (world.rs:78-91): only the NEAREST positive hit is consulted (Q5); when every
                // material casts shadows that equals "some hit has 0 < t < distance" (any-hit early exit).
                const bool shadowed = !behind && T.prim >= 0 &&
                                      (s.all_cast_shadow || RTC_MAT_CASTS(s.prims[T.prim].material));
                const auto &lt = RTC_LOAD_LIGHT(li);
                // everything this block and shade_hit's tail will read of the light and the material is requested here
                // in one go (one round trip; as separate look-ups behind each branch it was five)
                const DMaterial m = RTC_LOAD_MATERIAL(h_mat);
                const int lt_area = lt.is_area;
                const unsigned lt_usteps = lt.usteps, lt_vsteps = lt.vsteps, lt_samples = lt.samples;
                const V3 lt_pos = ld3(lt.position), lt_int = ld3(lt.intensity);
                __builtin_amdgcn_sched_barrier(0);
                double inten;
                if (behind) {
                    inten = 0.;  // not used by lighting(): the light is behind the surface
                } else if (lt_area) {  // light.rs:112-125: total += 1.0 per unshadowed sample, then total / samples
                    if (!shadowed) unshadowed++;
                    su++;
                    if (su >= lt_usteps) su = 0, sv++;
                    if (sv < lt_vsteps) {
                        need_setup = true;
                        continue;
                    }
                    inten = (double)unshadowed / (double)lt_samples;
                } else {  // light.rs:50-56
                    inten = shadowed ? 0. : 1.;
                }
                SUB_END(ST_PRIM0 + 0);  // (timing build: look-ups of the light block)
                const V3 over = ray.o;  // the shadow ray starts at over_point (world.rs:50, 79)
                V3 col = ld3(m.color);
                if (PATTERNS && m.pattern >= 0) col = pattern_at_shape(s, m.pattern, s.prims[h_prim], over);
                const V3 lightv = lt_area ? normalize(lt_pos - over) : ray.d;
                const V3 c = lighting(m, col, lightv, lt_int, unpark3(pk, PK_EYE), unpark3(pk, PK_NORMAL), inten);
                CNT(ST_LIGHT, 1);
                SUB_END(ST_PRIM0 + 1);  // (timing build: Material::lighting)
                if (lds_w)
                    surface_now = c;  // (the only light: shade_hit's tail follows in this visit)
                else
                    park3(pk, PK_SURFACE, (li == 0) ? c : unpark3(pk, PK_SURFACE) + c);  // world.rs:70 left fold
                li++;
                su = sv = 0;
                unshadowed = 0;
                behind = false;
                if (li < s.n_lights)
                    need_setup = true;
                else
                    finish = true, f_refl = m.reflectivity, f_transp = m.transparency;
            }
            if (!need_setup) break;
        }
        SUB_END(ST_SHADOW);
        if (finish) {
            // World::shade_hit (world.rs:42-76).  The reference evaluates reflected_color/refracted_color inside
            // the per-light loop (Q3), i.e. adds them n_lights times: that factor goes into the child's weight.
            const double w = get_w();
            if (CANVAS_ACC) {
                const V3 part = (lds_w ? surface_now : unpark3(pk, PK_SURFACE)) * w;
                double *px = pixel_ptr();
                if (CANVAS_RMW && !shared) {
                    px[0] = px[0] + part.x, px[1] = px[1] + part.y, px[2] = px[2] + part.z;
                } else {  // (no-return atomics: nothing to wait for)
                    __hip_atomic_fetch_add(px + 0, part.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(px + 1, part.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(px + 2, part.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                set_color(get_color() + (lds_w ? surface_now : unpark3(pk, PK_SURFACE)) * w);
            }
            next_ray = true;
            if (remaining > 0 && f_refl != 0.) {
                // reflected_color, world.rs:93-106 (the refracted ray, if any, already waits in the ring)
                const V3 h_normal = unpark3(pk, PK_NORMAL);
                const V3 h_eye = unpark3(pk, PK_EYE);
                ray.o = unpark3(pk, PK_POINT) + h_normal * RTC_EPS;  // over_point
                ray.d = reflect(-h_eye, h_normal);
                // (the Schlick factor exists where the containers block stored it: that block ran iff remaining > 0 and
                // the material is transparent, and stored iff it is reflective as well; everywhere else it is 1)
                const double kr = (f_transp != 0. && f_refl > 0. && f_transp > 0.) ? get_kr() : 1.;
                set_w(w * (nl * (f_refl * kr)));
                remaining -= 1;
                if (STATS) mult = mult * nl;
                next_ray = false;
                T.mode = MODE_CLOSEST;
                T.lim = inf, T.prim = -1;
                CNT(ST_SECONDARY, 1);
                CNT(ST_SECONDARY_REF, mult);
                rays++;
            }
        }
        SUB_END(ST_SECONDARY);
        if (next_ray) {
            if (sp == 0) {
                alive = false;
            } else {
                sp--;
                const Pending q = pend_at(RTC_PEND(sp));
                ray.o = q.o, ray.d = q.d, remaining = q.remaining;
                set_w(q.w);
                if (STATS) mult = pend_mult[RTC_PEND(sp)];
                T.mode = MODE_CLOSEST;
                T.lim = inf, T.prim = -1;
                CNT(ST_SECONDARY, 1);
                CNT(ST_SECONDARY_REF, mult);
                rays++;
            }
        }
        if (rays > (int)prm.max_rays) {
            CNT(ST_OVERFLOW, 1);
            flag_overflow();
            if (CANVAS_ACC)
                __hip_atomic_fetch_add(pixel_ptr(), __longlong_as_double(0x7ff8000000000000LL), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            else
                set_color(mk(__longlong_as_double(0x7ff8000000000000LL), 0., 0.));
            alive = false;
        }
        SUB_END(ST_BOX);
        if ((PERSISTENT || in_parts) && !alive) write_pixel();  // the lane is free for the next pixel / waiting ray
#if RTC_KILL_DEAD
        // A lane that went through the state machine is BETWEEN passes: the traversal's own state (cursor, stack pointer,
        // node in hand, the f32 form of the ray) and the fold's scratch values are dead for it -- the next pass sets them
        // before it reads them.  The compiler cannot see that (the reads sit behind `in_pass`), so it is told here: an
        // empty asm "defines" them anew, which ends their old live ranges at their last use above instead of carrying
        // them through the whole state machine.  (Per-lane liveness is what counts for VGPRs: the lanes that are inside a
        // pass are masked off here and keep their values.)
        asm volatile("" : "=v"(pos), "=v"(end), "=v"(sb), "=v"(cur));
        asm volatile("" : "=v"(slab_ray.ax), "=v"(slab_ray.ay), "=v"(slab_ray.az));
        asm volatile("" : "=v"(slab_ray.bx), "=v"(slab_ray.by), "=v"(slab_ray.bz));
#if RTC_SLAB_PERM
        asm volatile("" : "=v"(slab_ray.sx), "=v"(slab_ray.sy), "=v"(slab_ray.sz));
#endif
#if RTC_SEG_CACHE
        asm volatile("" : "=v"(slab_ray.hi), "=v"(slab_ray.lim_lo));
#endif
        asm volatile("" : "=v"(T.last), "=v"(T.a), "=v"(T.b), "=v"(T.c_best_prim), "=v"(T.c_hit_open));
#endif
        }  // if (alive)
        STAGE_END(ST_CYC_SHADE);
    }
    if (!PERSISTENT && !in_parts && rendered) write_pixel();  // the whole tile at once

#ifdef RTC_TIMING_ONLY
    if (STATS && lane == 0) {
        // wave lifetimes on the constant 100 MHz clock: sum of lifetimes, latest end, earliest start (as max of ~start)
        const unsigned long long end_rt = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&stats[ST_PRIM0 + 6], end_rt - wave_start_rt);
        atomicMax(&stats[ST_SHADED], end_rt);
        atomicMax(&stats[ST_LIGHT], ~wave_start_rt);
    }
#endif
    if (STATS) {
        for (int i = 0; i < ST_COUNT; i++) {
            unsigned long long v = cnt.c[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, RTC_WAVE);
            if (lane == 0 && v) atomicAdd(&stats[i], v);
        }
    }
}

// ---- rtc_render_device_static: the work buffer as a static device array ------------------------------------------
// One copy per GPU, loaded with the code object and held until the process ends.  `scratch` is a host-side stand-in
// for the symbol, not a device address: host code only ever uses the per-device pointer that hipGetSymbolAddress
// returned (kernel argument, hipMemsetAsync).  Contents survive from call to call; what must start at zero (tile
// counters, mailbox states) is cleared by launch_render's hipMemsetAsync before every persistent launch, and the
// overflow flag is compared with a per-launch id, so stale values never match.
static __device__ __align__(256) unsigned char scratch[RTC_WORK_TOTAL_BYTES];
static_assert(sizeof(Pending) == 64 && RTC_MAX_DEPTH_STACK == RTC_RING_SLOTS, "one cache line per waiting ray");
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
static unsigned resident_waves_per_cu(K kernel) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RTC_WAVE, 0) != hipSuccess || per_cu <= 0)
        per_cu = 4 * RTC_MIN_WAVES;
    return (unsigned)per_cu;
}

template <bool STATS, bool GROUPS, bool PATTERNS, bool QNODES>
static void launch_variant(const DScene &scene, DRenderParams &params, double *d_rgb, unsigned long long *ds,
                           unsigned *d_work, unsigned n_cu, hipStream_t st) {
    const unsigned tiles = params.tiles_x * params.tiles_y;
    static const unsigned order_env = getenv("RTC_TILE_ORDER") ? (unsigned)atoi(getenv("RTC_TILE_ORDER")) : 0u;
    params.tile_order = order_env;
    static const unsigned leaf_env = getenv("RTC_LEAF_DEN") ? (unsigned)atoi(getenv("RTC_LEAF_DEN")) : 0u;
    // large hierarchies (quantised nodes) are latency bound: there a primitive test waits until a third of the
    // traversing lanes want one (stress 4K 2.86 -> 2.61 s); on small scenes waiting buys nothing
    params.leaf_den = leaf_env ? leaf_env : (QNODES ? RTC_LEAF_DEN : 64u);
    // schedule: static = one workgroup per tile, the hardware dispatcher balances (best while the lanes of a
    // tile finish together); persistent = resident waves draw tiles from the counters and refill free lanes
    // (best when some pixels grow ray trees thousands of rays deep)
    if (params.persistent) {
        // occupancy is a property of the code object (queried once per variant); the CU count is this device's
        static const unsigned per_cu_env = getenv("RTC_GRID_PER_CU") ? (unsigned)atoi(getenv("RTC_GRID_PER_CU")) : 0u;  // (A/B)
        static const unsigned per_cu_occ = resident_waves_per_cu(k_render<STATS, GROUPS, PATTERNS, true, QNODES>);
        const unsigned per_cu = per_cu_env ? per_cu_env : per_cu_occ;
        // (the work buffer holds a mailbox and a ring of waiting rays per wave of the grid: RTC_MAX_WAVES of them)
        const unsigned cap = (RTC_RING_GLOBAL && per_cu * n_cu > RTC_MAX_WAVES) ? RTC_MAX_WAVES : per_cu * n_cu;
        const unsigned waves = tiles < cap ? tiles : cap;
        static const unsigned refill_env = getenv("RTC_REFILL_MIN") ? (unsigned)atoi(getenv("RTC_REFILL_MIN")) : 0u;
        params.refill_min = refill_env ? refill_env : RTC_REFILL_MIN;
        params.dyn_base = waves;  // wave b renders tile b first
        static const int share_env = getenv("RTC_SHARE") ? atoi(getenv("RTC_SHARE")) : 1;
        if (params.refuse_oversize && params.share != 0u && waves > RTC_MAX_WAVES) {
            params.refuse_oversize = 2u;  // reported by launch_render; nothing is launched
            return;
        }
        // (share > 1: the polls a wave without work waits for rays of others before it leaves, rtc_render_opts.tail_polls;
        // 1 = until the frame ends, i.e. the guard value 2^20)
        params.share = (params.share != 0u && share_env != 0 && waves <= RTC_MAX_WAVES) ? (params.share > 1u ? params.share : (1u << 20)) : 0u;
        params.n_waves = waves;
        static const unsigned stay_env = getenv("RTC_STAY_MIN") ? (unsigned)atoi(getenv("RTC_STAY_MIN")) : 0u;
        params.stay_min = stay_env ? stay_env : RTC_STAY_MIN;
        hipLaunchKernelGGL((k_render<STATS, GROUPS, PATTERNS, true, QNODES>), dim3(waves), dim3(RTC_WAVE), 0, st, scene,
                           params, d_rgb, ds, d_work);
    } else {
        params.refill_min = RTC_WAVE;
        params.dyn_base = tiles;
        params.share = 0u, params.n_waves = tiles;
        // (one workgroup per tile; the first n_heavy tiles of the list take one per part)
        hipLaunchKernelGGL((k_render<STATS, GROUPS, PATTERNS, false, QNODES>),
                           dim3(tiles + params.n_heavy * ((1u << params.part_shift) - 1u)),
                           dim3(RTC_WAVE), 0, st, scene, params, d_rgb, ds, d_work);
    }
}

template <bool STATS, bool GROUPS, bool PATTERNS>
static void launch_nodes(const DScene &scene, DRenderParams &params, double *d_rgb, unsigned long long *ds,
                         unsigned *d_work, unsigned n_cu, hipStream_t st) {
    if (scene.quantized_nodes)
        launch_variant<STATS, GROUPS, PATTERNS, true>(scene, params, d_rgb, ds, d_work, n_cu, st);
    else
        launch_variant<STATS, GROUPS, PATTERNS, false>(scene, params, d_rgb, ds, d_work, n_cu, st);
}

int launch_render(const DScene &scene, const DRenderParams &params_in, double *d_rgb, uint64_t *d_stats,
                  unsigned *d_work, unsigned n_cu, bool collect_stats, void *stream, uint32_t *launch_id_out,
                  bool refuse_oversize) {
    DRenderParams params = params_in;
    const unsigned tiles = params.tiles_x * params.tiles_y;
    // ids are unique per process (scenes on one GPU may share a work buffer, rtc_render_device_static) and never 0
    const uint32_t id = next_launch_id();
    params.launch_id = id;
    if (launch_id_out) *launch_id_out = id;
    if (params.max_rays == 0u || params.max_rays > 0x7fffffffu) params.max_rays = RTC_MAX_RAYS_PER_PIXEL;
    if (tiles == 0) return RTC_OK;
    hipStream_t st = (hipStream_t)stream;
    // the tile counters of the persistent grid start at zero for every launch
    if (params.persistent && hipMemsetAsync(d_work, 0, RTC_WORK_RESET_BYTES, st) != hipSuccess) {
        set_error("hipMemsetAsync(work counter) failed");
        return RTC_ERR_HIP;
    }
    params.refuse_oversize = refuse_oversize ? 1u : 0u;
    unsigned long long *ds = (unsigned long long *)d_stats;
    if (collect_stats)
        launch_nodes<true, true, true>(scene, params, d_rgb, ds, d_work, n_cu, st);
    else if (scene.any_pattern || scene.any_csg)  // the variant with the rarely used features
        launch_nodes<false, true, true>(scene, params, d_rgb, ds, d_work, n_cu, st);
    else if (scene.any_group)
        launch_nodes<false, true, false>(scene, params, d_rgb, ds, d_work, n_cu, st);
    else
        launch_nodes<false, false, false>(scene, params, d_rgb, ds, d_work, n_cu, st);
    if (params.refuse_oversize == 2u) {
        set_error("persistent grid of more than %u resident waves: the work buffer holds that many mailboxes; set "
                  "exchange = 1 or schedule = 1", RTC_MAX_WAVES);
        return RTC_ERR_INVALID;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("kernel launch failed: %s", hipGetErrorString(e));
        return RTC_ERR_HIP;
    }
    return RTC_OK;
}

#include "rtc_wavefront.inc"

}  // namespace rtc
