// rtc_scene.cpp -- flattening of the Shape tree into SoA device tables, host BVH, upload.
//
// The reference keeps shapes as Arc<dyn Shape> trees with a global parent registry
// (ray_tracer/src/shape.rs:22-24, 82-103, 1930-1942).  On the device there is no tree: every leaf shape
// becomes one DPrim in depth-first order (the order World::intersect / Group::local_intersect visit them,
// world.rs:176-186, shape.rs:1930-1942) and carries the root-first list of inverse transforms the reference
// would apply one after another (shape.rs:2035 then e.g. :256; shape.rs:82-103 for points and normals).
#include "rtc_scene.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <limits>

namespace rtc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

Scene::~Scene() { release_device(); }

uint32_t next_launch_id() {
    static std::atomic<uint32_t> next{1u};
    uint32_t id = next.fetch_add(1u, std::memory_order_relaxed);
    if (id == 0u) id = next.fetch_add(1u, std::memory_order_relaxed);
    return id;
}

DeviceGuard::DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) ok = false;
}
DeviceGuard::~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
}

void Scene::release_device() {
    for (DeviceReplica &r : reps) {
        if (r.device < 0) continue;
        DeviceGuard g(r.device);
        for (void *p : r.allocs) (void)hipFree(p);
        if (r.d_canvas) (void)hipFree(r.d_canvas);
        for (DeviceReplica::SharePerm &sp : r.share_perms) {
            if (sp.d_perm) (void)hipFree(sp.d_perm);
        }
        r.share_perms.clear();
        for (int k = 0; k < 2; k++) {
            if (r.wf.rays[k]) (void)hipFree(r.wf.rays[k]);
            if (r.wf.ctx[k]) (void)hipFree(r.wf.ctx[k]);
        }
        if (r.wf.hits) (void)hipFree(r.wf.hits);
        if (r.wf.counts) (void)hipFree(r.wf.counts);
        if (r.wf.h_counts) (void)hipHostFree(r.wf.h_counts);
        if (r.ev_start) (void)hipEventDestroy((hipEvent_t)r.ev_start);
        if (r.ev_stop) (void)hipEventDestroy((hipEvent_t)r.ev_stop);
        if (r.stream) (void)hipStreamDestroy((hipStream_t)r.stream);
    }
    for (void *&h : h_stage) {
        if (h) (void)hipHostFree(h);
        h = nullptr;
    }
    stage_bytes = 0;
    reps.clear();
    device_bytes = 0;
}

int Scene::add_item(Item &&it) {
    it.parent = open_groups.empty() ? -1 : open_groups.back();
    const int idx = (int)items.size();
    items.push_back(std::move(it));
    if (items[idx].parent >= 0)
        items[items[idx].parent].children.push_back(idx);
    else
        top_level.push_back(idx);
    return idx;
}

// ---------------------------------------------------------------------------------------------
// flatten
// ---------------------------------------------------------------------------------------------

static void store_inv(const Mat4 &inv, double out[12]) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) out[r * 4 + c] = inv.m[r][c];
}

static const double INF = std::numeric_limits<double>::infinity();

// Object-space bounds of an analytic shape; false = unbounded (goes to the always-tested list).
static bool local_bounds(const Item &it, Bounds &b) {
    switch (it.type) {
    case RTC_SPHERE:
    case RTC_CUBE:
        for (int a = 0; a < 3; a++) b.lo[a] = -1.0, b.hi[a] = 1.0;
        return true;
    case RTC_PLANE: return false;
    case RTC_CYLINDER:
        // side hits: y in (min, max), x^2+z^2 = 1; caps at y = min / max with x^2+z^2 <= 1 (shape.rs:895-959)
        if (!std::isfinite(it.minimum) || !std::isfinite(it.maximum)) return false;
        b.lo[0] = b.lo[2] = -1.0, b.hi[0] = b.hi[2] = 1.0;
        b.lo[1] = std::min(it.minimum, it.maximum), b.hi[1] = std::max(it.minimum, it.maximum);
        return true;
    case RTC_CONE: {
        // An OPEN cone returns an unclipped hit when |a| < EPSILON (shape.rs:1219-1225, Q8), so it can be hit
        // anywhere along the infinite double cone: never bounded.  A CLOSED cone yields side hits with
        // y in (min, max) (radius |y|) and cap hits on the planes y = |min|, y = |max| with
        // x^2 + z^2 <= |limit| (NOT squared, shape.rs:1269-1277 + intersection.rs:34-39).
        if (!it.closed) return false;
        if (!std::isfinite(it.minimum) || !std::isfinite(it.maximum)) return false;
        const double amin = std::fabs(it.minimum), amax = std::fabs(it.maximum);
        const double r = std::max(std::max(amin, amax), std::max(std::sqrt(amin), std::sqrt(amax)));
        b.lo[0] = b.lo[2] = -r, b.hi[0] = b.hi[2] = r;
        b.lo[1] = std::min(std::min(it.minimum, it.maximum), std::min(amin, amax));
        b.hi[1] = std::max(std::max(it.minimum, it.maximum), std::max(amin, amax));
        return true;
    }
    default: return false;
    }
}

static void bounds_empty(Bounds &b) {
    for (int a = 0; a < 3; a++) b.lo[a] = INF, b.hi[a] = -INF;
}
static void bounds_add(Bounds &b, Vec3 p) {
    const double v[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; a++) {
        b.lo[a] = std::min(b.lo[a], v[a]);
        b.hi[a] = std::max(b.hi[a], v[a]);
    }
}
// Conservative padding: the BVH must never cull a hit the reference's own (rounded) arithmetic would
// report.  1e-6 relative to the largest coordinate / extent dwarfs every f64 rounding error on the path
// (1e-16 relative) and costs nothing measurable in traversal.
static void bounds_pad(Bounds &b) {
    double scale = 0.0;
    for (int a = 0; a < 3; a++)
        scale = std::max(scale, std::max(std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])), b.hi[a] - b.lo[a]));
    const double pad = 1e-6 * scale + 1e-9;
    for (int a = 0; a < 3; a++) b.lo[a] -= pad, b.hi[a] += pad;
}

int Scene::flatten() {
    xforms.clear(), xform_fwd.clear(), chains.clear(), prims.clear(), tris.clear();
    prim_bounds.clear(), prim_bounded.clear(), unbounded.clear(), bvh.clear(), bvh_prims.clear();
    prim_anc_id.clear(), prim_csg.clear(), csg_roots.clear(), csg_nodes.clear();
    dmaterials.clear(), dpatterns.clear();

    if (!has_camera) {
        set_error("scene has no camera (parse_scene ParseError::MissingCamera)");
        return RTC_ERR_INVALID;
    }
    // camera.rs:17-37
    {
        Mat4 inv;
        if (!mat_inverse(view, inv)) {
            set_error("camera transformation is not invertible");
            return RTC_ERR_NON_INVERTIBLE;
        }
        store_inv(inv, cam.inv);
        const double half_view = std::tan(fov / 2.);
        const double aspect = (double)hsize / (double)vsize;
        if (aspect >= 1.) {
            cam.half_width = half_view;
            cam.half_height = half_view / aspect;
        } else {
            cam.half_width = half_view * aspect;
            cam.half_height = half_view;
        }
        cam.pixel_size = cam.half_width * 2. / (double)hsize;
        cam.hsize = hsize;
        cam.vsize = vsize;
    }

    for (const rtc_material &m : materials) {
        DMaterial d;
        for (int i = 0; i < 3; i++) d.color[i] = m.color[i];
        d.ambient = m.ambient, d.diffuse = m.diffuse, d.specular = m.specular, d.shininess = m.shininess;
        d.reflectivity = m.reflectivity, d.transparency = m.transparency, d.refractive_index = m.refractive_index;
        d.casts_shadow = m.casts_shadow ? 1 : 0;
        d.pattern = m.pattern;
        d.spec_skip_rde = (std::isfinite(m.shininess) && m.shininess >= 1.0) ? std::exp2(-66.0 / m.shininess) : 0.0;
        if (m.pattern >= (int32_t)patterns.size()) {
            set_error("material references pattern %d, only %zu defined", m.pattern, patterns.size());
            return RTC_ERR_INVALID;
        }
        dmaterials.push_back(d);
    }
    for (const HostPattern &p : patterns) {
        DPattern d{};
        d.kind = p.kind, d.sub_a = p.sub_a, d.sub_b = p.sub_b;
        for (int i = 0; i < 3; i++) d.a[i] = p.a[i], d.b[i] = p.b[i];
        Mat4 inv;
        if (!mat_inverse(p.transform, inv)) {
            set_error("pattern transformation is not invertible");
            return RTC_ERR_NON_INVERTIBLE;
        }
        store_inv(inv, d.inv);
        dpatterns.push_back(d);
    }

    // Depth-first walk in insertion order.  chain_of[item] = (offset, len) of the root-first xform chain that
    // applies to the CHILDREN of a group item.
    struct Frame {
        int item;
        int32_t chain_off, chain_len;  // chain of the parent (applies to this item before its own transform)
        int32_t group_material;        // material of the enclosing group or -1
        int32_t chain_id;
        int32_t csg_root;              // Csg tree this item belongs to (index into csg_roots) or -1
        int32_t marker;                // 0: visit the item; 1: the left operand of Csg `item` is complete; 2: both are
    };
    std::vector<Frame> stack;
    for (auto it = top_level.rbegin(); it != top_level.rend(); ++it) stack.push_back({*it, 0, 0, -1, -1, -1, 0});
    std::vector<int32_t> csg_begin(items.size(), 0), csg_mid(items.size(), 0);
    int32_t next_chain_id = 0;
    const int32_t root_chain_id = next_chain_id++;

    auto push_xform = [&](const Mat4 &t, int32_t &idx) -> bool {
        Mat4 inv;
        if (!mat_inverse(t, inv)) return false;
        DXform x;
        store_inv(inv, x.inv);
        idx = (int32_t)xforms.size();
        xforms.push_back(x);
        xform_fwd.push_back(t);
        return true;
    };
    auto world_bounds = [&](const Bounds &local, int32_t chain_off, int32_t chain_len, Bounds &out) {
        bounds_empty(out);
        for (int corner = 0; corner < 8; corner++) {
            Vec3 p = {(corner & 1) ? local.hi[0] : local.lo[0], (corner & 2) ? local.hi[1] : local.lo[1],
                      (corner & 4) ? local.hi[2] : local.lo[2]};
            for (int k = chain_len - 1; k >= 0; k--) p = mat_point(xform_fwd[chains[chain_off + k]], p);
            bounds_add(out, p);
        }
        bounds_pad(out);
    };

    while (!stack.empty()) {
        const Frame f = stack.back();
        stack.pop_back();
        const Item &it = items[f.item];
        if (f.marker == 1) {
            csg_mid[f.item] = (int32_t)prims.size();
            continue;
        }
        if (f.marker == 2) {
            // Csg::local_intersect's operands (shape.rs:2227-2235) as ranges of the tree's leaves; post-order
            DCsgRoot &root = csg_roots[f.csg_root];
            DCsgNode n{};
            n.op = it.type;
            n.l_first = csg_begin[f.item] - root.prim_first, n.l_count = csg_mid[f.item] - csg_begin[f.item];
            n.r_first = csg_mid[f.item] - root.prim_first, n.r_count = (int32_t)prims.size() - csg_mid[f.item];
            csg_nodes.push_back(n);
            root.n_nodes = (int32_t)csg_nodes.size() - root.node_first;
            root.n_prims = (int32_t)prims.size() - root.prim_first;
            continue;
        }
        if (it.kind == Item::CSG) {
            if (it.children.size() != 2) {
                set_error("a Csg takes exactly a left and a right shape (Csg::new, shape.rs:2166), %zu given", it.children.size());
                return RTC_ERR_INVALID;
            }
            // Csg::transformation() is the identity (shape.rs:2250-2252): for points and normals the node is one more
            // level of the parent chain -- inverse(identity) and, for normals, one more normalisation (shape.rs:82-103)
            int32_t xi;
            (void)push_xform(Mat4::identity(), xi);
            const int32_t off = (int32_t)chains.size();
            for (int k = 0; k < f.chain_len; k++) chains.push_back(chains[f.chain_off + k]);
            chains.push_back(xi);
            const int32_t cid = next_chain_id++;
            int32_t root = f.csg_root;
            if (root < 0) {
                root = (int32_t)csg_roots.size();
                csg_roots.push_back(DCsgRoot{(int32_t)prims.size(), 0, (int32_t)csg_nodes.size(), 0});
            }
            csg_begin[f.item] = (int32_t)prims.size();
            stack.push_back({f.item, 0, 0, -1, -1, root, 2});
            stack.push_back({it.children[1], off, f.chain_len + 1, -1, cid, root, 0});
            stack.push_back({f.item, 0, 0, -1, -1, root, 1});
            stack.push_back({it.children[0], off, f.chain_len + 1, -1, cid, root, 0});
            continue;
        }
        if (it.material >= (int32_t)materials.size()) {
            set_error("shape references material %d, only %zu defined", it.material, materials.size());
            return RTC_ERR_INVALID;
        }
        if (it.kind == Item::GROUP) {
            int32_t xi;
            if (!push_xform(it.transform, xi)) {
                set_error("group transformation is not invertible");
                return RTC_ERR_NON_INVERTIBLE;
            }
            const int32_t off = (int32_t)chains.size();
            for (int k = 0; k < f.chain_len; k++) chains.push_back(chains[f.chain_off + k]);
            chains.push_back(xi);
            const int32_t cid = next_chain_id++;
            for (auto c = it.children.rbegin(); c != it.children.rend(); ++c)
                stack.push_back({*c, off, f.chain_len + 1, it.material, cid, f.csg_root, 0});
        } else if (it.kind == Item::SHAPE) {
            if (it.material < 0) {
                set_error("shape without a material");
                return RTC_ERR_INVALID;
            }
            int32_t xi;
            if (!push_xform(it.transform, xi)) {
                set_error("shape transformation is not invertible");
                return RTC_ERR_NON_INVERTIBLE;
            }
            const int32_t off = (int32_t)chains.size();
            for (int k = 0; k < f.chain_len; k++) chains.push_back(chains[f.chain_off + k]);
            chains.push_back(xi);
            DPrim p{};
            p.type = it.type;  // rtc_shape_type values coincide with PrimType for the analytic shapes
            p.material = it.material;
            p.chain_off = off;
            p.chain_len = f.chain_len + 1;
            p.minimum = (it.type == RTC_CYLINDER || it.type == RTC_CONE) ? it.minimum : -INF;
            p.maximum = (it.type == RTC_CYLINDER || it.type == RTC_CONE) ? it.maximum : INF;
            p.closed = (it.type == RTC_CYLINDER || it.type == RTC_CONE) ? it.closed : 0;
            p.tri = -1;
            p.own_xform = xi;
            p.chain_id = next_chain_id++;
            if (it.type == RTC_PLANE) {
                // normal_to_world up the chain, one normalisation per level (shape.rs:94-103); transpose(inverse) * n
                // with the row-times-vector order of matrix.rs:348-365
                Vec3 n = {0., 1., 0.};
                for (int k = p.chain_len - 1; k >= 0; k--) {
                    const double *m = xforms[chains[p.chain_off + k]].inv;
                    const Vec3 r = {m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
                                    m[2] * n.x + m[6] * n.y + m[10] * n.z};
                    n = vnorm(r);
                }
                p.plane_normal[0] = n.x, p.plane_normal[1] = n.y, p.plane_normal[2] = n.z;
            }
            Bounds lb, wb;
            bounds_empty(wb);
            const bool bounded = local_bounds(it, lb);
            if (bounded) world_bounds(lb, p.chain_off, p.chain_len, wb);
            prims.push_back(p);
            prim_anc_id.push_back(f.chain_len > 0 ? f.chain_id : -1);
            prim_csg.push_back(f.csg_root);
            prim_bounds.push_back(wb);
            prim_bounded.push_back(bounded ? 1 : 0);
        } else {
            const size_t n = it.points.size() / 9;
            const bool smooth = !it.normals.empty();
            // shape.rs:1520-1532 / 1756-1768: the parent group's material wins when it is Some(..)
            const int32_t mat = (f.group_material >= 0) ? f.group_material : it.material;
            if (mat < 0) {
                set_error("triangles without a material");
                return RTC_ERR_INVALID;
            }
            for (size_t t = 0; t < n; t++) {
                const double *q = &it.points[t * 9];
                DTri d{};
                const Vec3 p1 = {q[0], q[1], q[2]}, p2 = {q[3], q[4], q[5]}, p3 = {q[6], q[7], q[8]};
                const Vec3 e1 = p2 - p1, e2 = p3 - p1;  // shape.rs:1463-1464
                d.p1[0] = p1.x, d.p1[1] = p1.y, d.p1[2] = p1.z;
                d.e1[0] = e1.x, d.e1[1] = e1.y, d.e1[2] = e1.z;
                d.e2[0] = e2.x, d.e2[1] = e2.y, d.e2[2] = e2.z;
                if (smooth) {
                    const double *nn = &it.normals[t * 9];
                    for (int i = 0; i < 3; i++) d.n1[i] = nn[i], d.n2[i] = nn[3 + i], d.n3[i] = nn[6 + i];
                } else {
                    const Vec3 nrm = vnorm(vcross(e2, e1));  // shape.rs:1465
                    d.n1[0] = nrm.x, d.n1[1] = nrm.y, d.n1[2] = nrm.z;
                }
                DPrim p{};
                p.type = smooth ? PT_SMOOTH_TRIANGLE : PT_TRIANGLE;
                p.material = mat;
                p.chain_off = f.chain_off;
                p.chain_len = f.chain_len;
                p.minimum = -INF, p.maximum = INF, p.closed = 0;
                p.tri = (int32_t)tris.size();
                p.own_xform = -1;
                p.chain_id = (f.chain_id >= 0) ? f.chain_id : root_chain_id;
                tris.push_back(d);
                Bounds wb;
                bounds_empty(wb);
                const Vec3 vs[3] = {p1, p2, p3};
                for (Vec3 v : vs) {
                    for (int k = f.chain_len - 1; k >= 0; k--) v = mat_point(xform_fwd[chains[f.chain_off + k]], v);
                    bounds_add(wb, v);
                }
                bounds_pad(wb);
                bool finite = true;
                for (int a = 0; a < 3; a++) finite = finite && std::isfinite(wb.lo[a]) && std::isfinite(wb.hi[a]);
                prims.push_back(p);
                prim_anc_id.push_back(f.chain_len > 0 ? f.chain_id : -1);
                prim_csg.push_back(f.csg_root);
                prim_bounds.push_back(wb);
                prim_bounded.push_back(finite ? 1 : 0);
            }
        }
    }
    n_transparent_prims = 0;
    for (const DPrim &p : prims)
        if (dmaterials[p.material].transparency != 0.) n_transparent_prims++;
    for (size_t i = 0; i < prims.size(); i++) {
        bool ok = prim_bounded[i];
        if (ok)
            for (int a = 0; a < 3; a++) ok = ok && std::isfinite(prim_bounds[i].lo[a]) && std::isfinite(prim_bounds[i].hi[a]);
        prim_bounded[i] = ok;
    }
    // one pseudo-primitive per Csg tree, after the real primitives (whose indices -- identity, tie order -- stay what
    // the depth-first walk gave them): the union of the leaves' boxes bounds whatever the operations keep
    const size_t n_real = prims.size();
    for (size_t r = 0; r < csg_roots.size(); r++) {
        const DCsgRoot &root = csg_roots[r];
        DPrim p{};
        p.type = PT_CSG;
        p.material = -1;  // (the tree has no material of its own: its leaves weigh the tiles in build_tile_order)
        p.tri = (int32_t)r;
        p.own_xform = -1;
        p.minimum = -INF, p.maximum = INF;
        Bounds wb;
        bounds_empty(wb);
        bool ok = root.n_prims > 0;
        for (int32_t k = 0; k < root.n_prims; k++) {
            const size_t q = (size_t)root.prim_first + k;
            ok = ok && prim_bounded[q];
            for (int a = 0; a < 3; a++)
                wb.lo[a] = std::min(wb.lo[a], prim_bounds[q].lo[a]), wb.hi[a] = std::max(wb.hi[a], prim_bounds[q].hi[a]);
        }
        prims.push_back(p);
        prim_anc_id.push_back(-1);
        prim_csg.push_back(-1);
        prim_bounds.push_back(wb);
        prim_bounded.push_back(ok ? 1 : 0);
    }
    for (size_t i = 0; i < prims.size(); i++)
        if (!prim_bounded[i] && (i >= n_real || prim_csg[i] < 0)) unbounded.push_back((int32_t)i);
    return RTC_OK;
}

// ---------------------------------------------------------------------------------------------
// BVH: binned SAH, binary, children adjacent, breadth-first order
// ---------------------------------------------------------------------------------------------

namespace {

struct BuildNode {
    Bounds b;
    int left = -1, right = -1;  // temp indices
    int first = 0, count = 0;
};

struct Builder {
    const std::vector<Bounds> &pb;
    std::vector<int32_t> &order;
    std::vector<BuildNode> nodes;
    bool median_only;
    uint32_t depth_limit;  // the device traversal stack bounds the depth of the tree
    uint32_t max_depth = 0;
#ifndef RTC_BVH_MAX_LEAF
#define RTC_BVH_MAX_LEAF 4
#endif
#ifndef RTC_BVH_BINS
#define RTC_BVH_BINS 16
#endif
#ifndef RTC_BVH_TRAV_COST
#define RTC_BVH_TRAV_COST 0.5
#endif
    static constexpr int MAX_LEAF = RTC_BVH_MAX_LEAF;
    static constexpr int BINS = RTC_BVH_BINS;

    static double area(const Bounds &b) {
        const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    static void merge(Bounds &a, const Bounds &b) {
        for (int k = 0; k < 3; k++) a.lo[k] = std::min(a.lo[k], b.lo[k]), a.hi[k] = std::max(a.hi[k], b.hi[k]);
    }
    static double centroid(const Bounds &b, int axis) { return 0.5 * (b.lo[axis] + b.hi[axis]); }

    int build(int first, int count, uint32_t depth) {
        const int idx = (int)nodes.size();
        nodes.emplace_back();
        {
            BuildNode &n = nodes[idx];
            n.first = first, n.count = count;
            bounds_empty(n.b);
            for (int i = 0; i < count; i++) merge(n.b, pb[order[first + i]]);
        }
        max_depth = std::max(max_depth, depth);
        if (count <= 1) return idx;

        Bounds cb;
        bounds_empty(cb);
        for (int i = 0; i < count; i++) {
            const Bounds &p = pb[order[first + i]];
            for (int k = 0; k < 3; k++) {
                const double c = centroid(p, k);
                cb.lo[k] = std::min(cb.lo[k], c), cb.hi[k] = std::max(cb.hi[k], c);
            }
        }
        int axis = 0;
        for (int k = 1; k < 3; k++)
            if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        const double extent = cb.hi[axis] - cb.lo[axis];

        // A SAH split may be arbitrarily lopsided (primitives in geometric progression, nested primitives): once the
        // depth left would no longer cover a balanced subtree of `count` primitives, the rest of this subtree is split
        // at the object median, whose depth is ceil(log2(count)) -- so the recursion (host stack) and the tree
        // (device stack) both stay within depth_limit whatever the input.
        uint32_t balanced = 0;
        while ((1u << balanced) < (uint32_t)count) balanced++;
        const bool force_median = median_only || depth + balanced + 1u >= depth_limit;
        int mid = -1;
        if (!force_median && extent > 0.0) {
            // binned SAH over the three axes
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_split = -1;
            for (int ax = 0; ax < 3; ax++) {
                const double ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.0)) continue;
                Bounds bb[BINS];
                int bc[BINS];
                for (int b = 0; b < BINS; b++) bounds_empty(bb[b]), bc[b] = 0;
                const double scale = BINS / ext;
                for (int i = 0; i < count; i++) {
                    const Bounds &p = pb[order[first + i]];
                    int b = (int)((centroid(p, ax) - cb.lo[ax]) * scale);
                    b = std::max(0, std::min(BINS - 1, b));
                    merge(bb[b], p);
                    bc[b]++;
                }
                double right_area[BINS];
                int right_cnt[BINS];
                Bounds acc;
                bounds_empty(acc);
                int cnt = 0;
                for (int b = BINS - 1; b >= 1; b--) {
                    merge(acc, bb[b]);
                    cnt += bc[b];
                    right_area[b] = area(acc), right_cnt[b] = cnt;
                }
                bounds_empty(acc);
                cnt = 0;
                for (int b = 0; b < BINS - 1; b++) {
                    merge(acc, bb[b]);
                    cnt += bc[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    const double cost = area(acc) * cnt + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = ax, best_split = b;
                }
            }
            const double leaf_cost = area(nodes[idx].b) * count;
            // traversal step ~ 1/2 primitive test (f64 slab test vs transform + quadratic / Moeller-Trumbore)
            if (best_axis >= 0 && (count > MAX_LEAF || best_cost + RTC_BVH_TRAV_COST * area(nodes[idx].b) < leaf_cost)) {
                const double ext = cb.hi[best_axis] - cb.lo[best_axis];
                const double scale = BINS / ext;
                const double lo = cb.lo[best_axis];
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int32_t p) {
                    int b = (int)((centroid(pb[p], best_axis) - lo) * scale);
                    b = std::max(0, std::min(BINS - 1, b));
                    return b <= best_split;
                });
                mid = (int)(it - order.begin());
                if (mid == first || mid == first + count) mid = -1;
            } else if (count <= MAX_LEAF) {
                return idx;  // leaf is cheaper
            }
        }
        if (mid < 0) {
            if (count <= MAX_LEAF) return idx;
            // object-median split along the widest centroid axis (also the depth-bounded fallback)
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](int32_t a, int32_t b) {
                                 const double ca = centroid(pb[a], axis), cbv = centroid(pb[b], axis);
                                 return ca < cbv || (ca == cbv && a < b);
                             });
        }
        const int l = build(first, mid - first, depth + 1);
        const int r = build(mid, first + count - mid, depth + 1);
        nodes[idx].left = l, nodes[idx].right = r;
        return idx;
    }
};

}  // namespace

// f64 -> f32 with the rounding directed away from the box interior
static float f32_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
static float f32_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

void Scene::build_bvh() {
    bvh.clear(), bvh4.clear(), bvh_q.clear(), bvh_prims.clear(), bvh_flags.clear(), compact_prims.clear();
    quantized_nodes = false;
    bvh_max_depth = 0, bvh_max_leaf = 0;
    root_ref = 0;
    for (int k = 0; k < 3; k++) root_min[k] = root_max[k] = 0.0;
    std::vector<int32_t> order;
    for (size_t i = 0; i < prims.size(); i++)
        if (prim_bounded[i] && prim_csg[i] < 0) order.push_back((int32_t)i);  // (leaves of a Csg: through its tree only)
    if (order.empty()) return;

    constexpr uint32_t DEPTH_LIMIT = 30;  // device traversal stack holds 32 entries (the kernel flags an overflow)
    Builder b{prim_bounds, order, {}, false, DEPTH_LIMIT};
    b.nodes.reserve(order.size() * 2);
    b.build(0, (int)order.size(), 0);
    bvh_max_depth = b.max_depth;
    // Leaf primitives in ascending flattened index (deterministic visiting order inside a leaf).
    for (const BuildNode &n : b.nodes)
        if (n.left < 0) std::sort(order.begin() + n.first, order.begin() + n.first + n.count);

    // "paired" subtrees hold only primitives whose records always come in pairs on both sides of their interior
    // (spheres: two records or none, shape.rs:218-239).  Such a primitive can only be an open container at the
    // hit if its records straddle the hit, so the containers pass may skip paired subtrees whose box ends
    // before the hit point (rtc_kernels.hip, MODE_CONTAINER).
    std::vector<uint8_t> paired(b.nodes.size(), 0);
    for (int k = (int)b.nodes.size() - 1; k >= 0; k--) {  // children are created after their parent
        const BuildNode &n = b.nodes[k];
        if (n.left < 0) {
            bool all = true;
            for (int q = 0; q < n.count; q++) all = all && prims[order[n.first + q]].type == PT_SPHERE;
            paired[k] = all;
        } else {
            paired[k] = paired[n.left] && paired[n.right];
        }
    }
    // Breadth-first relayout of the INTERNAL nodes only; every node stores both children's boxes.
    // Leaves become child references ~((first << 3) | (count - 1)), `first` relative to the leaf-record list (unbounded primitives first).
    const int32_t base = (int32_t)unbounded.size();
    // a sphere outside any group whose inverse transform is diagonal + translation fits a 64-byte DLeafS
    auto compact_sphere = [&](int32_t pi) {
        const DPrim &p = prims[pi];
        if (p.type != PT_SPHERE || p.chain_len != 1) return false;
        const double *a = xforms[chains[p.chain_off]].inv;
        return a[1] == 0.0 && a[2] == 0.0 && a[4] == 0.0 && a[6] == 0.0 && a[8] == 0.0 && a[9] == 0.0;
    };
    auto leaf_ref = [&](const BuildNode &n) -> int32_t {
        bvh_max_leaf = std::max<uint32_t>(bvh_max_leaf, (uint32_t)n.count);
        bool compact = quantized_nodes;  // large hierarchies only: small ones stay in L2 whatever the record size
        for (int q = 0; q < n.count && compact; q++) compact = compact_sphere(order[n.first + q]);
        if (compact) {
            const int32_t first = (int32_t)compact_prims.size();
            for (int q = 0; q < n.count; q++) compact_prims.push_back(order[n.first + q]);
            return ~(((RTC_LEAF_COMPACT | first) << 3) | (n.count - 1));
        }
        return ~(((base + n.first) << 3) | (n.count - 1));
    };
    for (int k = 0; k < 3; k++) root_min[k] = b.nodes[0].b.lo[k], root_max[k] = b.nodes[0].b.hi[k];
    if (b.nodes[0].left < 0) {
        root_ref = leaf_ref(b.nodes[0]);
    } else {
        root_ref = 0;
        std::vector<int> queue{0};
        std::vector<int> new_index(b.nodes.size(), -1);
        new_index[0] = 0;
        int next = 1;
        // first pass: assign indices in BFS order
        for (size_t head = 0; head < queue.size(); head++) {
            const BuildNode &n = b.nodes[queue[head]];
            for (int c : {n.left, n.right})
                if (b.nodes[c].left >= 0) {
                    new_index[c] = next++;
                    queue.push_back(c);
                }
        }
        bvh.resize(next);
        bvh_flags.assign(next, 0);
        // quantisation frame: the root box cut into 65535 cells per axis
        for (int k = 0; k < 3; k++) {
            q_org[k] = root_min[k];
            double cell = (root_max[k] - root_min[k]) / 65535.0;
            if (!(cell > 0.0)) cell = 1e-300;
            while (q_org[k] + 65535.0 * cell < root_max[k]) cell = std::nextafter(cell, INF);
            q_cell[k] = cell;
        }
        auto q_down = [&](double v, int k) -> uint16_t {
            double q = std::floor((v - q_org[k]) / q_cell[k]);
            q = std::max(0.0, std::min(65535.0, q));
            while (q > 0.0 && q_org[k] + q * q_cell[k] > v) q -= 1.0;  // the division rounds: step outward
            return (uint16_t)q;
        };
        auto q_up = [&](double v, int k) -> uint16_t {
            double q = std::ceil((v - q_org[k]) / q_cell[k]);
            q = std::max(0.0, std::min(65535.0, q));
            while (q < 65535.0 && q_org[k] + q * q_cell[k] < v) q += 1.0;
            return (uint16_t)q;
        };
        // hierarchies beyond 1 MB of f32 nodes are stored quantised and four wide (rtc_types.h)
        quantized_nodes = (size_t)next * sizeof(DBvhNode) > (1u << 20);
        // a leaf's reference is formed once (forming it appends to the compact list)
        std::vector<int32_t> ref_of(b.nodes.size(), 0);
        std::vector<uint8_t> ref_done(b.nodes.size(), 0);
        auto child_ref = [&](int c, const std::vector<int> &index) -> int32_t {
            if (b.nodes[c].left >= 0) return index[c];
            if (!ref_done[c]) ref_of[c] = leaf_ref(b.nodes[c]), ref_done[c] = 1;
            return ref_of[c];
        };
        if (!quantized_nodes || !RTC_BVH4) {
            bvh_q.resize(quantized_nodes ? next : 0);
            for (int t : queue) {
                const BuildNode &n = b.nodes[t];
                const BuildNode &l = b.nodes[n.left], &r = b.nodes[n.right];
                DBvhNode d{};
                DBvhNodeQ q{};
                for (int k = 0; k < 3; k++) {
                    float *lp = k == 0 ? d.lx : k == 1 ? d.ly : d.lz, *rp = k == 0 ? d.rx : k == 1 ? d.ry : d.rz;
                    lp[0] = f32_down(l.b.lo[k]), lp[1] = f32_up(l.b.hi[k]);
                    rp[0] = f32_down(r.b.lo[k]), rp[1] = f32_up(r.b.hi[k]);
                    q.lmin[k] = q_down(l.b.lo[k], k), q.lmax[k] = q_up(l.b.hi[k], k);
                    q.rmin[k] = q_down(r.b.lo[k], k), q.rmax[k] = q_up(r.b.hi[k], k);
                }
                q.left = d.left = child_ref(n.left, new_index);
                q.right = d.right = child_ref(n.right, new_index);
                bvh_flags[new_index[t]] = (uint8_t)((paired[n.left] ? 1 : 0) | (paired[n.right] ? 2 : 0));
                bvh[new_index[t]] = d;
                if (quantized_nodes) bvh_q[new_index[t]] = q;
            }
        } else {
            // ---- four-wide: a node adopts its grandchildren -- the internal child with the largest box first -- until
            // it has four children or only leaves are left; breadth-first order again ----
            auto kids_of = [&](int t, int kids[4]) {
                int n = 0;
                kids[n++] = b.nodes[t].left, kids[n++] = b.nodes[t].right;
                while (n < 4) {
                    int best = -1;
                    double best_area = -1.0;
                    for (int i = 0; i < n; i++)
                        if (b.nodes[kids[i]].left >= 0) {
                            const double a = Builder::area(b.nodes[kids[i]].b);
                            if (a > best_area) best_area = a, best = i;
                        }
                    if (best < 0) break;
                    const int c = kids[best];
                    kids[best] = b.nodes[c].left;
                    kids[n++] = b.nodes[c].right;
                }
                return n;
            };
            std::vector<int> queue4{0};
            std::vector<int> index4(b.nodes.size(), -1);
            index4[0] = 0;
            int next4 = 1;
            for (size_t head = 0; head < queue4.size(); head++) {
                int kids[4];
                const int n = kids_of(queue4[head], kids);
                for (int i = 0; i < n; i++)
                    if (b.nodes[kids[i]].left >= 0) {
                        index4[kids[i]] = next4++;
                        queue4.push_back(kids[i]);
                    }
            }
            bvh4.resize(next4);
            bvh_flags.assign(next4, 0);
            for (int t : queue4) {
                int kids[4];
                const int n = kids_of(t, kids);
                DBvhNode4 d{};
                uint8_t flags = 0;
                for (int c = 0; c < 4; c++) {
                    uint16_t lo[3] = {65535, 65535, 65535}, hi[3] = {0, 0, 0};  // unused slot: inverted box
                    d.child[c] = RTC_NO_CHILD;
                    if (c < n) {
                        const BuildNode &k = b.nodes[kids[c]];
                        for (int a = 0; a < 3; a++) lo[a] = q_down(k.b.lo[a], a), hi[a] = q_up(k.b.hi[a], a);
                        d.child[c] = child_ref(kids[c], index4);
                        if (paired[kids[c]]) flags |= (uint8_t)(1u << c);
                    }
                    d.box[3 * c + 0] = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
                    d.box[3 * c + 1] = (uint32_t)lo[2] | ((uint32_t)hi[0] << 16);
                    d.box[3 * c + 2] = (uint32_t)hi[1] | ((uint32_t)hi[2] << 16);
                }
                bvh_flags[index4[t]] = flags;
                bvh4[index4[t]] = d;
            }
        }
    }
    bvh_prims = order;
}

// ---------------------------------------------------------------------------------------------
// tile order of the static schedule
// ---------------------------------------------------------------------------------------------
// The static schedule launches one workgroup per 8x8 tile and the hardware starts them in index order.  A tile that
// shows glass (a ray tree per pixel, depth 5) takes ~100 rounds of one wave; when it starts in the middle of the frame
// the whole GPU waits for it at the end (cover.yaml at 1080p: 1.8 of 2.35 ms).  Longest job first: tiles are ordered
// by what the camera sees in them -- bounds of transparent (weight 4) and reflective (weight 1) primitives projected
// onto the canvas (camera.rs:41-59 read backwards) -- and rendered in that order.  A heuristic: it changes when a tile
// starts, never what it computes.
void Scene::build_tile_order() {
    tile_perm.clear(), tile_score.clear(), n_heavy_tiles = 0;
    static const int enabled = getenv("RTC_TILE_SORT") ? atoi(getenv("RTC_TILE_SORT")) : 1;
    const uint32_t tx = (hsize + 7u) / 8u, ty = (vsize + 7u) / 8u;
    if (!enabled || (uint64_t)tx * ty < 2 || (uint64_t)tx * ty > (1u << 24)) return;
    std::vector<int32_t> diff((size_t)(tx + 1) * (ty + 1), 0);  // 2-d difference array of the weights
    bool any = false;
    for (size_t pi = 0; pi < prims.size(); pi++) {
        if (!prim_bounded[pi] || prims[pi].material < 0) continue;
        const DMaterial &m = dmaterials[(size_t)prims[pi].material];
        const int32_t w = (m.transparency != 0.0 ? 4 : 0) + (m.reflectivity != 0.0 ? 1 : 0);
        if (w == 0) continue;
        const Bounds &b = prim_bounds[pi];
        double x0 = INF, x1 = -INF, y0 = INF, y1 = -INF;
        bool behind = false;
        for (int c = 0; c < 8 && !behind; c++) {
            const double p[3] = {(c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1], (c & 4) ? b.hi[2] : b.lo[2]};
            double q[3];
            for (int r = 0; r < 3; r++) q[r] = view.m[r][0] * p[0] + view.m[r][1] * p[1] + view.m[r][2] * p[2] + view.m[r][3];
            if (!(q[2] < -1e-6)) {  // at or behind the eye (the camera looks down -z): anywhere on the canvas
                behind = true;
                break;
            }
            const double px = (cam.half_width - q[0] / -q[2]) / cam.pixel_size;
            const double py = (cam.half_height - q[1] / -q[2]) / cam.pixel_size;
            x0 = std::min(x0, px), x1 = std::max(x1, px), y0 = std::min(y0, py), y1 = std::max(y1, py);
        }
        if (behind) x0 = y0 = 0.0, x1 = (double)hsize, y1 = (double)vsize;
        if (!(x1 >= 0.0 && y1 >= 0.0 && x0 <= (double)hsize && y0 <= (double)vsize)) continue;  // off the canvas (or NaN)
        const uint32_t t0x = (uint32_t)(std::max(0.0, std::floor(x0)) / 8.0), t0y = (uint32_t)(std::max(0.0, std::floor(y0)) / 8.0);
        const uint32_t t1x = std::min(tx - 1u, (uint32_t)(std::min((double)hsize, x1) / 8.0));
        const uint32_t t1y = std::min(ty - 1u, (uint32_t)(std::min((double)vsize, y1) / 8.0));
        if (t0x > t1x || t0y > t1y) continue;
        diff[(size_t)t0y * (tx + 1) + t0x] += w, diff[(size_t)t0y * (tx + 1) + t1x + 1] -= w;
        diff[(size_t)(t1y + 1) * (tx + 1) + t0x] -= w, diff[(size_t)(t1y + 1) * (tx + 1) + t1x + 1] += w;
        any = true;
    }
    if (!any) return;
    std::vector<int32_t> score((size_t)tx * ty);
    std::vector<int32_t> col(tx + 1, 0);
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    for (uint32_t y = 0; y < ty; y++) {
        int32_t run = 0;
        for (uint32_t x = 0; x < tx; x++) {
            col[x] += diff[(size_t)y * (tx + 1) + x];
            run += col[x];
            score[(size_t)y * tx + x] = run;
            lo = std::min(lo, run), hi = std::max(hi, run);
        }
    }
    if (lo == hi) return;  // every tile alike: raster order
    tile_perm.resize((size_t)tx * ty);
    for (uint32_t t = 0; t < tx * ty; t++) tile_perm[t] = t;
    std::stable_sort(tile_perm.begin(), tile_perm.end(), [&](uint32_t a, uint32_t c) { return score[a] > score[c]; });
    tile_score.swap(score);
    static const int heavy_env = getenv("RTC_HEAVY_SCORE") ? atoi(getenv("RTC_HEAVY_SCORE")) : 0;  // (A/B knob)
    const int32_t heavy = heavy_env > 0 ? heavy_env : RTC_HEAVY_TILE_SCORE;
    for (uint32_t t : tile_perm) n_heavy_tiles += tile_score[t] >= heavy ? 1u : 0u;  // (they come first)
}

// ---------------------------------------------------------------------------------------------
// upload
// ---------------------------------------------------------------------------------------------

template <class T>
static int upload_vec(DeviceReplica &s, const std::vector<T> &v, const T *&dptr) {
    dptr = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return RTC_ERR_HIP;
    }
    s.allocs.push_back(p);
    s.bytes += bytes;
    if (!v.empty()) {
        e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e));
            return RTC_ERR_HIP;
        }
    }
    dptr = (const T *)p;
    return RTC_OK;
}

int Scene::upload(int dev, DeviceReplica &rep) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); librtc_amd renders on the GPU only",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return RTC_ERR_NO_DEVICE;
    }
    if (dev >= count) {
        set_error("HIP device %d requested, %d present", dev, count);
        return RTC_ERR_NO_DEVICE;
    }
    DeviceGuard guard(dev);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", dev);
        return RTC_ERR_HIP;
    }
    rep.device = dev;
    DScene &dscene = rep.dscene;
    dscene = DScene{};
    {
        std::vector<DCamera> cv(1, cam);
        if (int rc0 = upload_vec(rep, cv, dscene.cam)) return rc0;
        dscene.hsize = cam.hsize, dscene.vsize = cam.vsize;
    }
    int rc;
    if ((rc = upload_vec(rep, xforms, dscene.xforms))) return rc;
    if ((rc = upload_vec(rep, chains, dscene.chains))) return rc;
    if ((rc = upload_vec(rep, prims, dscene.prims))) return rc;
    if ((rc = upload_vec(rep, tris, dscene.tris))) return rc;
    if ((rc = upload_vec(rep, dmaterials, dscene.materials))) return rc;
    if ((rc = upload_vec(rep, lights, dscene.lights))) return rc;
    if ((rc = upload_vec(rep, dpatterns, dscene.patterns))) return rc;
    if (quantized_nodes && RTC_QF32) {
        if ((rc = upload_vec(rep, bvh, dscene.bvh))) return rc;
    } else if (quantized_nodes) {
        if (RTC_BVH4) {
            if ((rc = upload_vec(rep, bvh4, dscene.bvh4))) return rc;
        } else {
            if ((rc = upload_vec(rep, bvh_q, dscene.bvh_q))) return rc;
        }
    } else {
        if ((rc = upload_vec(rep, bvh, dscene.bvh))) return rc;
    }
    dscene.quantized_nodes = quantized_nodes ? 1 : 0;
    if ((rc = upload_vec(rep, bvh_flags, dscene.node_flags))) return rc;
    for (int k = 0; k < 3; k++) dscene.q_org[k] = q_org[k], dscene.q_cell[k] = q_cell[k];
    {
        std::vector<int32_t> list(unbounded);
        list.insert(list.end(), bvh_prims.begin(), bvh_prims.end());
        // intersection-loop records, once in list order and once in primitive order
        auto make_leaf = [&](int32_t pi) {
            const DPrim &p = prims[pi];
            DLeaf l{};
            l.type = (uint8_t)p.type, l.prim = pi, l.closed = (uint8_t)p.closed;
            if (prim_csg[pi] >= 0) l.closed |= RTC_LEAF_IN_CSG;
            if (p.type == PT_CSG) {
                l.anc_off = p.tri;  // the tree
                l.anc_id = -1;
                l.minimum = l.maximum = 0;
                return l;
            }
            l.minimum = p.minimum, l.maximum = p.maximum;
            l.anc_off = p.chain_off;
            l.anc_id = prim_anc_id[pi];
            if (p.type >= PT_TRIANGLE) {
                l.anc_len = (uint16_t)p.chain_len;
                const DTri &t = tris[p.tri];
                for (int i = 0; i < 3; i++) l.a[i] = t.p1[i], l.a[3 + i] = t.e1[i], l.a[6 + i] = t.e2[i];
            } else {
                l.anc_len = (uint16_t)(p.chain_len - 1);
                const DXform &x = xforms[chains[p.chain_off + p.chain_len - 1]];
                for (int i = 0; i < 12; i++) l.a[i] = x.inv[i];
            }
            return l;
        };
        std::vector<DLeaf> lv(list.size()), la(prims.size());
        for (size_t i = 0; i < list.size(); i++) lv[i] = make_leaf(list[i]);
        for (size_t i = 0; i < prims.size(); i++) la[i] = make_leaf((int32_t)i);
        if ((rc = upload_vec(rep, lv, dscene.leaves))) return rc;
        if ((rc = upload_vec(rep, la, dscene.leaves_all))) return rc;
        std::vector<DLeafS> ls(compact_prims.size());
        for (size_t i = 0; i < compact_prims.size(); i++) {
            const DPrim &p = prims[compact_prims[i]];
            const double *a = xforms[chains[p.chain_off]].inv;
            DLeafS &l = ls[i];
            l = DLeafS{};
            l.prim = compact_prims[i];
            const int off_diag[6] = {1, 2, 4, 6, 8, 9};
            for (int k = 0; k < 6; k++)
                if (std::signbit(a[off_diag[k]])) l.zsign |= 1u << k;
            for (int k = 0; k < 3; k++) l.d[k] = a[5 * k], l.t[k] = a[4 * k + 3];
        }
        if ((rc = upload_vec(rep, ls, dscene.leaves_s))) return rc;
    }
    for (int k = 0; k < 3; k++) dscene.root_min[k] = root_min[k], dscene.root_max[k] = root_max[k];
    dscene.root_ref = root_ref;
    dscene.n_bounded = (int32_t)bvh_prims.size();
    dscene.n_prims = (int32_t)prims.size();
    dscene.n_unbounded = (int32_t)unbounded.size();
    dscene.n_bvh_nodes = (int32_t)((quantized_nodes && RTC_BVH4) ? bvh4.size() : bvh.size());
    dscene.n_lights = (int32_t)lights.size();
    dscene.n_materials = (int32_t)dmaterials.size();
    if ((rc = upload_vec(rep, csg_roots, dscene.csg_roots))) return rc;
    if ((rc = upload_vec(rep, csg_nodes, dscene.csg_nodes))) return rc;
    dscene.any_csg = csg_roots.empty() ? 0 : 1;
    dscene.all_cast_shadow = 1;
    dscene.any_pattern = 0;
    dscene.any_group = 0;
    for (int32_t id : prim_anc_id)
        if (id >= 0) dscene.any_group = 1;
    for (const DMaterial &m : dmaterials) {
        if (!m.casts_shadow) dscene.all_cast_shadow = 0;
        if (m.pattern >= 0) dscene.any_pattern = 1;
    }
    rep.d_tile_perm = nullptr;
    if (!tile_perm.empty() && (rc = upload_vec(rep, tile_perm, rep.d_tile_perm))) return rc;
    std::vector<uint64_t> zeros(ST_COUNT, 0);
    const uint64_t *dst = nullptr;
    if ((rc = upload_vec(rep, zeros, dst))) return rc;
    rep.d_stats = const_cast<uint64_t *>(dst);
    {
        // tile counters + end-of-frame ray exchange of the persistent schedule (layout: rtc_types.h)
        void *pw = nullptr;
        if (hipMalloc(&pw, RTC_WORK_TOTAL_BYTES) != hipSuccess || hipMemset(pw, 0, RTC_WORK_RESET_BYTES) != hipSuccess) {
            set_error("hipMalloc(%zu) for the work buffer failed", (size_t)RTC_WORK_TOTAL_BYTES);
            return RTC_ERR_HIP;
        }
        rep.allocs.push_back(pw);
        rep.bytes += RTC_WORK_TOTAL_BYTES;
        rep.d_work = (unsigned *)pw;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            set_error("hipGetDeviceProperties failed");
            return RTC_ERR_HIP;
        }
        rep.n_cu = (unsigned)prop.multiProcessorCount;  // CU count; the launcher multiplies by the occupancy
    }
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        set_error("hipEventCreate failed");
        return RTC_ERR_HIP;
    }
    rep.ev_start = a, rep.ev_stop = b;
    // the static scratch array of rtc_render_device_static is resolved here, once per device, with the device current
    unsigned *unused = nullptr;
    if (int rcs = static_scratch(dev, &unused)) return rcs;
    device_bytes += rep.bytes;
    return RTC_OK;
}

int Scene::commit(const int *devices, uint32_t n_devices) {
    if (committed) {
        set_error("scene already committed");
        return RTC_ERR_STATE;
    }
    if (!open_groups.empty()) {
        set_error("rtc_scene_commit with %zu unclosed group(s)", open_groups.size());
        return RTC_ERR_STATE;
    }
    if (hsize == 0 || vsize == 0) {
        set_error("camera hsize/vsize must be positive");
        return RTC_ERR_INVALID;
    }
    int rc = flatten();
    if (rc) return rc;
    if (prims.size() >= (size_t)RTC_LEAF_COMPACT) {
        set_error("%zu primitives: leaf references hold 27 bits", prims.size());
        return RTC_ERR_INVALID;
    }
    build_bvh();
    build_tile_order();
    device = -1;
    if (n_devices > 0 && devices[0] >= 0) {
        reps.resize(n_devices);
        for (uint32_t k = 0; k < n_devices; k++) {
            rc = devices[k] < 0 ? RTC_ERR_INVALID : upload(devices[k], reps[k]);
            if (rc == RTC_OK && n_devices > 1) {
                // replicas of a multi-device scene run concurrently: each renders on a stream of its own
                DeviceGuard g(devices[k]);
                hipStream_t st;
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
                    set_error("hipStreamCreate on device %d failed", devices[k]);
                    rc = RTC_ERR_HIP;
                } else {
                    reps[k].stream = st;
                }
            }
            if (rc) {
                if (devices[k] < 0) set_error("device list entry %u is negative", k);
                release_device();
                return rc;
            }
        }
        device = devices[0];
        // peer access towards the primary device (the canvas is assembled there); a failure only means that the
        // copies are staged by the runtime
        for (uint32_t k = 1; k < n_devices; k++)
            if (devices[k] != device) {
                DeviceGuard g(devices[k]);
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[k], device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(device, 0);
                (void)hipGetLastError();
            }
    }
    committed = true;
    return RTC_OK;
}

uint32_t local_rows(uint32_t vsize, uint32_t tile_rows, uint32_t tile_index, uint32_t tile_count) {
    if (tile_count == 0) tile_count = 1;
    if (tile_rows == 0 || tile_rows > vsize) tile_rows = vsize ? vsize : 1;
    const uint32_t ntiles = (vsize + tile_rows - 1) / tile_rows;
    uint32_t rows = 0;
    for (uint32_t t = tile_index; t < ntiles; t += tile_count) rows += std::min(tile_rows, vsize - t * tile_rows);
    return rows;
}

}  // namespace rtc
