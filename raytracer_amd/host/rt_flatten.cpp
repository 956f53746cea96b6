// World -> flat record tables (see rt_flatten.h, include/rt_hip.h).
#include "rt_flatten.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string>

namespace rt {
namespace {

void rows012(const Matrix4& m, double out[12]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = m.m[i][j];
}

// ---- acceleration structure (not part of the reference; never changes a result) -----------
// A box that certainly contains every point at which the reference can report a hit of the
// shape, in the space of the shape's parent.  Unlike Properties.bounds (the reference's
// quirky fold, Q5) this one is a true bound: cones use their real radius, compounds the
// union of their children.
struct ABox {
    double mn[3] = {INFINITY, INFINITY, INFINITY};  // empty
    double mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool unbounded = false;
    bool empty() const { return !unbounded && mn[0] > mx[0]; }
    void add(const Point& p) {
        const double c[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; ++a) {
            if (!std::isfinite(c[a])) unbounded = true;
            mn[a] = c[a] < mn[a] ? c[a] : mn[a];  // (comparisons, not fmin/fmax: those are libm calls; a NaN is caught above)
            mx[a] = c[a] > mx[a] ? c[a] : mx[a];
        }
    }
    void merge(const ABox& o) {
        if (o.unbounded) unbounded = true;
        if (o.empty()) return;
        for (int a = 0; a < 3; ++a) {
            mn[a] = o.mn[a] < mn[a] ? o.mn[a] : mn[a];
            mx[a] = o.mx[a] > mx[a] ? o.mx[a] : mx[a];
        }
    }
    ABox transformed(const Matrix4& t) const {
        ABox r;
        if (unbounded) {
            r.unbounded = true;
            return r;
        }
        if (empty()) return r;
        for (int k = 0; k < 8; ++k) r.add(t * Point(k & 1 ? mx[0] : mn[0], k & 2 ? mx[1] : mn[1], k & 4 ? mx[2] : mn[2]));
        return r;
    }
    // Outward padding: rounding in the reference's intersection arithmetic can accept a ray
    // that misses the exact surface by a relative ~1e-16 * (distance/size)^2; 1e-4 of the
    // box diagonal covers origins up to ~1e6 object sizes away.
    void pad() {
        if (unbounded || empty()) return;
        double diag = 0.0, big = 0.0;
        for (int a = 0; a < 3; ++a) {
            diag += (mx[a] - mn[a]) * (mx[a] - mn[a]);
            big = std::fmax(big, std::fmax(std::fabs(mn[a]), std::fabs(mx[a])));
        }
        const double e = 1e-4 * std::sqrt(diag) + 1e-9 * big + 1e-12;
        for (int a = 0; a < 3; ++a) mn[a] -= e, mx[a] += e;
    }
    double centroid(int a) const { return 0.5 * (mn[a] + mx[a]); }
    double half_area() const {
        const double x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
        return x * y + y * z + z * x;
    }
};

// `to_target` maps the primitive's PARENT space to the space the box is wanted in.
ABox primitive_box(const Primitive& p, const Matrix4& to_target) {
    const Matrix4 m = to_target * p.properties().transform;
    ABox b;
    switch (p.kind()) {
        case Primitive::Kind::Sphere:
        case Primitive::Kind::Cube:
            b.add(Point(-1.0, -1.0, -1.0)), b.add(Point(1.0, 1.0, 1.0));
            break;
        case Primitive::Kind::Plane: b.unbounded = true; break;
        case Primitive::Kind::Cylinder:
            b.add(Point(-1.0, p.conic().minimum, -1.0)), b.add(Point(1.0, p.conic().maximum, 1.0));
            break;
        case Primitive::Kind::Cone: {
            const double r = std::fmax(std::fabs(p.conic().minimum), std::fabs(p.conic().maximum));
            b.add(Point(-r, p.conic().minimum, -r)), b.add(Point(r, p.conic().maximum, r));
            break;
        }
        default: {  // a triangle is the hull of its vertices: no need to go through its local box
            ABox t;
            t.add(m * p.p1), t.add(m * p.p2), t.add(m * p.p3);
            return t;
        }
    }
    return b.transformed(m);
}

// The box of a whole subtree in the space `to_target` maps its parent's space to: the union of its
// primitives' own boxes, each carried up through the composed transforms.  (Transforming a compound's
// box as a box, level by level, inflates it at every rotated level; in the 10 000-primitive scene that
// was a third more box tests and group entries per ray.)
//
// A CSG node's box only has to hold this: a ray whose line misses it gets no hit from the node.  An operand the line
// misses contributes no entries and never toggles its in_l / in_r flag (compound.rs:142-168), and then
//   Difference   (l_hit && !in_r) || (!l_hit && in_l): without left entries nothing passes  -> the LEFT operand's box;
//   Intersection (l_hit && in_r) || (!l_hit && in_l): without either operand's entries nothing passes -> EITHER operand's
//                box, the smaller one;
//   Union        -> both.
// (Exact whatever the toggles do at grazed edges, where two ~equal hits dedupe into one and a flag stays set for the rest
// of the line: the argument needs "no entries", not "inside".  The geometric intersection of the operands' boxes would
// NOT do: a left hit far outside the right operand's box passes the filter once in_r is stuck.)
// csg.rs: tricylinder 2.2^3 -> one cylinder's box, ball 5.7 x 2 x 5.7 -> the sphere's box, hollowed_box 2.8^3 -> 2^3.
// "Left entries" are the entries with l_hit = left.includes(shape) -- STRUCTURAL equality (shape.rs:18-23, Q7), not "sits in
// the left operand": a primitive of the right operand that equals one of the left operand's counts as a left hit, and a
// primitive of the left operand with an infinite bound (never equal to itself, Q7) counts as a right one.  So
//   the LEFT operand's box holds only if no primitive of the right operand is a left hit (else right entries alone can pass);
//   the RIGHT operand's box (intersection) holds only if every primitive of the left operand is a left hit (else left
//   entries alone can toggle in_r and let the others through);
// a node for which neither holds keeps the union of both boxes.
bool any_included(const Shape& left, const Shape& s) {
    if (s.is_primitive()) return left.includes(s.primitive());
    for (const Shape& child : s.compound().children())
        if (any_included(left, child)) return true;
    return false;
}
bool all_included(const Shape& left, const Shape& s) {
    if (s.is_primitive()) return left.includes(s.primitive());
    for (const Shape& child : s.compound().children())
        if (!all_included(left, child)) return false;
    return true;
}
ABox tight_box(const Shape& s, const Matrix4& to_target) {
    if (s.is_primitive()) return primitive_box(s.primitive(), to_target);
    const Compound& c = s.compound();
    const Matrix4 m = to_target * c.properties().transform;
    if (c.kind() == Compound::Kind::Csg && c.operation() != Operation::Union) {
        const ABox l = tight_box(c.children()[0], m), r = tight_box(c.children()[1], m);
        const bool left_ok = !any_included(c.children()[0], c.children()[1]) && !l.unbounded && !l.empty();
        const bool right_ok = c.operation() == Operation::Intersection && all_included(c.children()[0], c.children()[0]) &&
                              !r.unbounded && !r.empty();
        if (left_ok && right_ok) return l.half_area() <= r.half_area() ? l : r;
        if (left_ok) return l;
        if (right_ok) return r;
        ABox b = l;
        b.merge(r);
        return b;
    }
    ABox b;
    for (const Shape& child : c.children()) b.merge(tight_box(child, m));
    return b;
}

// Cylinders and cones can return a hit outside their box (primitive.rs:486-490: a ray parallel to the surface, no y-range
// check); the direction index makes up for that at the world level and for whole CSG nodes, not inside one.
bool holds_conic(const Shape& s) {
    if (s.is_primitive()) return s.primitive().kind() == Primitive::Kind::Cylinder || s.primitive().kind() == Primitive::Kind::Cone;
    for (const Shape& child : s.compound().children())
        if (holds_conic(child)) return true;
    return false;
}

// f32 bounds that contain the f64 ones
inline float round_down(double x) {
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float round_up(double x) {
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x) f = std::nextafterf(f, INFINITY);
    return f;
}

struct BvhItem {
    ABox box;
    uint32_t shape;
};

// Skip-link layout in depth-first order: node i's subtree is [i, miss); a walk goes to i+1 on
// a box hit and to `miss` on a miss or after a leaf.  Binned surface-area split.
// The subtree of n items occupies nodes[self, self + 2n - 1): the two halves of a split write disjoint ranges of `items`
// and `nodes`, so the upper `par_depth` levels of a large tree hand their left halves to threads of their own (the tree is
// the same whatever the number of threads).
void build_bvh_range(std::vector<BvhItem>& items, size_t lo, size_t hi, int32_t after, RtBvhNode* nodes, size_t self, int par_depth) {
    const size_t n = hi - lo;
    ABox all, cent;
    for (size_t i = lo; i < hi; ++i) {
        all.merge(items[i].box);
        cent.add(Point(items[i].box.centroid(0), items[i].box.centroid(1), items[i].box.centroid(2)));
    }
    RtBvhNode node;
    for (int a = 0; a < 3; ++a) node.bmin[a] = round_down(all.mn[a]), node.bmax[a] = round_up(all.mx[a]);
    node.miss = after;
    if (n == 1) {
        node.shape = static_cast<int32_t>(items[lo].shape);
        nodes[self] = node;
        return;
    }
    node.shape = -1;
    // choose the split
    int best_axis = -1, best_bin = 0;
    double best_c0 = 0.0, best_scale = 0.0;
    size_t best_mid = lo + n / 2;
    double best_cost = INFINITY;
    constexpr int kBins = 32;
    for (int a = 0; a < 3; ++a) {
        const double c0 = cent.mn[a], c1 = cent.mx[a];
        if (!(c1 > c0)) continue;
        ABox bin_box[kBins];
        size_t bin_n[kBins] = {};
        const double scale = kBins / (c1 - c0);
        for (size_t i = lo; i < hi; ++i) {
            int b = static_cast<int>((items[i].box.centroid(a) - c0) * scale);
            b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            bin_box[b].merge(items[i].box);
            ++bin_n[b];
        }
        double right_area[kBins];
        size_t right_n[kBins];
        ABox acc;
        size_t cnt = 0;
        for (int b = kBins - 1; b > 0; --b) {
            acc.merge(bin_box[b]);
            cnt += bin_n[b];
            right_area[b] = cnt ? acc.half_area() : 0.0;
            right_n[b] = cnt;
        }
        acc = ABox();
        cnt = 0;
        for (int b = 0; b + 1 < kBins; ++b) {
            acc.merge(bin_box[b]);
            cnt += bin_n[b];
            if (cnt == 0 || right_n[b + 1] == 0) continue;
            const double cost = acc.half_area() * static_cast<double>(cnt) +
                                right_area[b + 1] * static_cast<double>(right_n[b + 1]);
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = a;
                best_bin = b;  // bins [0, b] go left
                best_c0 = c0, best_scale = scale;
                best_mid = lo + cnt;
            }
        }
    }
    const auto first = items.begin() + static_cast<long>(lo), last = items.begin() + static_cast<long>(hi);
    if (best_axis >= 0 && best_mid > lo && best_mid < hi) {
        // the items of bins [0, best_bin] to the left, order kept: linear, and the same tree whatever the input order
        std::stable_partition(first, last, [&](const BvhItem& x) {
            int b = static_cast<int>((x.box.centroid(best_axis) - best_c0) * best_scale);
            b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            return b <= best_bin;
        });
    } else {  // every centroid in one spot: halve by shape index
        std::sort(first, last, [](const BvhItem& x, const BvhItem& y) { return x.shape < y.shape; });
        best_mid = lo + n / 2;
    }
    nodes[self] = node;
    const size_t n_left = best_mid - lo;
    const size_t right_self = self + 1 + (2 * n_left - 1);
    const int32_t right_root = static_cast<int32_t>(right_self);
    if (par_depth > 0 && n >= 512) {
        std::thread left([&] { build_bvh_range(items, lo, best_mid, right_root, nodes, self + 1, par_depth - 1); });
        build_bvh_range(items, best_mid, hi, after, nodes, right_self, par_depth - 1);
        left.join();
    } else {
        build_bvh_range(items, lo, best_mid, right_root, nodes, self + 1, 0);
        build_bvh_range(items, best_mid, hi, after, nodes, right_self, 0);
    }
}

// Returns the root index, or -1 when nothing was built.
int32_t build_bvh(std::vector<BvhItem> items, std::vector<RtBvhNode>& nodes) {
    std::vector<BvhItem> bounded, unbounded;
    for (BvhItem& it : items) {
        if (it.box.empty()) continue;  // nothing to hit in there
        it.box.pad();
        (it.box.unbounded ? unbounded : bounded).push_back(it);
    }
    if (bounded.empty() && unbounded.empty()) return -1;
    const int32_t root = static_cast<int32_t>(nodes.size());
    const size_t tree_nodes = bounded.empty() ? 0 : 2 * bounded.size() - 1;
    const int32_t first_unbounded = static_cast<int32_t>(nodes.size() + tree_nodes);
    if (!bounded.empty()) {
        nodes.resize(nodes.size() + tree_nodes);
        unsigned hw = std::thread::hardware_concurrency();
        int par_depth = hw >= 16 ? 4 : (hw >= 8 ? 3 : (hw >= 4 ? 2 : (hw >= 2 ? 1 : 0)));
        if (const char* e = std::getenv("RT_FLATTEN_THREADS")) par_depth = std::atoi(e) <= 1 ? 0 : par_depth;  // study knob
        build_bvh_range(bounded, 0, bounded.size(), unbounded.empty() ? -1 : first_unbounded, nodes.data(), static_cast<size_t>(root), par_depth);
    }
    for (size_t k = 0; k < unbounded.size(); ++k) {  // always-visited leaves after the tree
        RtBvhNode leaf;
        for (int a = 0; a < 3; ++a) leaf.bmin[a] = -INFINITY, leaf.bmax[a] = INFINITY;
        leaf.shape = static_cast<int32_t>(unbounded[k].shape);
        leaf.miss = k + 1 < unbounded.size() ? first_unbounded + static_cast<int32_t>(k) + 1 : -1;
        nodes.push_back(leaf);
    }
    return root;
}

struct Flattener {
    FlatScene out;
    FlattenOptions opt;
    int csg_depth = 0;  // > 0 while emitting the operands of a CSG node: no acceleration structures there
    bool global = false;             // one world-space hierarchy over all leaves instead of one per group level
    std::vector<BvhItem> leaves;     // global: every primitive outside CSG nodes and every outermost CSG node
    std::vector<const Primitive*> prim_of;            // by shape index, nullptr for compounds
    std::unordered_map<std::string, uint32_t> material_cache;  // identical materials share one record

    static void key_append(std::string& k, const void* p, size_t n) { k.append(static_cast<const char*>(p), n); }
    static void pattern_key(std::string& k, const Pattern& p) {
        const uint32_t kind = static_cast<uint32_t>(p.kind());
        key_append(k, &kind, sizeof kind);
        key_append(k, p.inverse_transform().m, sizeof(double) * 16);
        key_append(k, &p.color(), sizeof(Color));
        if (p.a()) pattern_key(k, *p.a());
        if (p.b()) pattern_key(k, *p.b());
    }

    int32_t add_pattern(const Pattern& p) {
        const int32_t idx = static_cast<int32_t>(out.patterns.size());
        out.patterns.emplace_back();
        RtPattern rec;
        std::memset(&rec, 0, sizeof rec);
        rows012(p.inverse_transform(), rec.inv);
        rec.color[0] = p.color().red, rec.color[1] = p.color().green, rec.color[2] = p.color().blue;
        rec.kind = static_cast<uint32_t>(p.kind());
        rec.a = p.a() ? add_pattern(*p.a()) : -1;
        rec.b = p.b() ? add_pattern(*p.b()) : -1;
        out.patterns[static_cast<size_t>(idx)] = rec;
        return idx;
    }

    uint32_t add_material(const Material& m) {
        std::string key;
        key_append(key, &m.color, sizeof(Color));
        const double s[7] = {m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency,
                             m.refractive_index};
        key_append(key, s, sizeof s);
        if (m.pattern) {
            key.push_back('P');
            pattern_key(key, *m.pattern);
        }
        auto it = material_cache.find(key);
        if (it != material_cache.end()) return it->second;
        RtMaterial rec;
        std::memset(&rec, 0, sizeof rec);
        rec.color[0] = m.color.red, rec.color[1] = m.color.green, rec.color[2] = m.color.blue;
        rec.ambient = m.ambient, rec.diffuse = m.diffuse, rec.specular = m.specular, rec.shininess = m.shininess;
        rec.reflective = m.reflective, rec.transparency = m.transparency, rec.refractive_index = m.refractive_index;
        rec.pattern = m.pattern ? add_pattern(*m.pattern) : -1;
        const uint32_t idx = static_cast<uint32_t>(out.materials.size());
        out.materials.push_back(rec);
        material_cache.emplace(std::move(key), idx);
        return idx;
    }

    // A level is walked through an acceleration structure when it is large, or when it holds a
    // compound: culling a whole group or CSG subtree (the reference gives CSG nodes no gate at
    // all, compound.rs:106-116) pays even among a handful of siblings.  Small levels of plain
    // primitives stay linear: every lane then walks the same sequence.
    bool wants_bvh(size_t n, const std::vector<Shape>& children) const {
        if (n >= opt.bvh_min_children) return true;
        if (n < 2) return false;
        for (const Shape& c : children)
            if (!c.is_primitive()) return true;
        return false;
    }

    // Appends the subtree of `s`.
    // `to_world` maps the space of `s`'s parent to world space (only used for the global hierarchy).
    void emit(const Shape& s, int32_t parent, uint32_t depth, const Matrix4& to_world = Matrix4::identity()) {
        const uint32_t idx = static_cast<uint32_t>(out.shapes.size());
        if (global && csg_depth == 0 && (s.is_primitive() || s.compound().kind() == Compound::Kind::Csg))
            leaves.push_back(BvhItem{tight_box(s, to_world), idx});
        out.shapes.emplace_back();
        prim_of.push_back(nullptr);
        RtShape rec;
        std::memset(&rec, 0, sizeof rec);
        rows012(s.properties().inverse_transform, rec.inv);
        rec.parent = parent;
        rec.depth = depth;
        if (s.is_primitive()) {
            const Primitive& p = s.primitive();
            prim_of[idx] = &p;
            rec.kind_flags = static_cast<uint32_t>(p.kind());
            if (p.has_shadow()) rec.kind_flags |= RT_FLAG_HAS_SHADOW;
            rec.material = add_material(p.properties().material);
            switch (p.kind()) {
                case Primitive::Kind::Cylinder:
                case Primitive::Kind::Cone:
                    if (p.conic().closed) rec.kind_flags |= RT_FLAG_CLOSED;
                    rec.param = static_cast<uint32_t>(out.conics.size());
                    out.conics.push_back(RtConic{p.conic().minimum, p.conic().maximum});
                    break;
                case Primitive::Kind::Triangle:
                case Primitive::Kind::SmoothTriangle: {
                    RtTriangle t;
                    std::memset(&t, 0, sizeof t);
                    t.p1[0] = p.p1.x, t.p1[1] = p.p1.y, t.p1[2] = p.p1.z;
                    t.e1[0] = p.e1.x, t.e1[1] = p.e1.y, t.e1[2] = p.e1.z;
                    t.e2[0] = p.e2.x, t.e2[1] = p.e2.y, t.e2[2] = p.e2.z;
                    t.n1[0] = p.n1.x, t.n1[1] = p.n1.y, t.n1[2] = p.n1.z;
                    if (p.kind() == Primitive::Kind::SmoothTriangle) {
                        t.n2[0] = p.n2.x, t.n2[1] = p.n2.y, t.n2[2] = p.n2.z;
                        t.n3[0] = p.n3.x, t.n3[1] = p.n3.y, t.n3[2] = p.n3.z;
                    }
                    rec.param = static_cast<uint32_t>(out.triangles.size());
                    out.triangles.push_back(t);
                    break;
                }
                default: break;
            }
            rec.skip = idx + 1;
            out.shapes[idx] = rec;
            return;
        }

        const Compound& c = s.compound();
        rec.kind_flags = static_cast<uint32_t>(c.kind());
        if (c.kind() == Compound::Kind::Group) {
            // the box the reference rebuilds for every ray: compound.rs:93-94
            Bounds gate = c.properties().bounds;
            gate.transform(c.properties().transform);
            RtGroup g;
            std::memset(&g, 0, sizeof g);
            g.bmin[0] = gate.minimum.x, g.bmin[1] = gate.minimum.y, g.bmin[2] = gate.minimum.z;
            g.bmax[0] = gate.maximum.x, g.bmax[1] = gate.maximum.y, g.bmax[2] = gate.maximum.z;
            g.bvh_root = -1;
            g.n_children = static_cast<uint32_t>(c.children().size());
            rec.param = static_cast<uint32_t>(out.groups.size());
            out.groups.push_back(g);
            const size_t group_slot = rec.param;
            // a group inside a CSG node gets a hierarchy of its own only if no cylinder or cone sits below it (csg.rs: the
            // twelve wedges the ball is cut with)
            const bool tree = opt.build_bvh && !global && (csg_depth == 0 || !holds_conic(s));
            std::vector<BvhItem> items;
            const Matrix4 below = global ? to_world * c.properties().transform : to_world;
            for (const Shape& child : c.children()) {
                const uint32_t child_idx = static_cast<uint32_t>(out.shapes.size());
                emit(child, static_cast<int32_t>(idx), depth + 1, below);
                if (tree) items.push_back(BvhItem{tight_box(child, Matrix4::identity()), child_idx});
            }
            if (tree && wants_bvh(c.children().size(), c.children()))
                out.groups[group_slot].bvh_root = build_bvh(items, out.bvh_nodes);
        } else {
            rec.kind_flags |= static_cast<uint32_t>(c.operation()) << RT_FLAG_CSG_OP_SHIFT;
            rec.param = static_cast<uint32_t>(out.csgs.size());
            out.csgs.emplace_back();
            const size_t csg_slot = rec.param;
            ++csg_depth;
            emit(c.children()[0], static_cast<int32_t>(idx), depth + 1);
            emit(c.children()[1], static_cast<int32_t>(idx), depth + 1);
            --csg_depth;
            // l_hit table: left.includes(p) for every primitive p of the subtree (compound.rs:154, Q7)
            RtCsg cs;
            cs.op = static_cast<uint32_t>(c.operation());
            cs.lhit_offset = static_cast<uint32_t>(out.lhit.size());
            const uint32_t end = static_cast<uint32_t>(out.shapes.size());
            const Shape& left = c.children()[0];
            for (uint32_t p = idx + 1; p < end; ++p)
                out.lhit.push_back(prim_of[p] && left.includes(*prim_of[p]) ? 1 : 0);
            out.csgs[csg_slot] = cs;
        }
        rec.skip = static_cast<uint32_t>(out.shapes.size());
        out.shapes[idx] = rec;
    }
};

}  // namespace

RtSceneDesc FlatScene::desc() const {
    RtSceneDesc d;
    std::memset(&d, 0, sizeof d);
    d.n_shapes = static_cast<uint32_t>(shapes.size()), d.shapes = shapes.data();
    d.n_conics = static_cast<uint32_t>(conics.size()), d.conics = conics.data();
    d.n_triangles = static_cast<uint32_t>(triangles.size()), d.triangles = triangles.data();
    d.n_groups = static_cast<uint32_t>(groups.size()), d.groups = groups.data();
    d.n_csgs = static_cast<uint32_t>(csgs.size()), d.csgs = csgs.data();
    d.n_lhit = static_cast<uint32_t>(lhit.size()), d.lhit = lhit.data();
    d.n_materials = static_cast<uint32_t>(materials.size()), d.materials = materials.data();
    d.n_patterns = static_cast<uint32_t>(patterns.size()), d.patterns = patterns.data();
    d.n_lights = static_cast<uint32_t>(lights.size()), d.lights = lights.data();
    d.n_bvh_nodes = static_cast<uint32_t>(bvh_nodes.size()), d.bvh_nodes = bvh_nodes.data();
    d.world_bvh_root = world_bvh_root;
    d.dir_res = dir_res;
    d.n_dir_cands = static_cast<uint32_t>(dir_cands.size()), d.dir_cands = dir_cands.data();
    d.n_dir_entries = static_cast<uint32_t>(dir_entries.size()), d.dir_entries = dir_entries.data();
    d.dir_cell_start = dir_cell_start.empty() ? nullptr : dir_cell_start.data();  // null: the library builds the cells
    if (dir_entries.empty()) d.dir_entries = nullptr;
    return d;
}

namespace {

// a(d) = d^T Q d for a candidate
double quad_form(const double q[6], double x, double y, double z) {
    return q[0] * x * x + q[1] * y * y + q[2] * z * z + 2.0 * (q[3] * x * y + q[4] * x * z + q[5] * y * z);
}

// Direction of cube-map point (face, s, t), s,t in [-1,1]: faces +x -x +y -y +z -z.
void cube_dir(int face, double s, double t, double d[3]) {
    switch (face) {
        case 0: d[0] = 1.0, d[1] = s, d[2] = t; break;
        case 1: d[0] = -1.0, d[1] = s, d[2] = t; break;
        case 2: d[0] = s, d[1] = 1.0, d[2] = t; break;
        case 3: d[0] = s, d[1] = -1.0, d[2] = t; break;
        case 4: d[0] = s, d[1] = t, d[2] = 1.0; break;
        default: d[0] = s, d[1] = t, d[2] = -1.0; break;
    }
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] /= n, d[1] /= n, d[2] /= n;
}

// Conservative rasterisation of {unit u : |u^T Q u| <= thr} into cube-map cells by quadtree
// descent.  For unit u, c with r = |u - c|:  a(u) - a(c) = 2 (Qc).(u-c) + (u-c)^T Q (u-c), so
// |a(u) - a(c)| <= 2 |Qc| r + |Q|_F r^2; a region of half-width h in (s,t) lies within chord
// distance h*sqrt(2) of its centre.
void raster_cand(const RtDirCand& c, double qnorm, uint32_t cand, uint32_t res, int face, uint32_t i0, uint32_t j0,
                 uint32_t size, std::vector<std::pair<uint32_t, uint32_t>>& out) {
    const double h = static_cast<double>(size) / static_cast<double>(res);  // half-width in s,t units
    const double sc = -1.0 + (2.0 * i0 + size) / static_cast<double>(res);
    const double tc = -1.0 + (2.0 * j0 + size) / static_cast<double>(res);
    double d[3];
    cube_dir(face, sc, tc, d);
    const double reach = h * 1.4142135623730951 * 1.000001 + 1e-9;
    const double gx = c.q[0] * d[0] + c.q[3] * d[1] + c.q[4] * d[2];
    const double gy = c.q[3] * d[0] + c.q[1] * d[1] + c.q[5] * d[2];
    const double gz = c.q[4] * d[0] + c.q[5] * d[1] + c.q[2] * d[2];
    const double grad = std::sqrt(gx * gx + gy * gy + gz * gz);
    const double slack = (2.0 * grad * reach + qnorm * reach * reach) * 1.000001;
    if (std::fabs(quad_form(c.q, d[0], d[1], d[2])) > c.thr + slack) return;
    if (size == 1) {
        out.emplace_back((static_cast<uint32_t>(face) * res + j0) * res + i0, cand);
        return;
    }
    const uint32_t half = size / 2;
    raster_cand(c, qnorm, cand, res, face, i0, j0, half, out);
    raster_cand(c, qnorm, cand, res, face, i0 + half, j0, half, out);
    raster_cand(c, qnorm, cand, res, face, i0, j0 + half, half, out);
    raster_cand(c, qnorm, cand, res, face, i0 + half, j0 + half, half, out);
}

void build_direction_index(FlatScene& f, uint32_t res, size_t device_min) {
    // res must be a power of two
    uint32_t r2 = 1;
    while (r2 * 2 <= res) r2 *= 2;
    res = r2;
    for (uint32_t i = 0; i < f.shapes.size(); ++i) {
        const RtShape& s = f.shapes[i];
        const unsigned kind = RT_KIND(s.kind_flags);
        if (kind != RT_CYLINDER && kind != RT_CONE) continue;
        const RtConic& cn = f.conics[s.param];
        if (!std::isfinite(cn.minimum) && !std::isfinite(cn.maximum)) continue;  // its box is unbounded: always visited
        // ancestors, innermost first
        bool accelerated = f.world_bvh_root >= 0;
        uint32_t visit = i;
        double a[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) a[r][c] = s.inv[r * 4 + c];
        for (int32_t p = s.parent; p >= 0; p = f.shapes[static_cast<size_t>(p)].parent) {
            const RtShape& ps = f.shapes[static_cast<size_t>(p)];
            if (RT_KIND(ps.kind_flags) == RT_CSG) visit = static_cast<uint32_t>(p);
            if (RT_KIND(ps.kind_flags) == RT_GROUP && f.groups[ps.param].bvh_root >= 0) accelerated = true;
            double m[3][3];  // a = a * inv3(parent): local direction = a_self (a_parent d)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    m[r][c] = 0.0;
                    for (int k = 0; k < 3; ++k) m[r][c] += a[r][k] * ps.inv[k * 4 + c];
                }
            std::memcpy(a, m, sizeof a);
        }
        if (!accelerated) continue;
        const double dy = kind == RT_CONE ? -1.0 : 0.0;  // a = lx^2 + dy*ly^2 + lz^2 (primitive.rs:315,323)
        auto qij = [&](int x, int y) { return a[0][x] * a[0][y] + dy * a[1][x] * a[1][y] + a[2][x] * a[2][y]; };
        RtDirCand c;
        c.q[0] = qij(0, 0), c.q[1] = qij(1, 1), c.q[2] = qij(2, 2), c.q[3] = qij(0, 1), c.q[4] = qij(0, 2), c.q[5] = qij(1, 2);
        const double qnorm = std::sqrt(c.q[0] * c.q[0] + c.q[1] * c.q[1] + c.q[2] * c.q[2] +
                                       2.0 * (c.q[3] * c.q[3] + c.q[4] * c.q[4] + c.q[5] * c.q[5]));
        c.thr = 1e-4 * (1.0 + 1e-5) + 1e-9 * qnorm;
        c.shape = i;
        c.visit = visit;
        f.dir_cands.push_back(c);
    }
    if (f.dir_cands.empty()) return;
    f.dir_res = res;
    // many candidates: the cells are left to the library, which rasterises them on the device at a resolution of its
    // choosing (rt_dirindex.hip; 2 526 candidates at 256 cells per edge: 100 ms on 16 host threads, < 1 ms there)
    if (f.dir_cands.size() >= device_min) return;
    // candidates are rasterised independently: blocks of them on host threads, merged in candidate order
    // so that the table does not depend on the number of threads
    unsigned n_threads = std::thread::hardware_concurrency();
    n_threads = n_threads == 0 ? 1 : (n_threads > 16 ? 16 : n_threads);
    if (f.dir_cands.size() < 64) n_threads = 1;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> part(n_threads);
    auto work = [&](unsigned t) {
        const size_t k0 = f.dir_cands.size() * t / n_threads, k1 = f.dir_cands.size() * (t + 1) / n_threads;
        for (size_t k = k0; k < k1; ++k) {
            const RtDirCand& c = f.dir_cands[k];
            const double qnorm = (c.thr - 1e-4 * (1.0 + 1e-5)) / 1e-9;
            for (int face = 0; face < 6; ++face)
                raster_cand(c, qnorm, static_cast<uint32_t>(k), res, face, 0, 0, res, part[t]);
        }
    };
    if (n_threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
        for (std::thread& th : pool) th.join();
    }
    const size_t n_cells = static_cast<size_t>(6) * res * res;
    f.dir_res = res;
    f.dir_cell_start.assign(n_cells + 1, 0);
    size_t n_pairs = 0;
    for (const auto& pairs : part) {
        n_pairs += pairs.size();
        for (const auto& pr : pairs) ++f.dir_cell_start[pr.first + 1];
    }
    for (size_t c = 0; c < n_cells; ++c) f.dir_cell_start[c + 1] += f.dir_cell_start[c];
    f.dir_entries.resize(n_pairs);
    std::vector<uint32_t> fill(f.dir_cell_start.begin(), f.dir_cell_start.end() - 1);
    for (const auto& pairs : part)
        for (const auto& pr : pairs) f.dir_entries[fill[pr.first]++] = pr.second;
}

}  // namespace

FlatScene flatten(const World& world, const FlattenOptions& opt) {
    Flattener f;
    f.opt = opt;
    for (const PointLight& l : world.lights()) {
        RtLight rec;
        rec.position[0] = l.position.x, rec.position[1] = l.position.y, rec.position[2] = l.position.z;
        rec.intensity[0] = l.intensity.red, rec.intensity[1] = l.intensity.green, rec.intensity[2] = l.intensity.blue;
        f.out.lights.push_back(rec);
    }
    // Policy: scenes made of MANY SMALL groups get the global hierarchy -- going in and out of a group (its
    // transform, its gate, a hierarchy of its own) per level is what costs there: the 10 000-primitive
    // scene, 500 groups of 25 and 4, 89 -> 64 ms; hexagon.rs 0.86 -> 0.81.  A few large groups (the teapot:
    // one group of 9 216 triangles, 1.97 -> 2.09 ms with the global form, which repeats the way down for
    // every leaf) and worlds of plain primitives keep the nested form.
    size_t n_leaves = 0, n_groups = 0;
    bool nested_groups = false, has_csg = false;
    {
        std::vector<const Shape*> todo;
        for (const Shape& s : world.shapes()) todo.push_back(&s);
        while (!todo.empty()) {
            const Shape* s = todo.back();
            todo.pop_back();
            if (s->is_primitive() || s->compound().kind() == Compound::Kind::Csg) {
                ++n_leaves;
                has_csg = has_csg || !s->is_primitive();
                continue;
            }
            nested_groups = true;
            ++n_groups;
            for (const Shape& c : s->compound().children()) todo.push_back(&c);
        }
    }
    // Thousands of primitives and no CSG node: the library walks such a world through its 4-wide hierarchy (rt_bvh4.h),
    // which is built over the global form (the teapot 1.63 -> 1.31 ms).
    f.global = opt.build_bvh && (opt.hierarchy == 2 || (opt.hierarchy == 0 && nested_groups && n_leaves >= opt.bvh_min_children &&
                                                         (n_leaves < 64 * n_groups || (n_leaves >= 2048 && !has_csg))));
    f.out.shapes.reserve(n_leaves + n_groups), f.prim_of.reserve(n_leaves + n_groups), f.out.groups.reserve(n_groups);
    f.material_cache.reserve(n_leaves);
    if (f.global) f.leaves.reserve(n_leaves);
    const bool timing = std::getenv("RT_FLATTEN_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t0 = now();
    std::vector<BvhItem> top;
    for (const Shape& s : world.shapes()) {
        const uint32_t idx = static_cast<uint32_t>(f.out.shapes.size());
        f.emit(s, -1, 0);
        if (opt.build_bvh && !f.global) top.push_back(BvhItem{tight_box(s, Matrix4::identity()), idx});
    }
    if (timing) std::fprintf(stderr, "flatten: tables %.2f ms\n", ms_since(t0)), t0 = now();
    if (f.global) {
        if (!f.leaves.empty()) f.out.world_bvh_root = build_bvh(f.leaves, f.out.bvh_nodes);
    } else if (opt.build_bvh && f.wants_bvh(world.shapes().size(), world.shapes())) {
        f.out.world_bvh_root = build_bvh(top, f.out.bvh_nodes);
    }
    if (timing) std::fprintf(stderr, "flatten: hierarchy %.2f ms\n", ms_since(t0)), t0 = now();
    if (opt.build_bvh && opt.dir_res > 0) build_direction_index(f.out, opt.dir_res, opt.dir_device_min);
    if (timing) std::fprintf(stderr, "flatten: direction index %.2f ms (%zu candidates, %zu entries)\n", ms_since(t0), f.out.dir_cands.size(), f.out.dir_entries.size());
    return std::move(f.out);
}

RtCameraDesc camera_desc(const Camera& camera) {
    RtCameraDesc d;
    std::memset(&d, 0, sizeof d);
    d.hsize = static_cast<uint32_t>(camera.h_size());
    d.vsize = static_cast<uint32_t>(camera.v_size());
    d.half_width = camera.half_width();
    d.half_height = camera.half_height();
    d.pixel_size = camera.pixel_size();
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) d.inv[i * 4 + j] = camera.inverse_transform().m[i][j];
    return d;
}

}  // namespace rt
