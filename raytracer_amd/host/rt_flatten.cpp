// World -> flat record tables (see rt_flatten.h, include/rt_hip.h).
#include "rt_flatten.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <string>

namespace rt {
namespace {

void rows012(const Matrix4& m, double out[12]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = m.m[i][j];
}

struct Flattener {
    FlatScene out;
    FlattenOptions opt;
    std::vector<const Primitive*> prim_of;            // by shape index, nullptr for compounds
    std::map<std::string, uint32_t> material_cache;   // identical materials share one record

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

    uint32_t emit(const Shape& s, int32_t parent, uint32_t depth) {
        const uint32_t idx = static_cast<uint32_t>(out.shapes.size());
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
            return idx;
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
            for (const Shape& child : c.children()) emit(child, static_cast<int32_t>(idx), depth + 1);
        } else {
            rec.kind_flags |= static_cast<uint32_t>(c.operation()) << RT_FLAG_CSG_OP_SHIFT;
            rec.param = static_cast<uint32_t>(out.csgs.size());
            out.csgs.emplace_back();
            const size_t csg_slot = rec.param;
            emit(c.children()[0], static_cast<int32_t>(idx), depth + 1);
            emit(c.children()[1], static_cast<int32_t>(idx), depth + 1);
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
        return idx;
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
    return d;
}

FlatScene flatten(const World& world, const FlattenOptions& opt) {
    Flattener f;
    f.opt = opt;
    for (const PointLight& l : world.lights()) {
        RtLight rec;
        rec.position[0] = l.position.x, rec.position[1] = l.position.y, rec.position[2] = l.position.z;
        rec.intensity[0] = l.intensity.red, rec.intensity[1] = l.intensity.green, rec.intensity[2] = l.intensity.blue;
        f.out.lights.push_back(rec);
    }
    for (const Shape& s : world.shapes()) f.emit(s, -1, 0);
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
