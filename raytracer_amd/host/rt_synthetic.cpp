// Synthetic benchmark scene (BASELINE.json config 5): a
// cover.rs/room.rs-style world of `n_primitives` mixed primitives arranged in
// two levels of groups with random TRS transforms, one backdrop plane, the two
// lights and the camera of exercises/cover.rs:174-186.  There is no such scene
// in the reference; it exists to load the traversal, and is built only through
// the public API so it renders identically on any implementation of that API.
//
// Randomness: SplitMix64; every value is (u64 >> 11) * 2^-53, so the scene is
// bit-reproducible across languages.
#include <cmath>

#include "rt_exercises.h"

namespace rt {
namespace {

struct SplitMix64 {
    uint64_t s;
    uint64_t next_u64() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double unit() { return static_cast<double>(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    double range(double lo, double hi) { return lo + (hi - lo) * unit(); }
};

using M = Matrix4;

Material random_material(SplitMix64& rng) {
    const double u = rng.unit();
    Material m;
    const Color c(rng.range(0.2, 1.0), rng.range(0.2, 1.0), rng.range(0.2, 1.0));
    if (u < 0.70) {  // diffuse
        m.color = c;
        m.diffuse = 0.7;
        m.ambient = 0.1;
        m.specular = 0.3;
    } else if (u < 0.85) {  // reflective 0.1 .. 0.7
        m.color = c;
        m.diffuse = 0.7;
        m.ambient = 0.1;
        m.specular = 0.0;
        m.reflective = rng.range(0.1, 0.7);
    } else if (u < 0.95) {  // glass, as cover.rs:61-72
        m.color = Color(0.373, 0.404, 0.550);
        m.diffuse = 0.2;
        m.ambient = 0.0;
        m.specular = 1.0;
        m.shininess = 200.0;
        m.reflective = 0.7;
        m.transparency = 0.7;
        m.refractive_index = 1.5;
    } else {  // two-colour checker or stripe
        const Color d(rng.range(0.0, 0.6), rng.range(0.0, 0.6), rng.range(0.0, 0.6));
        const double s = rng.range(0.15, 0.5);
        Pattern p = (rng.unit() < 0.5) ? Pattern::new_checker(Pattern::new_solid(c), Pattern::new_solid(d))
                                       : Pattern::new_stripe(Pattern::new_solid(c), Pattern::new_solid(d));
        m.pattern = p.with_transform(M::rotation_y(rng.range(0.0, 1.5)) * M::scaling(s, s, s));
        m.diffuse = 0.7;
        m.ambient = 0.1;
        m.specular = 0.3;
    }
    return m;
}

Shape random_primitive(SplitMix64& rng, const Point& at, double size) {
    const double k = rng.unit();
    const Material mat = random_material(rng);
    const M place = M::translation(at.x, at.y, at.z) * M::rotation_y(rng.range(0.0, 3.0)) *
                    M::rotation_x(rng.range(0.0, 3.0));
    const double sx = size * rng.range(0.6, 1.0), sy = size * rng.range(0.6, 1.0), sz = size * rng.range(0.6, 1.0);
    if (k < 0.40)
        return Primitive::new_cube().with_material(mat).with_transform(place * M::scaling(sx, sy, sz)).to_shape();
    if (k < 0.70)
        return Primitive::new_sphere().with_material(mat).with_transform(place * M::scaling(sx, sx, sx)).to_shape();
    if (k < 0.85)
        return Primitive::new_cylinder(Conic(-1.0, 1.0, true))
            .with_material(mat)
            .with_transform(place * M::scaling(sx * 0.7, sy, sx * 0.7))
            .to_shape();
    if (k < 0.95)
        return Primitive::new_cone(Conic(-1.0, 0.0, true))
            .with_material(mat)
            .with_transform(place * M::translation(0.0, 0.5 * sy, 0.0) * M::scaling(sx, sy * 1.5, sx))
            .to_shape();
    const Point p1(rng.range(-1.0, 1.0), rng.range(-1.0, 1.0), rng.range(-1.0, 1.0));
    const Point p2(rng.range(-1.0, 1.0), rng.range(-1.0, 1.0), rng.range(-1.0, 1.0));
    const Point p3(rng.range(-1.0, 1.0), rng.range(-1.0, 1.0), rng.range(-1.0, 1.0));
    return Primitive::new_triangle(p1, p2, p3)
        .with_material(mat)
        .with_transform(place * M::scaling(size * 1.4, size * 1.4, size * 1.4))
        .to_shape();
}

}  // namespace

Exercise synthetic_scene(size_t width, size_t height, size_t n_primitives, uint64_t seed) {
    SplitMix64 rng{seed};
    constexpr size_t kPerSub = 25, kSubs = 4, kPerTop = kPerSub * kSubs;
    const size_t n_top = (n_primitives + kPerTop - 1) / kPerTop;

    // top-level lattice over a 60 x 30 x 60 box, roughly 2:1:2 cells
    size_t gy = static_cast<size_t>(std::ceil(std::cbrt(static_cast<double>(n_top) / 4.0)));
    if (gy == 0) gy = 1;
    size_t gx = static_cast<size_t>(std::ceil(std::sqrt(static_cast<double>(n_top) / static_cast<double>(gy))));
    if (gx == 0) gx = 1;
    const size_t gz = gx;
    const double x0 = -14.0, y0 = -26.0, z0 = 2.0;
    const double cx = 60.0 / static_cast<double>(gx), cy = 30.0 / static_cast<double>(gy),
                 cz = 60.0 / static_cast<double>(gz);

    Material backdrop_material;  // cover.rs:50-59
    backdrop_material.color = Color::WHITE();
    backdrop_material.ambient = 1.0;
    backdrop_material.diffuse = 0.0;
    backdrop_material.specular = 0.0;
    std::vector<Shape> shapes;
    shapes.push_back(Primitive::new_plane()
                         .with_material(backdrop_material)
                         .with_transform(M::translation(0.0, 0.0, 500.0) * M::rotation_x(1.57079632679489661923))
                         .to_shape());

    size_t made = 0;
    for (size_t g = 0; g < n_top && made < n_primitives; ++g) {
        const size_t ix = g % gx, iz = (g / gx) % gz, iy = g / (gx * gz);
        const Point center(x0 + (static_cast<double>(ix) + 0.5) * cx, y0 + (static_cast<double>(iy) + 0.5) * cy,
                           z0 + (static_cast<double>(iz) + 0.5) * cz);
        const double gs = rng.range(0.8, 1.05);
        std::vector<Shape> subs;
        for (size_t s = 0; s < kSubs && made < n_primitives; ++s) {
            // each subgroup owns one quadrant (x,z) of the cell, in group-local units
            const double qx = (s & 1) ? 0.25 : -0.25, qz = (s & 2) ? 0.25 : -0.25;
            std::vector<Shape> prims;
            for (size_t p = 0; p < kPerSub && made < n_primitives; ++p, ++made) {
                // 5 x 5 jittered lattice in the quadrant, random height
                const double lx = (static_cast<double>(p % 5) + 0.5 + rng.range(-0.3, 0.3)) / 5.0 - 0.5;
                const double lz = (static_cast<double>(p / 5) + 0.5 + rng.range(-0.3, 0.3)) / 5.0 - 0.5;
                const double ly = rng.range(-0.45, 0.45);
                const Point at(lx * 0.5 * cx, ly * cy, lz * 0.5 * cz);
                const double size = 0.32 * std::fmin(0.5 * cx / 5.0, 0.5 * cz / 5.0) * 2.0;
                prims.push_back(random_primitive(rng, at, size));
            }
            subs.push_back(Compound::new_group(prims)
                               .with_transform(M::translation(qx * cx, 0.0, qz * cz) *
                                               M::rotation_z(rng.range(-0.15, 0.15)))
                               .to_shape());
        }
        shapes.push_back(Compound::new_group(subs)
                             .with_transform(M::translation(center.x, center.y, center.z) *
                                             M::rotation_y(rng.range(-0.4, 0.4)) * M::scaling(gs, gs, gs))
                             .to_shape());
    }

    World world(shapes, {PointLight(Point(50.0, 100.0, -50.0), Color::WHITE()),
                         PointLight(Point(-400.0, 50.0, -10.0), Color(0.7, 0.7, 0.7))});  // cover.rs:176-179
    Camera camera = Camera(width, height, 0.785398163397448309615660845819875721)
                        .with_transform(M::view_transform(Point(-6.0, 6.0, -10.0), Point(6.0, -2.0, 6.0),
                                                          Vector(-0.45, 1.0, 0.0)));  // cover.rs:182-186
    return {world, camera};
}

}  // namespace rt
