// Host-side mirror of the reference's scene-building API, in C++ because no
// Rust toolchain exists in the build image.  Same names, same argument meaning,
// same error behaviour (a singular transform throws where the reference asserts).
//
//   reference                               here
//   ---------------------------------------------------------------------------
//   Primitive::new_sphere() ...             rt::Primitive::new_sphere() ...   (primitive.rs:52-108)
//   SetProperties::with_transform/material  .with_transform() .with_material() (shape.rs:62-73)
//   Compound::new_group / new_csg           rt::Compound::new_group / new_csg  (compound.rs:52-62)
//   Pattern::new_* / with_transform         rt::Pattern::new_* / with_transform (pattern.rs:29-74)
//   Material { .. Default }                 rt::Material (aggregate, same defaults) (material.rs:18-32)
//   PointLight::new                         rt::PointLight                     (point_light.rs:5-17)
//   World::new / empty / default / color_at rt::World                          (world.rs:24-43,130-145)
//   Camera::new / with_transform / render   rt::Camera                         (camera.rs:24-66)
//   Canvas                                  rt::Canvas                         (canvas.rs:7-90)
//   obj::parse_file                         rt::obj::parse_file                (obj.rs:55-60)
//
// Nothing here traces a ray on the CPU.  World::color_at and Camera::render
// flatten the scene (rt_flatten.cpp) and call the HIP library through the C ABI
// of include/rt_hip.h; they throw rt::RenderError if that library reports an
// error (including "no GPU").
#pragma once
#include <cstdint>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "rt_math.h"

namespace rt {

struct RenderError : std::runtime_error {
    int code;
    RenderError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// ---------------------------------------------------------------- Pattern ---
// reference: src/graphics/pattern.rs:13-101
class Pattern {
public:
    enum class Kind : uint32_t {  // numbering == RT_PAT_* of rt_hip.h
        Solid = 0, Perturb = 1, Blend = 2, Stripe = 3, Gradient = 4,
        RadialGradient = 5, Ring = 6, Checker = 7, Test = 8
    };
    static Pattern new_solid(const Color& c);
    static Pattern new_blend(const Pattern& a, const Pattern& b) { return mixture(Kind::Blend, a, b); }
    static Pattern new_perturb(const Pattern& p);
    static Pattern new_stripe(const Pattern& a, const Pattern& b) { return mixture(Kind::Stripe, a, b); }
    static Pattern new_gradient(const Pattern& a, const Pattern& b) { return mixture(Kind::Gradient, a, b); }
    static Pattern new_radial_gradient(const Pattern& a, const Pattern& b) { return mixture(Kind::RadialGradient, a, b); }
    static Pattern new_ring(const Pattern& a, const Pattern& b) { return mixture(Kind::Ring, a, b); }
    static Pattern new_checker(const Pattern& a, const Pattern& b) { return mixture(Kind::Checker, a, b); }
    static Pattern new_test();  // pattern.rs:65-68 (cfg(test) there; kept for the KATs)

    Pattern with_transform(const Matrix4& t) const;  // pattern.rs:70-74

    Kind kind() const { return kind_; }
    const Color& color() const { return color_; }
    const Matrix4& transform() const { return transform_; }
    const Matrix4& inverse_transform() const { return inverse_; }
    bool transform_was_set() const { return transform_set_; }
    const Pattern* a() const { return a_.get(); }
    const Pattern* b() const { return b_.get(); }
    bool structural_eq(const Pattern& o) const;  // derived PartialEq, pattern.rs:13,94

private:
    static Pattern mixture(Kind k, const Pattern& a, const Pattern& b);
    Kind kind_ = Kind::Solid;
    Color color_;
    Matrix4 transform_ = Matrix4::identity();
    Matrix4 inverse_ = Matrix4::identity();
    bool transform_set_ = false;
    std::shared_ptr<const Pattern> a_, b_;
};

// --------------------------------------------------------------- Material ---
// reference: src/raytracer/material.rs:5-32 (defaults) and :34-60 (equality)
struct Material {
    Color color = Color::WHITE();
    double ambient = 0.1;
    double diffuse = 0.9;
    double specular = 0.9;
    double shininess = 200.0;
    double reflective = 0.0;
    double transparency = 0.0;
    double refractive_index = 1.0;
    std::optional<Pattern> pattern;

    bool structural_eq(const Material& o) const;
};

// reference: src/raytracer/point_light.rs:5-17
struct PointLight {
    Point position;
    Color intensity;
    PointLight() = default;
    PointLight(const Point& p, const Color& i) : position(p), intensity(i) {}
};

// ----------------------------------------------------------------- Bounds ---
// reference: src/raytracer/bounds.rs:8-87.  Only the construction-time half is
// needed on the host (transform / union / add_point); the ray test runs on the
// device against the box computed here.
struct Bounds {
    Point minimum{std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(),
                  std::numeric_limits<double>::infinity()};
    Point maximum{-std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity(),
                  -std::numeric_limits<double>::infinity()};
    Bounds() = default;
    Bounds(const Point& mn, const Point& mx) : minimum(mn), maximum(mx) {}
    void add_point(const Point& p);        // bounds.rs:44-51 (f64::min/max ignore NaN)
    void union_with(const Bounds& o);      // bounds.rs:38-41
    void transform(const Matrix4& t);      // bounds.rs:19-35 (only ever grows)
    bool structural_eq(const Bounds& o) const;  // derived PartialEq over Point (1e-4)
};

// reference: src/raytracer/shapes/primitive.rs:453-471, 543-571
struct Conic {
    double minimum = -std::numeric_limits<double>::infinity();
    double maximum = std::numeric_limits<double>::infinity();
    bool closed = false;
    Conic() = default;
    Conic(double mn, double mx, bool c) : minimum(mn), maximum(mx), closed(c) {}
};

// reference: src/raytracer/shapes/compound.rs:135-139
enum class Operation : uint32_t { Union = 0, Intersection = 1, Difference = 2 };

class Shape;

// reference: src/raytracer/shapes/shape.rs:37-54
struct Properties {
    Matrix4 transform = Matrix4::identity();
    Matrix4 inverse_transform = Matrix4::identity();
    bool transform_set = false;  // identity default has no cofactor-made -0.0 entries
    Material material;
    Bounds bounds;
};

// -------------------------------------------------------------- Primitive ---
// reference: src/raytracer/shapes/primitive.rs:17-108
class Primitive {
public:
    enum class Kind : uint32_t {  // numbering == RT_SPHERE.. of rt_hip.h
        Sphere = 0, Plane = 1, Cube = 2, Cylinder = 3, Cone = 4, Triangle = 5, SmoothTriangle = 6
    };
    static Primitive new_sphere();
    static Primitive new_glass_sphere();
    static Primitive new_plane();
    static Primitive new_cube();
    static Primitive new_cylinder(const Conic& c);
    static Primitive new_cone(const Conic& c);
    static Primitive new_triangle(const Point& p1, const Point& p2, const Point& p3);
    static Primitive new_smooth_triangle(const Point& p1, const Point& p2, const Point& p3,
                                         const Vector& n1, const Vector& n2, const Vector& n3);

    Primitive with_transform(const Matrix4& t) const;
    Primitive with_material(const Material& m) const;
    Primitive with_shadow(bool s) const;
    Shape to_shape() const;

    Kind kind() const { return kind_; }
    const Properties& properties() const { return props_; }
    bool has_shadow() const { return has_shadow_; }
    const Conic& conic() const { return conic_; }
    // triangle data: p1,p2,p3,e1,e2 (primitive.rs:395-409) and normals
    Point p1, p2, p3;
    Vector e1, e2;
    Vector n1, n2, n3;  // flat triangle: n1 = cached face normal (primitive.rs:80-85)

    bool structural_eq(const Primitive& o) const;  // derived PartialEq, primitive.rs:17

private:
    explicit Primitive(Kind k);
    Kind kind_;
    Properties props_;
    bool has_shadow_ = true;
    Conic conic_;
};

// --------------------------------------------------------------- Compound ---
// reference: src/raytracer/shapes/compound.rs:10-73
class Compound {
public:
    enum class Kind : uint32_t { Group = 7, Csg = 8 };
    static Compound new_group(const std::vector<Shape>& children);
    static Compound new_group(std::vector<Shape>&& children);  // takes the vector over (large OBJ groups)
    static Compound new_csg(Operation op, const Shape& left, const Shape& right);

    Compound with_transform(const Matrix4& t) const;
    Compound with_material(const Material& m) const;  // accepted and ignored by rendering (Q8)
    Shape to_shape() const;

    Kind kind() const { return kind_; }
    Operation operation() const { return op_; }
    const Properties& properties() const { return props_; }
    const std::vector<Shape>& children() const { return *children_; }  // csg: [left, right]
    bool includes(const Primitive& other) const;  // compound.rs:64-69

private:
    Compound() = default;
    Kind kind_ = Kind::Group;
    Operation op_ = Operation::Union;
    Properties props_;
    std::shared_ptr<const std::vector<Shape>> children_;
};

// reference: src/raytracer/shapes/shape.rs:10-24
class Shape {
public:
    Shape(const Primitive& p) : v_(p) {}
    Shape(const Compound& c) : v_(c) {}
    bool is_primitive() const { return std::holds_alternative<Primitive>(v_); }
    const Primitive& primitive() const { return std::get<Primitive>(v_); }
    const Compound& compound() const { return std::get<Compound>(v_); }
    const Properties& properties() const {
        return is_primitive() ? primitive().properties() : compound().properties();
    }
    bool includes(const Primitive& other) const;  // shape.rs:18-23

private:
    std::variant<Primitive, Compound> v_;
};

// ------------------------------------------------------------------ Canvas ---
// reference: src/graphics/canvas.rs:7-90
class Canvas {
public:
    Canvas(size_t width, size_t height, std::vector<Color> pixels);
    static Canvas empty(size_t width, size_t height);
    void write_pixel(size_t x, size_t y, const Color& c) { pixels_[y * width_ + x] = c; }
    Color pixel_at(size_t x, size_t y) const { return pixels_[y * width_ + x]; }
    size_t width() const { return width_; }
    size_t height() const { return height_; }
    const std::vector<Color>& pixels() const { return pixels_; }
    std::string to_ppm() const;                 // canvas.rs:46-83
    // the same text from samples already quantised by scale_to_ppm_data (e.g. on the device): rgb8[height*width*3]
    static std::string ppm_from_rgb8(size_t width, size_t height, const uint8_t* rgb8);
    // ... written into out[cap] when it fits; returns the length of the text either way
    static size_t ppm_from_rgb8_into(size_t width, size_t height, const uint8_t* rgb8, char* out, size_t cap);
    void write_ppm(const std::string& path) const;  // canvas.rs:40-44
    static uint8_t scale_to_ppm_data(double c);  // canvas.rs:85-90

private:
    size_t width_, height_;
    std::vector<Color> pixels_;
};

struct Ray {  // reference: src/raytracer/ray.rs:5-8
    Point origin;
    Vector direction;
};

// The reference's hand-rolled pixel loops (exercises/shading.rs:37-60) call the shape API per ray instead of
// Camera::render:  xs = shape.intersect(&r, ..); hit = xs.hit(None); point = r.position(hit.t);
// normal = shape.normal_at(&point, ..); colour = hit.shape.lighting(&point, &light, &-r.direction, &normal, false, ..)
// (shape.rs:28, intersection.rs:154, ray.rs:15, primitive.rs:153, primitive.rs:111).  The same, for a batch of
// rays, on the GPU (rt_hip_shade_rays): colors[i] is meaningful where hit[i] != 0 (BLACK elsewhere).
struct ShadedRays {
    std::vector<Color> colors;
    std::vector<uint8_t> hit;
};
ShadedRays shade_rays(const Shape& shape, const PointLight& light, const std::vector<Ray>& rays);

// ------------------------------------------------------------------- World ---
// reference: src/raytracer/world.rs:17-43,130-145
class World {
public:
    World(std::vector<Shape> shapes, std::vector<PointLight> lights)
        : shapes_(std::move(shapes)), lights_(std::move(lights)) {}
    static World empty() { return World({}, {}); }
    static World default_world();

    // world.rs:32 -- traced on the GPU (one-ray batch through rt_hip_color_at).
    Color color_at(const Ray& ray, uint8_t remaining_recursions) const;
    // batched form of the same call
    std::vector<Color> color_at(const std::vector<Ray>& rays, uint8_t remaining_recursions) const;

    const std::vector<Shape>& shapes() const { return shapes_; }
    const std::vector<PointLight>& lights() const { return lights_; }
    std::vector<Shape>& shapes_mut() { return shapes_; }
    std::vector<PointLight>& lights_mut() { return lights_; }

private:
    std::vector<Shape> shapes_;
    std::vector<PointLight> lights_;
};

// ------------------------------------------------------------------ Camera ---
// reference: src/raytracer/camera.rs:13-66
class Camera {
public:
    Camera(size_t h_size, size_t v_size, double field_of_view);
    Camera with_transform(const Matrix4& t) const;
    Canvas render(const World& world) const;  // camera.rs:51 -- runs on the GPU
    // `camera.render(&world).to_ppm()` (camera.rs:51 + canvas.rs:46) with the quantisation done on the device:
    // byte-identical text, an eighth of the device-to-host traffic, no f64 canvas on the host
    std::string render_ppm(const World& world) const;
    void render_to_file(const World& world, const std::string& path) const;  // ... .write_ppm(path) (canvas.rs:40)

    size_t h_size() const { return h_size_; }
    size_t v_size() const { return v_size_; }
    double field_of_view() const { return fov_; }
    double half_width() const { return half_width_; }
    double half_height() const { return half_height_; }
    double pixel_size() const { return pixel_size_; }
    const Matrix4& transform() const { return transform_; }
    const Matrix4& inverse_transform() const { return inverse_; }
    bool transform_was_set() const { return transform_set_; }
    static constexpr uint8_t kDepth = 5;  // camera.rs:59

private:
    size_t h_size_, v_size_;
    double fov_;
    Matrix4 transform_ = Matrix4::identity();
    Matrix4 inverse_ = Matrix4::identity();
    bool transform_set_ = false;
    double half_width_, half_height_, pixel_size_;
};

namespace obj {
// reference: src/io/obj.rs:10-26
struct ParseError : std::runtime_error {
    enum class Kind { Logic, Syntax, Io } kind;
    ParseError(Kind k, const std::string& s) : std::runtime_error(s), kind(k) {}
};
Shape parse_file(const std::string& path);    // obj.rs:55-60
Shape parse_string(const std::string& text, size_t* ignored_lines = nullptr);  // obj.rs:62-112
}  // namespace obj

// Serialise a world as a flat array of doubles describing the object tree
// exactly as the user built it (transforms, not inverses; no bounds, no derived
// data).  Used only to hand the same scene to an independent checker.
std::vector<double> describe_world(const World& w);

}  // namespace rt
