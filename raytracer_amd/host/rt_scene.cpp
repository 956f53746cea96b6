// Scene-construction half of the host mirror (see rt_scene.h).  Everything in
// this file runs once per scene on the host; none of it is on the per-pixel path.
#include "rt_scene.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <sstream>
#include <string_view>
#include <thread>
#include <unordered_map>

namespace rt {

// ------------------------------------------------------------------ Pattern ---
Pattern Pattern::new_solid(const Color& c) {
    Pattern p;
    p.kind_ = Kind::Solid;
    p.color_ = c;
    return p;
}
Pattern Pattern::new_perturb(const Pattern& child) {
    Pattern p;
    p.kind_ = Kind::Perturb;
    p.a_ = std::make_shared<const Pattern>(child);
    return p;
}
Pattern Pattern::new_test() {
    Pattern p;
    p.kind_ = Kind::Test;
    return p;
}
Pattern Pattern::mixture(Kind k, const Pattern& a, const Pattern& b) {
    Pattern p;
    p.kind_ = k;
    p.a_ = std::make_shared<const Pattern>(a);
    p.b_ = std::make_shared<const Pattern>(b);
    return p;
}
Pattern Pattern::with_transform(const Matrix4& t) const {
    Pattern p = *this;
    p.transform_ = t;
    p.inverse_ = t.inverse();
    p.transform_set_ = true;
    return p;
}
bool Pattern::structural_eq(const Pattern& o) const {
    if (!transform_.approx_eq(o.transform_) || !inverse_.approx_eq(o.inverse_)) return false;
    if (kind_ != o.kind_) return false;
    switch (kind_) {
        case Kind::Solid: return approx_eq(color_, o.color_);
        case Kind::Test: return true;
        case Kind::Perturb: return a_->structural_eq(*o.a_);
        default: return a_->structural_eq(*o.a_) && b_->structural_eq(*o.b_);
    }
}

// ----------------------------------------------------------------- Material ---
bool Material::structural_eq(const Material& o) const {  // material.rs:47-59
    if (!approx_eq(color, o.color)) return false;
    if (!abs_diff_eq(ambient, o.ambient, 1e-5) || !abs_diff_eq(diffuse, o.diffuse, 1e-5) ||
        !abs_diff_eq(specular, o.specular, 1e-5) || !abs_diff_eq(shininess, o.shininess, 1e-5) ||
        !abs_diff_eq(reflective, o.reflective, 1e-5) ||
        !abs_diff_eq(transparency, o.transparency, 1e-5) ||
        !abs_diff_eq(refractive_index, o.refractive_index, 1e-5))
        return false;
    if (pattern.has_value() != o.pattern.has_value()) return false;
    return !pattern.has_value() || pattern->structural_eq(*o.pattern);
}

// ------------------------------------------------------------------- Bounds ---
void Bounds::add_point(const Point& p) {
    minimum.x = std::fmin(minimum.x, p.x);
    minimum.y = std::fmin(minimum.y, p.y);
    minimum.z = std::fmin(minimum.z, p.z);
    maximum.x = std::fmax(maximum.x, p.x);
    maximum.y = std::fmax(maximum.y, p.y);
    maximum.z = std::fmax(maximum.z, p.z);
}
void Bounds::union_with(const Bounds& o) {
    add_point(o.minimum);
    add_point(o.maximum);
}
void Bounds::transform(const Matrix4& t) {
    // corner order of bounds.rs:20-29, captured before the box starts growing
    const Point c[8] = {minimum,
                        Point(minimum.x, minimum.y, maximum.z),
                        Point(minimum.x, maximum.y, minimum.z),
                        Point(minimum.x, maximum.y, maximum.z),
                        Point(maximum.x, minimum.y, minimum.z),
                        Point(maximum.x, minimum.y, maximum.z),
                        Point(maximum.x, maximum.y, minimum.z),
                        maximum};
    for (const Point& corner : c) add_point(t * corner);
}
bool Bounds::structural_eq(const Bounds& o) const {
    return approx_eq(minimum, o.minimum) && approx_eq(maximum, o.maximum);
}

// ---------------------------------------------------------------- Primitive ---
Primitive::Primitive(Kind k) : kind_(k) {}

static Bounds unit_box() { return Bounds(Point(-1.0, -1.0, -1.0), Point(1.0, 1.0, 1.0)); }

Primitive Primitive::new_sphere() {
    Primitive p(Kind::Sphere);
    p.props_.bounds = unit_box();  // primitive.rs:38-40
    return p;
}
Primitive Primitive::new_glass_sphere() {  // primitive.rs:56-62
    Material m;
    m.transparency = 1.0;
    m.refractive_index = 1.5;
    return new_sphere().with_material(m);
}
Primitive Primitive::new_plane() {
    Primitive p(Kind::Plane);
    const double inf = std::numeric_limits<double>::infinity();
    p.props_.bounds = Bounds(Point(-inf, 0.0, -inf), Point(inf, 0.0, inf));  // primitive.rs:41-44
    return p;
}
Primitive Primitive::new_cube() {
    Primitive p(Kind::Cube);
    p.props_.bounds = unit_box();
    return p;
}
Primitive Primitive::new_cylinder(const Conic& c) {
    Primitive p(Kind::Cylinder);
    p.conic_ = c;
    p.props_.bounds = Bounds(Point(-1.0, c.minimum, -1.0), Point(1.0, c.maximum, 1.0));  // :535-540
    return p;
}
Primitive Primitive::new_cone(const Conic& c) {
    Primitive p(Kind::Cone);
    p.conic_ = c;
    p.props_.bounds = Bounds(Point(-1.0, c.minimum, -1.0), Point(1.0, c.maximum, 1.0));
    return p;
}
static void set_triangular(Primitive& p, const Point& p1, const Point& p2, const Point& p3, Bounds& b) {
    p.p1 = p1, p.p2 = p2, p.p3 = p3;
    p.e1 = p2 - p1;  // primitive.rs:404-409
    p.e2 = p3 - p1;
    b = Bounds();
    b.add_point(p1);  // primitive.rs:442-450
    b.add_point(p2);
    b.add_point(p3);
}
Primitive Primitive::new_triangle(const Point& p1, const Point& p2, const Point& p3) {
    Primitive p(Kind::Triangle);
    set_triangular(p, p1, p2, p3, p.props_.bounds);
    p.n1 = p.e1.cross(p.e2).normalize();  // primitive.rs:82
    return p;
}
Primitive Primitive::new_smooth_triangle(const Point& p1, const Point& p2, const Point& p3,
                                         const Vector& n1, const Vector& n2, const Vector& n3) {
    Primitive p(Kind::SmoothTriangle);
    set_triangular(p, p1, p2, p3, p.props_.bounds);
    p.n1 = n1, p.n2 = n2, p.n3 = n3;
    return p;
}
Primitive Primitive::with_transform(const Matrix4& t) const {  // shape.rs:63-67
    Primitive p = *this;
    p.props_.transform = t;
    p.props_.inverse_transform = t.inverse();
    p.props_.transform_set = true;
    return p;
}
Primitive Primitive::with_material(const Material& m) const {
    Primitive p = *this;
    p.props_.material = m;
    return p;
}
Primitive Primitive::with_shadow(bool s) const {
    Primitive p = *this;
    p.has_shadow_ = s;
    return p;
}
Shape Primitive::to_shape() const { return Shape(*this); }

static bool conic_eq(const Conic& a, const Conic& b) {  // primitive.rs:553-571, eps 1e-5
    return abs_diff_eq(a.minimum, b.minimum, 1e-5) && abs_diff_eq(a.maximum, b.maximum, 1e-5) &&
           a.closed == b.closed;
}
bool Primitive::structural_eq(const Primitive& o) const {
    const Properties &a = props_, &b = o.props_;
    if (!a.transform.approx_eq(b.transform) || !a.inverse_transform.approx_eq(b.inverse_transform))
        return false;
    if (!a.material.structural_eq(b.material)) return false;
    if (!a.bounds.structural_eq(b.bounds)) return false;  // false for any infinite box (Q7)
    if (has_shadow_ != o.has_shadow_ || kind_ != o.kind_) return false;
    switch (kind_) {
        case Kind::Cylinder:
        case Kind::Cone: return conic_eq(conic_, o.conic_);
        case Kind::Triangle:
            return approx_eq(p1, o.p1) && approx_eq(p2, o.p2) && approx_eq(p3, o.p3) &&
                   approx_eq(e1, o.e1) && approx_eq(e2, o.e2) && approx_eq(n1, o.n1);
        case Kind::SmoothTriangle:
            return approx_eq(p1, o.p1) && approx_eq(p2, o.p2) && approx_eq(p3, o.p3) &&
                   approx_eq(e1, o.e1) && approx_eq(e2, o.e2) && approx_eq(n1, o.n1) &&
                   approx_eq(n2, o.n2) && approx_eq(n3, o.n3);
        default: return true;
    }
}

// ----------------------------------------------------------------- Compound ---
Compound Compound::new_group(const std::vector<Shape>& children) { return new_group(std::vector<Shape>(children)); }
Compound Compound::new_group(std::vector<Shape>&& children) {
    Compound c;
    c.kind_ = Kind::Group;
    c.children_ = std::make_shared<const std::vector<Shape>>(std::move(children));
    Bounds b;  // compound.rs:29-36: union with the child's box, then grow by the child's transform
    for (const Shape& child : *c.children_) {
        b.union_with(child.properties().bounds);
        b.transform(child.properties().transform);
    }
    c.props_.bounds = b;
    return c;
}
Compound Compound::new_csg(Operation op, const Shape& left, const Shape& right) {
    Compound c;
    c.kind_ = Kind::Csg;
    c.op_ = op;
    c.children_ = std::make_shared<const std::vector<Shape>>(std::vector<Shape>{left, right});
    Bounds b;  // compound.rs:38-47
    b.union_with(left.properties().bounds);
    b.transform(left.properties().transform);
    b.union_with(right.properties().bounds);
    b.transform(right.properties().transform);
    c.props_.bounds = b;
    return c;
}
Compound Compound::with_transform(const Matrix4& t) const {
    Compound c = *this;
    c.props_.transform = t;
    c.props_.inverse_transform = t.inverse();
    c.props_.transform_set = true;
    return c;
}
Compound Compound::with_material(const Material& m) const {
    Compound c = *this;
    c.props_.material = m;
    return c;
}
Shape Compound::to_shape() const { return Shape(*this); }
bool Compound::includes(const Primitive& other) const {
    for (const Shape& s : *children_)
        if (s.includes(other)) return true;
    return false;
}
bool Shape::includes(const Primitive& other) const {
    return is_primitive() ? primitive().structural_eq(other) : compound().includes(other);
}

// ------------------------------------------------------------------- Canvas ---
Canvas::Canvas(size_t width, size_t height, std::vector<Color> pixels)
    : width_(width), height_(height), pixels_(std::move(pixels)) {}
Canvas Canvas::empty(size_t width, size_t height) {
    return Canvas(width, height, std::vector<Color>(width * height, Color::BLACK()));
}
uint8_t Canvas::scale_to_ppm_data(double c) {
    double v = c * 255.0;
    if (std::isnan(v)) return 0;  // Rust `as u8` maps NaN to 0
    if (v < 0.0) v = 0.0;
    if (v > 255.0) v = 255.0;
    return static_cast<uint8_t>(std::round(v));  // half away from zero, like f64::round
}
namespace {
// decimal text of 0..255, one trailing separator slot: the formatter copies 4 bytes per token
struct ByteText {
    char text[256][4];
    uint8_t len[256];
    ByteText() {
        for (int v = 0; v < 256; ++v) {
            const int n = std::snprintf(text[v], 4, "%d", v);
            len[v] = static_cast<uint8_t>(n);
            for (int k = n; k < 4; ++k) text[v][k] = ' ';
        }
    }
};
const ByteText& byte_text() {
    static const ByteText t;
    return t;
}
// canvas.rs:68-79 for one image row: a token joins the current line while the running sum of
// (token length + 1) stays <= 70; the sum restarts with each line.  Returns the end of the output;
// with dst == nullptr only counts.
char* format_row(const uint8_t* row, size_t n_tokens, char* dst, size_t* count) {
    const ByteText& bt = byte_text();
    size_t line_len = 0, total = 0;
    for (size_t i = 0; i < n_tokens; ++i) {
        const uint8_t v = row[i];
        const size_t n = bt.len[v];
        char sep = ' ';
        if (line_len + n + 1 > 70 && line_len != 0) {
            sep = '\n';
            line_len = 0;
        }
        if (dst) {
            if (i != 0) *dst++ = sep;
            // 4-byte copy: the slack is overwritten by the next token; the last token of a row must not
            // spill into the next row's bytes (another thread may own them)
            std::memcpy(dst, bt.text[v], i + 1 < n_tokens ? 4 : n);
            dst += n;
        }
        total += n + (i != 0 ? 1 : 0);
        line_len += n + 1;
    }
    if (n_tokens > 0) {
        if (dst) *dst++ = '\n';
        ++total;
    }
    if (count) *count = total;
    return dst;
}
}  // namespace

namespace {
// lays the text out (pass 1), asks `place(total)` where it goes (nullptr: nowhere) and writes it (pass 2)
template <class Place>
size_t ppm_build(size_t width, size_t height, const uint8_t* rgb8, Place&& place) {
    const std::string header = "P3\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
    const size_t n_tokens = width * 3;
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads == 0) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    if (height * n_tokens < (1u << 20)) n_threads = 1;
    auto for_rows = [&](auto&& fn) {  // contiguous row ranges: each thread touches its own pages of the output
        if (n_threads == 1) {
            for (size_t y = 0; y < height; ++y) fn(y);
            return;
        }
        std::vector<std::thread> pool;
        const size_t chunk = (height + n_threads - 1) / n_threads;
        for (unsigned t = 0; t < n_threads; ++t)
            pool.emplace_back([&, t] {
                const size_t y1 = std::min(height, (t + 1) * chunk);
                for (size_t y = t * chunk; y < y1; ++y) fn(y);
            });
        for (std::thread& th : pool) th.join();
    };
    // pass 1: the byte count of every row (rows are independent: the line budget restarts per row)
    std::vector<size_t> offset(height + 1, 0);
    for_rows([&](size_t y) { format_row(rgb8 + y * n_tokens, n_tokens, nullptr, &offset[y + 1]); });
    for (size_t y = 0; y < height; ++y) offset[y + 1] += offset[y];
    const size_t total = header.size() + offset[height];
    char* out = place(total);
    if (!out) return total;
    // pass 2: every row straight into its final position
    std::memcpy(out, header.data(), header.size());
    char* body = out + header.size();
    for_rows([&](size_t y) { format_row(rgb8 + y * n_tokens, n_tokens, body + offset[y], nullptr); });
    return total;
}
}  // namespace

size_t Canvas::ppm_from_rgb8_into(size_t width, size_t height, const uint8_t* rgb8, char* out, size_t cap) {
    return ppm_build(width, height, rgb8, [&](size_t total) { return out && total <= cap ? out : nullptr; });
}
std::string Canvas::ppm_from_rgb8(size_t width, size_t height, const uint8_t* rgb8) {
    std::string out;
    ppm_build(width, height, rgb8, [&](size_t total) {
        out.resize(total);
        return &out[0];
    });
    return out;
}
std::string Canvas::to_ppm() const {
    std::vector<uint8_t> rgb8(pixels_.size() * 3);
    for (size_t i = 0; i < pixels_.size(); ++i) {
        rgb8[i * 3 + 0] = scale_to_ppm_data(pixels_[i].red);
        rgb8[i * 3 + 1] = scale_to_ppm_data(pixels_[i].green);
        rgb8[i * 3 + 2] = scale_to_ppm_data(pixels_[i].blue);
    }
    return ppm_from_rgb8(width_, height_, rgb8.data());
}
void Canvas::write_ppm(const std::string& path) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("file should be written successfully: " + path);
    const std::string s = to_ppm();
    f.write(s.data(), static_cast<std::streamsize>(s.size()));
    if (!f) throw std::runtime_error("file should be written successfully: " + path);
}

// ------------------------------------------------------------------- World ---
World World::default_world() {  // world.rs:130-145
    Material m1;
    m1.color = Color(0.8, 1.0, 0.6);
    m1.diffuse = 0.7;
    m1.specular = 0.2;
    Primitive s1 = Primitive::new_sphere().with_material(m1);
    Primitive s2 = Primitive::new_sphere().with_transform(Matrix4::scaling(0.5, 0.5, 0.5));
    return World({s1.to_shape(), s2.to_shape()},
                 {PointLight(Point(-10.0, 10.0, -10.0), Color::WHITE())});
}

// ------------------------------------------------------------------ Camera ---
Camera::Camera(size_t h_size, size_t v_size, double field_of_view)
    : h_size_(h_size), v_size_(v_size), fov_(field_of_view) {
    const double half_view = std::tan(field_of_view / 2.0);  // camera.rs:25
    const double aspect = static_cast<double>(h_size) / static_cast<double>(v_size);
    if (aspect >= 1.0) {
        half_width_ = half_view;
        half_height_ = half_view / aspect;
    } else {
        half_width_ = half_view * aspect;
        half_height_ = half_view;
    }
    pixel_size_ = (half_width_ * 2.0) / static_cast<double>(h_size);  // camera.rs:41
}
Camera Camera::with_transform(const Matrix4& t) const {
    Camera c = *this;
    c.transform_ = t;
    c.inverse_ = t.inverse();
    c.transform_set_ = true;
    return c;
}

// --------------------------------------------------------------------- OBJ ---
namespace obj {
namespace {
// Tokens are views into the text: the teapot is 45 000 lines, and a std::string per token was most of
// the parse time.
using Tok = std::string_view;

// Rust's str::parse::<f64>: decimal float syntax, whole token, no whitespace.  The token is followed in
// the text by a separator or the terminating NUL, where strtod stops by itself.
bool parse_f64(Tok s, double& out) {
    if (s.empty() || std::isspace(static_cast<unsigned char>(s[0]))) return false;
    if (s.find_first_of("xX") != Tok::npos) return false;  // strtod would take hex floats
    char* end = nullptr;
    errno = 0;
    out = std::strtod(s.data(), &end);
    return end == s.data() + s.size();
}
bool parse_usize(Tok s, size_t& out) {
    if (s.empty()) return false;
    size_t i = (s[0] == '+') ? 1 : 0;  // usize::from_str accepts a leading '+'
    if (i == s.size()) return false;
    size_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + static_cast<size_t>(s[i] - '0');
    }
    out = v;
    return true;
}
// str::split(sep): every separator ends a (possibly empty) part; `parts` is reused between lines
void split(Tok s, char sep, std::vector<Tok>& parts) {
    parts.clear();
    size_t start = 0;
    while (true) {
        const size_t pos = s.find(sep, start);
        if (pos == Tok::npos) {
            parts.push_back(s.substr(start));
            break;
        }
        parts.push_back(s.substr(start, pos - start));
        start = pos + 1;
    }
}
Tok trim(Tok s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace(static_cast<unsigned char>(s[b]))) ++b;
    while (e > b && std::isspace(static_cast<unsigned char>(s[e - 1]))) --e;
    return s.substr(b, e - b);
}
struct VertexRef {
    Point p;
    bool has_n = false;
    Vector n;
};
VertexRef parse_vertex_ref(Tok v, const std::vector<Point>& vertices, const std::vector<Vector>& normals) {  // obj.rs:164-185
    Tok parts[3];
    size_t n_parts = 0, start = 0;
    while (true) {  // split('/')
        const size_t pos = v.find('/', start);
        const Tok part = pos == Tok::npos ? v.substr(start) : v.substr(start, pos - start);
        if (n_parts < 3) parts[n_parts] = part;
        ++n_parts;
        if (pos == Tok::npos) break;
        start = pos + 1;
    }
    const auto bad = [&]() {
        return ParseError(ParseError::Kind::Syntax,
                          "Syntax error, indices must be natural numbers, given: " + std::string(v));
    };
    VertexRef r;
    size_t vi = 0, ni = 0;
    if (n_parts == 3) {
        if (!parse_usize(parts[0], vi) || !parse_usize(parts[2], ni)) throw bad();
        r.p = vertices.at(vi - 1);  // out of range panics in the reference
        r.n = normals.at(ni - 1);
        r.has_n = true;
    } else if (n_parts == 1 || n_parts == 2) {
        if (!parse_usize(parts[0], vi)) throw bad();
        r.p = vertices.at(vi - 1);
    } else {
        throw ParseError(ParseError::Kind::Syntax, "Syntax error, invalid vertex ref: " + std::string(v));
    }
    return r;
}
Primitive parse_triangle(Tok a, Tok b, Tok c, const std::vector<Point>& vertices, const std::vector<Vector>& normals) {
    const VertexRef r1 = parse_vertex_ref(a, vertices, normals);
    const VertexRef r2 = parse_vertex_ref(b, vertices, normals);
    const VertexRef r3 = parse_vertex_ref(c, vertices, normals);
    if (r1.has_n && r2.has_n && r3.has_n)
        return Primitive::new_smooth_triangle(r1.p, r2.p, r3.p, r1.n, r2.n, r3.n);
    if (!r1.has_n && !r2.has_n && !r3.has_n) return Primitive::new_triangle(r1.p, r2.p, r3.p);
    throw ParseError(ParseError::Kind::Syntax, "Syntax error, invalid triangle: " + std::string(a) + " " +
                                                   std::string(b) + " " + std::string(c));
}
}  // namespace

Shape parse_string(const std::string& text, size_t* ignored_lines) {
    std::vector<Shape> shapes;
    std::vector<Point> vertices;
    std::vector<Vector> normals;
    // The reference keeps named groups in a HashMap (obj.rs:36) whose iteration
    // order is unspecified; first-seen order is used here so results are repeatable.
    std::vector<std::string> group_order;
    std::unordered_map<std::string, std::vector<Primitive>> groups;
    bool have_group = false;
    std::string current_group;
    std::vector<Primitive>* current = nullptr;  // the open group's triangles, once it has any
    size_t ignored = 0;

    const Tok all(text);
    std::vector<Tok> tok;
    size_t at = 0;
    while (at < all.size()) {  // str::lines: split at \n, a trailing \r goes with it, no empty last line
        const size_t nl = all.find('\n', at);
        Tok line = nl == Tok::npos ? all.substr(at) : all.substr(at, nl - at);
        at = nl == Tok::npos ? all.size() : nl + 1;
        if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
        split(trim(line), ' ', tok);  // obj.rs:76: single-space split
        if (tok.size() == 4 && (tok[0] == "v" || tok[0] == "vn")) {
            double x, y, z;
            if (!parse_f64(tok[1], x) || !parse_f64(tok[2], y) || !parse_f64(tok[3], z))
                throw ParseError(ParseError::Kind::Syntax,
                                 std::string("Syntax error, invalid ") + (tok[0] == "v" ? "vertex" : "normal") + ": " +
                                     std::string(tok[1]) + " " + std::string(tok[2]) + " " + std::string(tok[3]));
            if (tok[0] == "v")
                vertices.emplace_back(x, y, z);
            else
                normals.emplace_back(x, y, z);
        } else if (tok.size() >= 4 && tok[0] == "f") {
            for (size_t i = 2; i + 1 < tok.size(); ++i) {  // fan triangulation, obj.rs:132-140
                Primitive t = parse_triangle(tok[1], tok[i], tok[i + 1], vertices, normals);
                if (!have_group) {
                    shapes.push_back(t.to_shape());
                    continue;
                }
                if (!current) {  // a group exists from its first face on
                    auto it = groups.find(current_group);
                    if (it == groups.end()) {
                        group_order.push_back(current_group);
                        it = groups.emplace(current_group, std::vector<Primitive>()).first;
                    }
                    current = &it->second;
                }
                current->push_back(std::move(t));
            }
        } else if (tok.size() == 2 && tok[0] == "g") {
            have_group = true;
            current_group = std::string(tok[1]);
            current = nullptr;
        } else {
            ++ignored;
        }
    }
    if (ignored_lines) *ignored_lines = ignored;

    shapes.reserve(shapes.size() + group_order.size());
    for (const std::string& name : group_order) {  // obj.rs:40-45
        std::vector<Primitive>& members = groups[name];
        std::vector<Shape> kids;
        kids.reserve(members.size());
        for (Primitive& t : members) kids.push_back(t.to_shape());
        shapes.push_back(Compound::new_group(std::move(kids)).to_shape());
    }
    if (shapes.empty()) throw ParseError(ParseError::Kind::Logic, "Logic error, no shapes found");
    if (shapes.size() == 1) return std::move(shapes.back());
    return Compound::new_group(std::move(shapes)).to_shape();
}

Shape parse_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ParseError(ParseError::Kind::Io, "IO error, cannot read " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_string(ss.str());
}
}  // namespace obj

// ---------------------------------------------------------------- describe ---
namespace {
void put_matrix(std::vector<double>& out, const Matrix4& m) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out.push_back(m.m[i][j]);
}
void put_pattern(std::vector<double>& out, const Pattern& p) {
    out.push_back(static_cast<double>(static_cast<uint32_t>(p.kind())));
    out.push_back(p.transform_was_set() ? 1.0 : 0.0);
    put_matrix(out, p.transform());
    switch (p.kind()) {
        case Pattern::Kind::Solid:
            out.push_back(p.color().red), out.push_back(p.color().green), out.push_back(p.color().blue);
            break;
        case Pattern::Kind::Test: break;
        case Pattern::Kind::Perturb: put_pattern(out, *p.a()); break;
        default:
            put_pattern(out, *p.a());
            put_pattern(out, *p.b());
    }
}
void put_material(std::vector<double>& out, const Material& m) {
    out.push_back(m.color.red), out.push_back(m.color.green), out.push_back(m.color.blue);
    out.push_back(m.ambient), out.push_back(m.diffuse), out.push_back(m.specular);
    out.push_back(m.shininess), out.push_back(m.reflective), out.push_back(m.transparency);
    out.push_back(m.refractive_index);
    out.push_back(m.pattern ? 1.0 : 0.0);
    if (m.pattern) put_pattern(out, *m.pattern);
}
void put_point(std::vector<double>& out, double x, double y, double z) {
    out.push_back(x), out.push_back(y), out.push_back(z);
}
void put_shape(std::vector<double>& out, const Shape& s) {
    const Properties& pr = s.properties();
    if (s.is_primitive()) {
        const Primitive& p = s.primitive();
        out.push_back(static_cast<double>(static_cast<uint32_t>(p.kind())));
        out.push_back(p.has_shadow() ? 1.0 : 0.0);
        out.push_back(pr.transform_set ? 1.0 : 0.0);
        put_matrix(out, pr.transform);
        put_material(out, pr.material);
        switch (p.kind()) {
            case Primitive::Kind::Cylinder:
            case Primitive::Kind::Cone:
                out.push_back(p.conic().minimum), out.push_back(p.conic().maximum);
                out.push_back(p.conic().closed ? 1.0 : 0.0);
                break;
            case Primitive::Kind::Triangle:
            case Primitive::Kind::SmoothTriangle:
                put_point(out, p.p1.x, p.p1.y, p.p1.z);
                put_point(out, p.p2.x, p.p2.y, p.p2.z);
                put_point(out, p.p3.x, p.p3.y, p.p3.z);
                if (p.kind() == Primitive::Kind::SmoothTriangle) {
                    put_point(out, p.n1.x, p.n1.y, p.n1.z);
                    put_point(out, p.n2.x, p.n2.y, p.n2.z);
                    put_point(out, p.n3.x, p.n3.y, p.n3.z);
                }
                break;
            default: break;
        }
    } else {
        const Compound& c = s.compound();
        out.push_back(static_cast<double>(static_cast<uint32_t>(c.kind())));
        out.push_back(1.0);
        out.push_back(pr.transform_set ? 1.0 : 0.0);
        put_matrix(out, pr.transform);
        put_material(out, pr.material);
        if (c.kind() == Compound::Kind::Csg) {
            out.push_back(static_cast<double>(static_cast<uint32_t>(c.operation())));
        } else {
            out.push_back(static_cast<double>(c.children().size()));
        }
        for (const Shape& child : c.children()) put_shape(out, child);
    }
}
}  // namespace

std::vector<double> describe_world(const World& w) {
    std::vector<double> out;
    out.push_back(static_cast<double>(w.lights().size()));
    for (const PointLight& l : w.lights()) {
        put_point(out, l.position.x, l.position.y, l.position.z);
        put_point(out, l.intensity.red, l.intensity.green, l.intensity.blue);
    }
    out.push_back(static_cast<double>(w.shapes().size()));
    for (const Shape& s : w.shapes()) put_shape(out, s);
    return out;
}

}  // namespace rt
