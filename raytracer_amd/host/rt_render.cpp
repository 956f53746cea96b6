// The two calls of the reference API that trace rays -- Camera::render
// (camera.rs:51) and World::color_at (world.rs:32) -- implemented as
// flatten + one call through the C ABI of include/rt_hip.h.  There is no CPU
// path here: if the HIP library reports an error (no device included) the call
// throws rt::RenderError.
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>

#include "rt_flatten.h"
#include "rt_scene.h"

namespace rt {
namespace {
[[noreturn]] void fail(int code, const char* where) {
    std::string msg = std::string(where) + ": " + rt_hip_strerror(code);
    const char* hip = rt_hip_last_hip_error();
    if (hip && *hip) msg += std::string(" (") + hip + ")";
    throw RenderError(code, msg);
}
}  // namespace

Canvas Camera::render(const World& world) const {
    const FlatScene flat = flatten(world);
    const RtSceneDesc desc = flat.desc();
    const RtCameraDesc cam = camera_desc(*this);
    static_assert(sizeof(Color) == 3 * sizeof(double), "Color must be three packed doubles");
    std::vector<Color> pixels(h_size_ * v_size_);
    const int rc = rt_hip_render(&desc, &cam, kDepth, reinterpret_cast<double*>(pixels.data()), nullptr);
    if (rc != RT_OK) fail(rc, "Camera::render");
    return Canvas(h_size_, v_size_, std::move(pixels));
}

std::string Camera::render_ppm(const World& world) const {
    const FlatScene flat = flatten(world);
    const RtSceneDesc desc = flat.desc();
    const RtCameraDesc cam = camera_desc(*this);
    std::vector<uint8_t> rgb8(h_size_ * v_size_ * 3);
    const int rc = rt_hip_render_rgb8(&desc, &cam, kDepth, rgb8.data(), nullptr);
    if (rc != RT_OK) fail(rc, "Camera::render_ppm");
    return Canvas::ppm_from_rgb8(h_size_, v_size_, rgb8.data());
}

void Camera::render_to_file(const World& world, const std::string& path) const {
    const FlatScene flat = flatten(world);
    const RtSceneDesc desc = flat.desc();
    const RtCameraDesc cam = camera_desc(*this);
    std::vector<uint8_t> rgb8(h_size_ * v_size_ * 3);
    const int rc = rt_hip_render_rgb8(&desc, &cam, kDepth, rgb8.data(), nullptr);
    if (rc != RT_OK) fail(rc, "Camera::render");
    const size_t cap = 64 + rgb8.size() * 4;  // header + at most "255 " per sample
    std::unique_ptr<char[]> text(new char[cap]);  // not zero-filled: the writer's threads touch the pages first
    const size_t n = Canvas::ppm_from_rgb8_into(h_size_, v_size_, rgb8.data(), text.get(), cap);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("file should be written successfully: " + path);
    f.write(text.get(), static_cast<std::streamsize>(n));
    if (!f) throw std::runtime_error("file should be written successfully: " + path);
}

std::vector<Color> World::color_at(const std::vector<Ray>& rays, uint8_t remaining_recursions) const {
    const FlatScene flat = flatten(*this);
    const RtSceneDesc desc = flat.desc();
    RtScene* scene = nullptr;
    int rc = rt_hip_scene_create(&desc, &scene);
    if (rc != RT_OK) fail(rc, "World::color_at");
    static_assert(sizeof(Ray) == 6 * sizeof(double), "Ray must be six packed doubles");
    std::vector<Color> out(rays.size());
    rc = rt_hip_color_at(scene, reinterpret_cast<const double*>(rays.data()), static_cast<uint32_t>(rays.size()),
                         remaining_recursions, reinterpret_cast<double*>(out.data()));
    rt_hip_scene_destroy(scene);
    if (rc != RT_OK) fail(rc, "World::color_at");
    return out;
}

ShadedRays shade_rays(const Shape& shape, const PointLight& light, const std::vector<Ray>& rays) {
    // a world of just this shape: World::intersections (world.rs:45-53) is then shape.intersect with an empty trail
    const FlatScene flat = flatten(World({shape}, {}));
    const RtSceneDesc desc = flat.desc();
    RtScene* scene = nullptr;
    int rc = rt_hip_scene_create(&desc, &scene);
    if (rc != RT_OK) fail(rc, "shade_rays");
    RtLight l;
    l.position[0] = light.position.x, l.position[1] = light.position.y, l.position[2] = light.position.z;
    l.intensity[0] = light.intensity.red, l.intensity[1] = light.intensity.green, l.intensity[2] = light.intensity.blue;
    ShadedRays out;
    out.colors.resize(rays.size());
    out.hit.resize(rays.size());
    rc = rt_hip_shade_rays(scene, &l, reinterpret_cast<const double*>(rays.data()), static_cast<uint32_t>(rays.size()),
                           reinterpret_cast<double*>(out.colors.data()), out.hit.data());
    rt_hip_scene_destroy(scene);
    if (rc != RT_OK) fail(rc, "shade_rays");
    return out;
}

Color World::color_at(const Ray& ray, uint8_t remaining_recursions) const {
    return color_at(std::vector<Ray>{ray}, remaining_recursions)[0];
}

}  // namespace rt
