// rt_exercise -- the reference's render exercises (exercises/<name>.rs) as one executable:
// build the scene through the mirrored public API, `camera.render(&world).write_ppm(path)`
// (e.g. exercises/plane.rs:68-74).  Rendering happens on the GPU through the C ABI of
// include/rt_hip.h; without a GPU the render call throws.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>

#include "../host/rt_exercises.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <exercise> [width height] [out.ppm] [resource_dir]\n  exercises:", argv[0]);
        for (const std::string& n : rt::exercise_names()) std::fprintf(stderr, " %s", n.c_str());
        std::fprintf(stderr, " shading\n");
        return 2;
    }
    const std::string name = argv[1];
    const size_t width = argc > 3 ? std::strtoul(argv[2], nullptr, 10) : 2048;   // exercises/plane.rs:68
    const size_t height = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 1080;
    const std::string out = argc > 4 ? argv[4] : name + ".ppm";
    const std::string resources = argc > 5 ? argv[5] : "tests/golden";
    try {
        if (name == "shading") {  // exercises/shading.rs: its own pixel loop, 500 x 500
            const std::string path = argc > 2 && argc != 4 ? argv[argc - 1] : out;
            rt::shading_exercise().write_ppm(path);
            std::printf("shading 500x500 written -> %s\n", path.c_str());
            return 0;
        }
        rt::Exercise e = rt::build_exercise(name, width, height, resources);
        const auto t0 = std::chrono::steady_clock::now();
        e.camera.render_to_file(e.world, out);  // camera.render(&world).write_ppm(path), quantised on the device
        const auto t1 = std::chrono::steady_clock::now();
        std::printf("%s %zux%zu rendered and written in %.2f ms -> %s\n", name.c_str(), width, height,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), out.c_str());
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "%s: %s\n", name.c_str(), ex.what());
        return 1;
    }
    return 0;
}
