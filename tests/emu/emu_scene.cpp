// tests/emu/emu_scene.cpp -- TEST INFRASTRUCTURE: the scene of an emulator run, built by the host mirror (see emu_main.cpp).
#include <cstdlib>
#include <string>

#include "../../include/rt_hip.h"
#include "../../raytracer_amd/host/rt_exercises.h"
#include "../../raytracer_amd/host/rt_flatten.h"

int emu_build_scene(const char* name_c, size_t width, size_t height, size_t prims, RtSceneDesc* desc, RtCameraDesc* cam) {
    const std::string name = name_c;
    const char* res = std::getenv("RT_EMU_RESOURCES");
    rt::Exercise ex = name == "synthetic" ? rt::synthetic_scene(width, height, prims, 0x5EED5EEDull)
                                          : rt::build_exercise(name, width, height, res ? res : "tests/golden");
    rt::FlattenOptions opt;
    opt.dir_device_min = (size_t)-1;  // the direction cells from the host: rt_dirindex.hip is not part of the emulation
    if (const char* h = std::getenv("RT_EMU_HIERARCHY")) opt.hierarchy = std::atoi(h);
    static rt::FlatScene flat;  // (the descriptor points into it)
    flat = rt::flatten(ex.world, opt);
    *desc = flat.desc();
    *cam = rt::camera_desc(ex.camera);
    return 0;
}
