// Scene builders: the reference's nine render exercises (exercises/*.rs) and the
// synthetic many-primitive scene of BASELINE.json config 5.
#pragma once
#include <string>
#include <vector>

#include "rt_scene.h"

namespace rt {

struct Exercise {
    World world;
    Camera camera;
};

// Names accepted by build_exercise: plane scene pattern glass_bubble teapot
// hexagon csg cover room synthetic.
const std::vector<std::string>& exercise_names();

// `resource_dir` must hold teapot.obj (reference: exercises/resources/teapot.obj)
// for the "teapot" exercise; it is ignored otherwise.
Exercise build_exercise(const std::string& name, size_t width, size_t height, const std::string& resource_dir);

// exercises/shading.rs:17-63, the exercise that goes round Camera::render: a 500 x 500 canvas of one
// sphere shaded by intersect / hit / normal_at / lighting per pixel (on the GPU: rt::shade_rays).
Canvas shading_exercise();

// Deterministic cover.rs/room.rs-style scene with `n_primitives` mixed primitives
// in two levels of groups (SplitMix64, bit-reproducible).  BASELINE.json config 5
// uses n_primitives = 10000, seed 0x5EED5EED.
Exercise synthetic_scene(size_t width, size_t height, size_t n_primitives, uint64_t seed);

}  // namespace rt
