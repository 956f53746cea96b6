// The reference's render exercises, transcribed against the C++ mirror of its
// API.  Each function builds exactly the scene of the named reference program
// (same construction order, same constants); only the canvas size is a
// parameter (the reference hard-codes 2048x1080, e.g. exercises/plane.rs:68).
#include "rt_exercises.h"

#include <cmath>

namespace rt {
namespace {

constexpr double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI
constexpr double FRAC_PI_2 = 1.57079632679489661923132169163975144;
constexpr double FRAC_PI_3 = 1.04719755119659774615421446109316763;
constexpr double FRAC_PI_4 = 0.785398163397448309615660845819875721;
constexpr double FRAC_PI_6 = 0.52359877559829887307710723054658381;

using M = Matrix4;

Camera standard_camera(size_t w, size_t h) {  // shared by plane/pattern/teapot/hexagon/scene
    return Camera(w, h, FRAC_PI_3)
        .with_transform(M::view_transform(Point(0.0, 1.5, -5.0), Point(0.0, 1.0, 0.0), Vector(0.0, 1.0, 0.0)));
}
PointLight standard_light() { return PointLight(Point(-10.0, 10.0, -10.0), Color::WHITE()); }

Material sphere_material(const Color& c) {
    Material m;
    m.color = c;
    m.diffuse = 0.7;
    m.specular = 0.3;
    return m;
}

// exercises/plane.rs:19-75
Exercise plane(size_t w, size_t h) {
    Material floor_material;
    floor_material.color = Color(1.0, 0.9, 0.9);
    floor_material.specular = 0.0;
    Shape floor = Primitive::new_plane().with_material(floor_material).to_shape();
    Shape middle = Primitive::new_sphere()
                       .with_transform(M::translation(-0.5, 1.0, 0.5))
                       .with_material(sphere_material(Color(0.1, 1.0, 0.5)))
                       .to_shape();
    Shape right = Primitive::new_sphere()
                      .with_transform(M::translation(1.5, 0.5, -0.5) * M::scaling(0.5, 0.5, 0.5))
                      .with_material(sphere_material(Color(0.5, 1.0, 0.1)))
                      .to_shape();
    Shape left = Primitive::new_sphere()
                     .with_transform(M::translation(-1.5, 0.33, -0.75) * M::scaling(0.33, 0.33, 0.33))
                     .with_material(sphere_material(Color(1.0, 0.8, 0.1)))
                     .to_shape();
    return {World({floor, middle, right, left}, {standard_light()}), standard_camera(w, h)};
}

// exercises/scene.rs:19-96
Exercise scene(size_t w, size_t h) {
    Material floor_material;
    floor_material.color = Color(1.0, 0.9, 0.9);
    floor_material.specular = 0.0;
    Shape floor = Primitive::new_sphere()
                      .with_transform(M::scaling(10.0, 0.01, 10.0))
                      .with_material(floor_material)
                      .to_shape();
    Shape left_wall = Primitive::new_sphere()
                          .with_transform(M::translation(0.0, 0.0, 5.0) * M::rotation_y(-FRAC_PI_4) *
                                          M::rotation_x(FRAC_PI_2) * M::scaling(10.0, 0.01, 10.0))
                          .with_material(floor_material)
                          .to_shape();
    Shape right_wall = Primitive::new_sphere()
                           .with_transform(M::translation(0.0, 0.0, 5.0) * M::rotation_y(FRAC_PI_4) *
                                           M::rotation_x(FRAC_PI_2) * M::scaling(10.0, 0.01, 10.0))
                           .with_material(floor_material)
                           .to_shape();
    Shape middle = Primitive::new_sphere()
                       .with_transform(M::translation(-0.5, 1.0, 0.5))
                       .with_material(sphere_material(Color(0.1, 1.0, 0.5)))
                       .to_shape();
    Shape right = Primitive::new_sphere()
                      .with_transform(M::translation(1.5, 0.5, -0.5) * M::scaling(0.5, 0.5, 0.5))
                      .with_material(sphere_material(Color(0.5, 1.0, 0.1)))
                      .to_shape();
    Shape left = Primitive::new_sphere()
                     .with_transform(M::translation(-1.5, 0.33, -0.75) * M::scaling(0.33, 0.33, 0.33))
                     .with_material(sphere_material(Color(1.0, 0.8, 0.1)))
                     .to_shape();
    return {World({floor, left_wall, right_wall, middle, right, left}, {standard_light()}),
            standard_camera(w, h)};
}

// exercises/pattern.rs:19-117
Exercise pattern(size_t w, size_t h) {
    Pattern sub_pattern = Pattern::new_checker(Pattern::new_solid(Color(0.20, 0.20, 0.20)),
                                               Pattern::new_solid(Color(0.55, 0.55, 0.55)))
                              .with_transform(M::scaling(0.25, 0.25, 0.25));
    Pattern floor_pattern = Pattern::new_stripe(sub_pattern, Pattern::new_solid(Color(0.20, 0.20, 0.20)))
                                .with_transform(M::rotation_y(FRAC_PI_3) * M::scaling(0.5, 0.5, 0.5));
    Material floor_material;
    floor_material.specular = 0.0;
    floor_material.pattern = floor_pattern;
    Shape floor = Primitive::new_plane().with_material(floor_material).to_shape();
    Shape left_wall = Primitive::new_plane()
                          .with_transform(M::translation(0.0, 0.0, 5.0) * M::rotation_y(-FRAC_PI_4) *
                                          M::rotation_x(FRAC_PI_2))
                          .with_material(floor_material)
                          .to_shape();
    Shape right_wall = Primitive::new_plane()
                           .with_transform(M::translation(0.0, 0.0, 5.0) * M::rotation_y(FRAC_PI_4) *
                                           M::rotation_x(FRAC_PI_2))
                           .with_material(floor_material)
                           .to_shape();

    Pattern middle_pattern =
        Pattern::new_perturb(Pattern::new_ring(Pattern::new_solid(Color(0.0, 0.3, 0.6)),
                                               Pattern::new_solid(Color(0.1, 1.0, 0.8))))
            .with_transform(M::rotation_x(-FRAC_PI_3) * M::scaling(0.2, 0.2, 0.2));
    Material middle_material = sphere_material(Color::WHITE());
    middle_material.pattern = middle_pattern;
    Shape middle = Primitive::new_sphere()
                       .with_transform(M::translation(-0.5, 1.0, 0.5))
                       .with_material(middle_material)
                       .to_shape();

    Pattern right_pattern =
        Pattern::new_stripe(Pattern::new_solid(Color::BLACK()), Pattern::new_solid(Color::WHITE()))
            .with_transform(M::rotation_z(-FRAC_PI_4) * M::scaling(0.2, 0.2, 0.2));
    Material right_material = sphere_material(Color::WHITE());
    right_material.pattern = right_pattern;
    Shape right = Primitive::new_sphere()
                      .with_transform(M::translation(1.5, 0.5, -0.5) * M::scaling(0.5, 0.5, 0.5))
                      .with_material(right_material)
                      .to_shape();

    Pattern left_pattern =
        Pattern::new_gradient(Pattern::new_solid(Color(1.0, 0.0, 0.0)), Pattern::new_solid(Color(0.0, 1.0, 1.0)))
            .with_transform(M::translation(-1.0, 0.0, 0.0) * M::scaling(2.0, 2.0, 2.0));
    Material left_material = sphere_material(Color::WHITE());
    left_material.pattern = left_pattern;
    Shape left = Primitive::new_sphere()
                     .with_transform(M::translation(-1.5, 0.33, -0.75) * M::scaling(0.33, 0.33, 0.33))
                     .with_material(left_material)
                     .to_shape();

    return {World({floor, left_wall, right_wall, middle, left, right}, {standard_light()}),
            standard_camera(w, h)};
}

// exercises/glass_bubble.rs:19-71
Exercise glass_bubble(size_t w, size_t h) {
    Material floor_material;
    floor_material.specular = 0.0;
    floor_material.pattern =
        Pattern::new_checker(Pattern::new_solid(Color::WHITE()), Pattern::new_solid(Color::BLACK()));
    Shape floor = Primitive::new_plane()
                      .with_material(floor_material)
                      .with_transform(M::translation(0.0, -10.0, 0.0))
                      .to_shape();
    Material glass_material;
    glass_material.diffuse = 0.1;
    glass_material.shininess = 300.0;
    glass_material.reflective = 1.0;
    glass_material.transparency = 1.0;
    glass_material.refractive_index = 1.52;
    Shape glass = Primitive::new_sphere().with_material(glass_material).to_shape();
    Material air_material = glass_material;
    air_material.refractive_index = 1.0;
    Shape air = Primitive::new_sphere()
                    .with_material(air_material)
                    .with_transform(M::scaling(0.5, 0.5, 0.5))
                    .to_shape();
    World world({floor, glass, air}, {PointLight(Point(20.0, 10.0, 0.0), Color(0.6, 0.6, 0.6))});
    Camera camera = Camera(w, h, FRAC_PI_3)
                        .with_transform(M::view_transform(Point(0.0, 3.5, 0.0), Point(0.0, 0.0, 0.0),
                                                          Vector(0.0, 0.0, 1.0)));
    return {world, camera};
}

Material checker_floor_material() {  // teapot.rs:21-30, hexagon.rs:20-29
    Material m;
    m.color = Color(1.0, 0.9, 0.9);
    m.specular = 0.0;
    m.pattern = Pattern::new_checker(Pattern::new_solid(Color::WHITE()), Pattern::new_solid(Color::BLACK()));
    return m;
}

// exercises/teapot.rs:20-54
Exercise teapot(size_t w, size_t h, const std::string& resource_dir) {
    Shape floor = Primitive::new_plane().with_material(checker_floor_material()).to_shape();
    Shape pot = obj::parse_file(resource_dir + "/teapot.obj");
    return {World({floor, pot}, {standard_light()}), standard_camera(w, h)};
}

// exercises/hexagon.rs:19-88
Primitive hexagon_corner() {
    return Primitive::new_sphere().with_transform(M::translation(0.0, 0.0, -1.0) * M::scaling(0.25, 0.25, 0.25));
}
Primitive hexagon_edge() {
    return Primitive::new_cylinder(Conic(0.0, 1.0, false))
        .with_transform(M::translation(0.0, 0.0, -1.0) * M::rotation_y(-FRAC_PI_6) * M::rotation_z(-FRAC_PI_2) *
                        M::scaling(0.25, 1.0, 0.25));
}
Compound hexagon_side() {
    return Compound::new_group({hexagon_corner().to_shape(), hexagon_edge().to_shape()});
}
Exercise hexagon(size_t w, size_t h) {
    Shape floor = Primitive::new_plane().with_material(checker_floor_material()).to_shape();
    std::vector<Shape> hex;
    for (int n = 0; n < 6; ++n)
        hex.push_back(hexagon_side().with_transform(M::rotation_y(static_cast<double>(n) * FRAC_PI_3)).to_shape());
    Shape hexagon_shape = Compound::new_group(hex)
                              .with_transform(M::translation(0.0, 1.0, 0.0) * M::rotation_x(-FRAC_PI_2) *
                                              M::scaling(0.75, 0.75, 0.75))
                              .to_shape();
    return {World({floor, hexagon_shape}, {standard_light()}), standard_camera(w, h)};
}

// exercises/csg.rs:23-200
Shape wedge(const Matrix4& transform) {
    return Primitive::new_cube()
        .with_transform(transform * M::scaling(1.0, 1.0, 0.15) * M::translation(std::sqrt(2.0), 0.0, 0.0) *
                        M::rotation_y(FRAC_PI_4))
        .with_shadow(false)
        .to_shape();
}
Exercise csg(size_t w, size_t h) {
    Material red_material;
    red_material.color = Color(1.0, 0.0, 0.0);
    red_material.ambient = 0.2;
    Material green_material;
    green_material.color = Color(0.0, 1.0, 0.0);
    green_material.ambient = 0.2;
    Material blue_material;
    blue_material.color = Color(0.0, 0.0, 1.0);
    blue_material.ambient = 0.2;

    Material dark_mirror;
    dark_mirror.color = Color::BLACK();
    dark_mirror.ambient = 0.0;
    dark_mirror.diffuse = 0.4;
    dark_mirror.reflective = 0.5;
    Material transparent;
    transparent.color = Color::BLACK();
    transparent.ambient = 0.0;
    transparent.diffuse = 0.0;
    transparent.reflective = 0.0;
    transparent.transparency = 1.0;
    transparent.refractive_index = 1.0;

    Material room_material;
    room_material.ambient = 0.1;
    room_material.diffuse = 0.7;
    room_material.reflective = 0.05;
    room_material.pattern =
        Pattern::new_checker(Pattern::new_solid(Color::WHITE()), Pattern::new_solid(Color(0.9, 0.9, 0.9)))
            .with_transform(M::scaling(0.05, 0.05, 0.05));
    Shape room = Primitive::new_cube()
                     .with_material(room_material)
                     .with_transform(M::scaling(5.0, 5.0, 5.0) * M::translation(0.0, 1.0, 0.0))
                     .to_shape();

    const auto shiny = [](Material base) {
        base.ambient = 0.1;
        base.diffuse = 0.5;
        base.reflective = 0.3;
        return base;
    };
    const Conic unit_closed(-1.0, 1.0, true);
    Shape left = Primitive::new_cylinder(unit_closed)
                     .with_material(shiny(red_material))
                     .with_transform(M::scaling(0.5, 1.1, 0.5))
                     .to_shape();
    Shape right = Compound::new_csg(Operation::Intersection,
                                    Primitive::new_cylinder(unit_closed)
                                        .with_material(shiny(green_material))
                                        .with_transform(M::rotation_x(FRAC_PI_2) * M::scaling(0.5, 1.1, 0.5))
                                        .to_shape(),
                                    Primitive::new_cylinder(unit_closed)
                                        .with_material(shiny(blue_material))
                                        .with_transform(M::rotation_z(FRAC_PI_2) * M::scaling(0.5, 1.1, 0.5))
                                        .to_shape())
                      .to_shape();
    Shape tricylinder = Compound::new_csg(Operation::Intersection, left, right)
                            .with_transform(M::translation(-1.5, 0.7, 0.0) * M::rotation_z(-0.2) *
                                            M::rotation_x(-0.1) * M::rotation_y(0.4))
                            .to_shape();

    Material sphere_mat;
    sphere_mat.color = Color(0.1, 0.1, 0.1);
    sphere_mat.ambient = 0.2;
    sphere_mat.diffuse = 0.9;
    sphere_mat.specular = 1.0;
    sphere_mat.shininess = 50.0;
    Shape sphere = Primitive::new_sphere().with_material(sphere_mat).with_transform(M::scaling(1.4, 1.4, 1.4)).to_shape();
    Shape cylinders =
        Compound::new_group({Primitive::new_cylinder(unit_closed)
                                 .with_material(red_material)
                                 .with_transform(M::scaling(0.5, 1.1, 0.5))
                                 .to_shape(),
                             Primitive::new_cylinder(unit_closed)
                                 .with_material(green_material)
                                 .with_transform(M::rotation_x(FRAC_PI_2) * M::scaling(0.5, 1.1, 0.5))
                                 .to_shape(),
                             Primitive::new_cylinder(unit_closed)
                                 .with_material(blue_material)
                                 .with_transform(M::rotation_z(FRAC_PI_2) * M::scaling(0.5, 1.1, 0.5))
                                 .to_shape()})
            .to_shape();
    Shape cube_minus_cylinders =
        Compound::new_csg(Operation::Difference, Primitive::new_cube().with_material(dark_mirror).to_shape(), cylinders)
            .to_shape();
    Shape hollowed_box = Compound::new_csg(Operation::Intersection, sphere, cube_minus_cylinders)
                             .with_transform(M::rotation_y(1.3) * M::scaling(0.5, 0.5, 0.5) *
                                             M::translation(0.0, 1.0, 0.0))
                             .to_shape();

    Shape inside = Primitive::new_sphere().with_material(red_material).to_shape();
    std::vector<Shape> children;
    for (int i = 0; i < 12; ++i) children.push_back(wedge(M::rotation_y(static_cast<double>(i) * FRAC_PI_6)));
    Shape outside = Compound::new_group(children).with_material(transparent).to_shape();
    Shape ball = Compound::new_csg(Operation::Intersection, inside, outside)
                     .with_transform(M::translation(1.5, 0.25, 0.0) * M::rotation_z(0.1) * M::rotation_x(-0.1) *
                                     M::rotation_y(-0.5) * M::scaling(0.5, 0.5, 0.5) *
                                     M::translation(0.0, 1.0, 0.0))
                     .to_shape();

    World world({room, tricylinder, hollowed_box, ball}, {PointLight(Point(-2.0, 5.0, -2.0), Color::WHITE())});
    Camera camera = Camera(w, h, 0.9).with_transform(
        M::view_transform(Point(0.0, 2.0, -4.9), Point(0.0, 0.5, 0.0), Vector(0.0, 1.0, 0.0)));
    return {world, camera};
}

// exercises/cover.rs:19-189
Exercise cover(size_t w, size_t h) {
    Material white_material;
    white_material.color = Color::WHITE();
    white_material.diffuse = 0.7;
    white_material.ambient = 0.1;
    white_material.specular = 0.0;
    white_material.reflective = 0.1;
    Material blue_material = white_material;
    blue_material.color = Color(0.537, 0.831, 0.914);
    Material red_material = white_material;
    red_material.color = Color(0.941, 0.322, 0.388);
    Material purple_material = white_material;
    purple_material.color = Color(0.373, 0.404, 0.550);

    const M standard_transform = M::scaling(0.5, 0.5, 0.5) * M::translation(1.0, -1.0, 1.0);
    const M large_object = M::scaling(3.5, 3.5, 3.5) * standard_transform;
    const M medium_object = M::scaling(3.0, 3.0, 3.0) * standard_transform;
    const M small_object = M::scaling(2.0, 2.0, 2.0) * standard_transform;

    Material backdrop_material;
    backdrop_material.color = Color::WHITE();
    backdrop_material.ambient = 1.0;
    backdrop_material.diffuse = 0.0;
    backdrop_material.specular = 0.0;
    Shape backdrop = Primitive::new_plane()
                         .with_material(backdrop_material)
                         .with_transform(M::translation(0.0, 0.0, 500.0) * M::rotation_x(FRAC_PI_2))
                         .to_shape();
    Material main_material;
    main_material.color = Color(0.373, 0.404, 0.550);
    main_material.diffuse = 0.2;
    main_material.ambient = 0.0;
    main_material.specular = 1.0;
    main_material.shininess = 200.0;
    main_material.reflective = 0.7;
    main_material.transparency = 0.7;
    main_material.refractive_index = 1.5;
    Shape main_sphere = Primitive::new_sphere().with_material(main_material).with_transform(large_object).to_shape();

    struct Spec {
        const Material* material;
        M transform;
        const M* base;
    };
    const Spec cube_specs[] = {
        {&white_material, M::translation(4.0, 0.0, 0.0), &medium_object},
        {&blue_material, M::translation(8.5, 1.5, -0.5), &large_object},
        {&red_material, M::translation(0.0, 0.0, 4.0), &large_object},
        {&white_material, M::translation(4.0, 0.0, 4.0), &small_object},
        {&purple_material, M::translation(7.5, 0.5, 4.0), &medium_object},
        {&white_material, M::translation(-0.25, 0.25, 8.0), &medium_object},
        {&blue_material, M::translation(4.0, 1.0, 7.5), &large_object},
        {&red_material, M::translation(10.0, 2.0, 7.5), &medium_object},
        {&white_material, M::translation(8.0, 2.0, 12.0), &small_object},
        {&white_material, M::translation(20.0, 1.0, 9.0), &small_object},
        {&blue_material, M::translation(-0.5, -5.0, 0.25), &large_object},
        {&red_material, M::translation(4.0, -4.0, 0.0), &large_object},
        {&white_material, M::translation(8.5, -4.0, 0.0), &large_object},
        {&white_material, M::translation(0.0, -4.0, 4.0), &large_object},
        {&purple_material, M::translation(-0.5, -4.5, 8.0), &large_object},
        {&white_material, M::translation(0.0, -8.0, 4.0), &large_object},
        {&white_material, M::translation(-0.5, -8.5, 8.0), &large_object},
    };
    std::vector<Shape> shapes{backdrop, main_sphere};
    for (const Spec& s : cube_specs)
        shapes.push_back(
            Primitive::new_cube().with_material(*s.material).with_transform(s.transform * *s.base).to_shape());

    World world(shapes, {PointLight(Point(50.0, 100.0, -50.0), Color::WHITE()),
                         PointLight(Point(-400.0, 50.0, -10.0), Color(0.7, 0.7, 0.7))});
    Camera camera = Camera(w, h, FRAC_PI_4)
                        .with_transform(M::view_transform(Point(-6.0, 6.0, -10.0), Point(6.0, -2.0, 6.0),
                                                          Vector(-0.45, 1.0, 0.0)));
    return {world, camera};
}

// exercises/room.rs:19-253
Shape make_table_leg(double x, double z) {
    Material m;
    m.color = Color(0.5529, 0.4235, 0.3255);
    m.ambient = 0.2;
    m.diffuse = 0.7;
    return Primitive::new_cube()
        .with_material(m)
        .with_transform(M::translation(x, 1.5, z) * M::scaling(0.1, 1.5, 0.1))
        .to_shape();
}
Shape plain_cube(const Color& c, const M& t, double diffuse = 0.9) {
    Material m;
    m.color = c;
    m.diffuse = diffuse;
    return Primitive::new_cube().with_material(m).with_transform(t).to_shape();
}
Exercise room(size_t w, size_t h) {
    Material fc;
    fc.ambient = 0.25;
    fc.diffuse = 0.75;
    fc.specular = 0.9;
    fc.shininess = 300.0;
    fc.reflective = 0.1;
    fc.pattern = Pattern::new_checker(Pattern::new_solid(Color::BLACK()), Pattern::new_solid(Color(0.25, 0.25, 0.25)))
                     .with_transform(M::scaling(0.07, 0.07, 0.07));
    Shape floor_ceiling = Primitive::new_cube()
                              .with_material(fc)
                              .with_transform(M::scaling(20.0, 7.0, 20.0) * M::translation(0.0, 1.0, 0.0))
                              .to_shape();
    Material wm;
    wm.ambient = 0.1;
    wm.diffuse = 0.7;
    wm.specular = 0.9;
    wm.shininess = 300.0;
    wm.reflective = 0.1;
    wm.pattern = Pattern::new_checker(Pattern::new_solid(Color(0.4863, 0.3765, 0.2941)),
                                      Pattern::new_solid(Color(0.3725, 0.2902, 0.2275)))
                     .with_transform(M::scaling(0.05, 20.0, 0.05));
    Shape walls = Primitive::new_cube().with_material(wm).with_transform(M::scaling(10.0, 10.0, 10.0)).to_shape();
    Material tm;
    tm.ambient = 0.1;
    tm.diffuse = 0.7;
    tm.specular = 0.9;
    tm.shininess = 300.0;
    tm.reflective = 0.2;
    tm.pattern = Pattern::new_stripe(Pattern::new_solid(Color(0.5529, 0.4235, 0.3255)),
                                     Pattern::new_solid(Color(0.6588, 0.5098, 0.4000)))
                     .with_transform(M::scaling(0.05, 0.05, 0.05) * M::rotation_y(0.1));
    Shape table_top = Primitive::new_cube()
                          .with_material(tm)
                          .with_transform(M::translation(0.0, 3.1, 0.0) * M::scaling(3.0, 0.1, 2.0))
                          .to_shape();
    Shape leg1 = make_table_leg(2.7, -1.7);
    Shape leg2 = make_table_leg(2.7, 1.7);
    Shape leg3 = make_table_leg(-2.7, -1.7);
    Shape leg4 = make_table_leg(-2.7, 1.7);

    Material gm;
    gm.color = Color(1.0, 1.0, 0.8);
    gm.ambient = 0.0;
    gm.diffuse = 0.3;
    gm.specular = 0.9;
    gm.shininess = 300.0;
    gm.reflective = 0.7;
    gm.transparency = 0.7;
    gm.refractive_index = 1.5;
    Shape glass_cube = Primitive::new_cube()
                           .with_material(gm)
                           .with_transform(M::translation(0.0, 3.450001, 0.0) * M::rotation_y(0.2) *
                                           M::scaling(0.25, 0.25, 0.25))
                           .with_shadow(false)
                           .to_shape();
    Material l1;
    l1.color = Color(1.0, 0.5, 0.5);
    l1.diffuse = 0.4;
    l1.reflective = 0.6;
    Shape little_cube1 = Primitive::new_cube()
                             .with_material(l1)
                             .with_transform(M::translation(1.0, 3.35, -0.9) * M::rotation_y(-0.4) *
                                             M::scaling(0.15, 0.15, 0.15))
                             .to_shape();
    Shape little_cube2 = plain_cube(Color(1.0, 0.5, 0.5),
                                    M::translation(-1.5, 3.27, 0.3) * M::rotation_y(0.4) * M::scaling(0.15, 0.17, 0.15));
    Shape little_cube3 = plain_cube(Color(0.5, 1.0, 0.5),
                                    M::translation(0.0, 3.25, 1.0) * M::rotation_y(0.4) * M::scaling(0.2, 0.05, 0.05));
    Shape little_cube4 = plain_cube(Color(0.5, 0.5, 1.0),
                                    M::translation(-0.6, 3.4, -1.0) * M::rotation_y(0.8) * M::scaling(0.05, 0.2, 0.05));
    Shape little_cube5 = plain_cube(Color(0.5, 1.0, 1.0),
                                    M::translation(2.0, 3.4, 1.0) * M::rotation_y(0.8) * M::scaling(0.05, 0.2, 0.05));
    Shape frame1 = plain_cube(Color(0.7098, 0.2471, 0.2196),
                              M::translation(-10.0, 4.0, 1.0) * M::scaling(0.05, 1.0, 1.0), 0.6);
    Shape frame2 = plain_cube(Color(0.2667, 0.2706, 0.6902),
                              M::translation(-10.0, 3.4, 2.7) * M::scaling(0.05, 0.4, 0.4), 0.6);
    Shape frame3 = plain_cube(Color(0.3098, 0.5961, 0.3098),
                              M::translation(-10.0, 4.6, 2.7) * M::scaling(0.05, 0.4, 0.4), 0.6);
    Shape mirror_frame = plain_cube(Color(0.3882, 0.2627, 0.1882),
                                    M::translation(-2.0, 3.5, 9.95) * M::scaling(5.0, 1.5, 0.05), 0.7);
    Material mm;
    mm.color = Color::BLACK();
    mm.ambient = 0.0;
    mm.diffuse = 0.0;
    mm.specular = 1.0;
    mm.shininess = 300.0;
    mm.reflective = 1.0;
    Shape mirror = Primitive::new_cube()
                       .with_material(mm)
                       .with_transform(M::translation(-2.0, 3.5, 9.95) * M::scaling(4.8, 1.4, 0.06))
                       .to_shape();

    World world({floor_ceiling, walls, table_top, leg1, leg2, leg3, leg4, glass_cube, little_cube1, little_cube2,
                 little_cube3, little_cube4, little_cube5, frame1, frame2, frame3, mirror_frame, mirror},
                {PointLight(Point(0.0, 6.9, -5.0), Color::WHITE())});
    Camera camera = Camera(w, h, FRAC_PI_4)
                        .with_transform(M::view_transform(Point(8.0, 6.0, -8.0), Point(0.0, 3.0, 0.0),
                                                          Vector(0.0, 1.0, 0.0)));
    return {world, camera};
}

}  // namespace

// exercises/shading.rs:17-63
Canvas shading_exercise() {
    const size_t side_len = 500;
    Canvas canvas = Canvas::empty(side_len, side_len);
    const Point ray_origin(0.0, 0.0, -5.0);
    const double wall_z = 10.0;
    const double wall_size = 7.0;
    const double pixel_size = wall_size / static_cast<double>(side_len);
    const double half = wall_size / 2.0;
    Material m;
    m.color = Color(1.0, 0.2, 1.0);
    const Shape sphere = Primitive::new_sphere().with_material(m).to_shape();
    const PointLight light(Point(-10.0, 10.0, -10.0), Color::WHITE());
    std::vector<Ray> rays;
    rays.reserve(side_len * side_len);
    for (size_t y = 0; y < side_len; ++y) {
        const double world_y = half - pixel_size * static_cast<double>(y);
        for (size_t x = 0; x < side_len; ++x) {
            const double world_x = -half + pixel_size * static_cast<double>(x);
            const Point position(world_x, world_y, wall_z);
            rays.push_back(Ray{ray_origin, (position - ray_origin).normalize()});
        }
    }
    const ShadedRays shaded = shade_rays(sphere, light, rays);  // :45-56 for every pixel at once
    for (size_t y = 0; y < side_len; ++y)
        for (size_t x = 0; x < side_len; ++x)
            if (shaded.hit[y * side_len + x]) canvas.write_pixel(x, y, shaded.colors[y * side_len + x]);
    return canvas;
}

const std::vector<std::string>& exercise_names() {
    static const std::vector<std::string> names = {"plane",  "scene", "pattern", "glass_bubble", "teapot",
                                                   "hexagon", "csg",   "cover",   "room",         "synthetic"};
    return names;
}

Exercise build_exercise(const std::string& name, size_t width, size_t height, const std::string& resource_dir) {
    (void)PI;
    if (name == "plane") return plane(width, height);
    if (name == "scene") return scene(width, height);
    if (name == "pattern") return pattern(width, height);
    if (name == "glass_bubble") return glass_bubble(width, height);
    if (name == "teapot") return teapot(width, height, resource_dir);
    if (name == "hexagon") return hexagon(width, height);
    if (name == "csg") return csg(width, height);
    if (name == "cover") return cover(width, height);
    if (name == "room") return room(width, height);
    if (name == "synthetic") return synthetic_scene(width, height, 10000, 0x5EED5EEDull);
    throw std::invalid_argument("unknown exercise: " + name);
}

}  // namespace rt
