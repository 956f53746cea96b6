"""Known-answer scenarios taken from the reference's inline unit tests (5-digit values,
compared at the reference's own 1e-4 tolerance).  Each scenario builds its world through the
mirrored public API and names rays plus the values the reference asserts.  The same list is
run against the CPU oracle (tests/test_oracle_kat.py, CPU) and against the HIP path
(tests/test_gpu_kat.py, GPU).
"""
import math

import numpy as np

import raytracer_amd as rt
from raytracer_amd import Material, Matrix, Pattern, PointLight, Primitive, World

S2 = math.sqrt(2.0) / 2.0
WHITE = (1.0, 1.0, 1.0)


def default_parts(s1_material=None, s2_material=None, light=None):
    """World::default (world.rs:130-145) with optional substitutions."""
    m1 = s1_material or Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)
    s1 = Primitive.new_sphere().with_material(m1)
    s2 = Primitive.new_sphere().with_transform(Matrix.scaling(0.5, 0.5, 0.5))
    if s2_material is not None:
        s2 = s2.with_material(s2_material)
    return [s1, s2], [light or PointLight((-10.0, 10.0, -10.0), WHITE)]


def reflective_floor(**kw):
    return Primitive.new_plane().with_transform(Matrix.translation(0.0, -1.0, 0.0)).with_material(Material(**kw))


def red_ball():
    return (Primitive.new_sphere().with_transform(Matrix.translation(0.0, -3.5, -0.5))
            .with_material(Material(color=(1.0, 0.0, 0.0), ambient=0.5)))


# (name, world builder, ray, remaining, expected colour, reference citation)
def color_at_cases():
    cases = []
    shapes, lights = default_parts()
    w = World(shapes, lights)
    cases.append(("shade_intersection", w, (0, 0, -5, 0, 0, 1), 5, (0.38066, 0.47583, 0.2855), "world.rs:199-216"))
    cases.append(("color_ray_misses", w, (0, 0, -5, 0, 1, 0), 5, (0.0, 0.0, 0.0), "world.rs:235-243"))
    shapes, lights = default_parts(light=PointLight((0.0, 0.25, 0.0), WHITE))
    cases.append(("shade_intersection_from_inside", World(shapes, lights), (0, 0, 0, 0, 0, 1), 5,
                  (0.90498, 0.90498, 0.90498), "world.rs:218-233"))
    shapes, lights = default_parts(s1_material=Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0),
                                   s2_material=Material(ambient=1.0))
    cases.append(("color_intersection_behind_ray", World(shapes, lights), (0, 0, 0.75, 0, 0, -1), 5, WHITE,
                  "world.rs:255-266"))
    s1 = Primitive.new_sphere()
    s2 = Primitive.new_sphere().with_transform(Matrix.translation(0.0, 0.0, 10.0))
    cases.append(("shade_hit_given_intersection_in_shadow",
                  World([s1, s2], [PointLight((0.0, 0.0, -10.0), WHITE)]), (0, 0, 5, 0, 0, 1), 5, (0.1, 0.1, 0.1),
                  "world.rs:300-315"))
    shapes, lights = default_parts()
    cases.append(("shade_hit_with_reflective_material", World(shapes + [reflective_floor(reflective=0.5)], lights),
                  (0, 0, -3, 0, -S2, S2), 5, (0.87677, 0.92436, 0.82918), "world.rs:362-386"))
    shapes, lights = default_parts()
    cases.append(("shade_hit_with_transparent_material",
                  World(shapes + [reflective_floor(transparency=0.5, refractive_index=1.5), red_ball()], lights),
                  (0, 0, -3, 0, -S2, S2), 5, (0.93642, 0.68642, 0.68642), "world.rs:549-584"))
    shapes, lights = default_parts()
    cases.append(("shade_hit_with_reflective_transparent_material",
                  World(shapes + [reflective_floor(reflective=0.5, transparency=0.5, refractive_index=1.5), red_ball()],
                        lights),
                  (0, 0, -3, 0, -S2, S2), 5, (0.93391, 0.69643, 0.69243), "world.rs:586-622"))
    # refracted_color_with_refracted_ray (world.rs:510-547): the glass sphere's own surface terms are
    # zeroed so that color_at == refracted_color
    m1 = Material(color=(0.8, 1.0, 0.6), ambient=1.0, diffuse=0.7, specular=0.2, pattern=Pattern.new_test())
    m2 = Material(transparency=1.0, refractive_index=1.5, ambient=0.0, diffuse=0.0, specular=0.0)
    shapes, lights = default_parts(s1_material=m1, s2_material=m2)
    cases.append(("refracted_color_with_refracted_ray", World(shapes, lights), (0, 0, 0.1, 0, 1, 0), 5,
                  (0.0, 0.99888, 0.04725), "world.rs:510-547"))
    # total internal reflection (world.rs:482-508): surface terms zeroed -> BLACK
    m1 = Material(color=(0.8, 1.0, 0.6), diffuse=0.0, specular=0.0, ambient=0.0, transparency=1.0, refractive_index=1.5)
    shapes, lights = default_parts(s1_material=m1, s2_material=Material(ambient=0.0, diffuse=0.0, specular=0.0))
    cases.append(("refracted_color_under_total_internal_refraction", World([shapes[0]], lights),
                  (0, 0, S2, 0, 1, 0), 5, (0.0, 0.0, 0.0), "world.rs:482-508"))
    return cases


def mutually_reflective_world():
    """world.rs:388-411: two facing mirrors; color_at must terminate."""
    shapes, lights = default_parts()
    lower = reflective_floor(reflective=1.0)
    upper = Primitive.new_plane().with_transform(Matrix.translation(0.0, 1.0, 0.0)).with_material(Material(reflective=1.0))
    return World(shapes + [lower, upper], lights), (0, 0, 0, 0, 1, 0)


def shadow_cases():
    """world.rs:268-298: is_shadowed(point) on the default world."""
    return [((0.0, 10.0, 0.0), False), ((10.0, -10.0, 10.0), True), ((-20.0, 20.0, -20.0), False),
            ((-2.0, 2.0, -2.0), False)]


def glass_spheres_world():
    """intersection.rs:342-394 finding_n1_and_n2_at_various_intersections."""
    def glass(ri):
        return Material(transparency=1.0, refractive_index=ri)
    a = Primitive.new_sphere().with_transform(Matrix.scaling(2.0, 2.0, 2.0)).with_material(glass(1.5))
    b = Primitive.new_sphere().with_transform(Matrix.translation(0.0, 0.0, -0.25)).with_material(glass(2.0))
    c = Primitive.new_sphere().with_transform(Matrix.translation(0.0, 0.0, 0.25)).with_material(glass(2.5))
    world = World([a, b, c], [PointLight((-10.0, 10.0, -10.0), WHITE)])
    # intersections from z=-4 are at t = 2, 2.75, 3.25, 4.75, 5.25, 6; start each ray 0.1 before the k-th
    ts = [2.0, 2.75, 3.25, 4.75, 5.25, 6.0]
    rays = [(0.0, 0.0, -4.0 + t - 0.1, 0.0, 0.0, 1.0) for t in ts]
    expected = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    return world, rays, expected


def schlick_cases():
    """intersection.rs:411-455: (eye, normal, n1, n2, reflectance) of the three Schlick tests."""
    return [
        ((0.0, -1.0, 0.0), (0.0, -S2, -S2), 1.5, 1.0, 1.0),        # total internal reflection
        ((0.0, -1.0, 0.0), (0.0, -1.0, 0.0), 1.5, 1.0, 0.04),      # perpendicular viewing angle
        ((0.0, 0.0, -1.0), (0.0, 0.99, -0.14106735979665894), 1.0, 1.5, 0.48873),  # small angle, n2 > n1
    ]


def random_rays(n, seed, origin_scale=6.0):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-origin_scale, origin_scale, size=(n, 3))
    target = rng.uniform(-2.0, 2.0, size=(n, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1)


def primitive_zoo():
    """Every primitive kind, nested groups and a CSG in one small world, for probe/color_at sweeps."""
    checker = Pattern.new_checker(Pattern.new_solid((0.9, 0.1, 0.1)), Pattern.new_solid((0.1, 0.1, 0.9)))
    blend = Pattern.new_blend(
        Pattern.new_stripe(Pattern.new_solid(WHITE), Pattern.new_solid((0.0, 0.5, 0.0))).with_transform(
            Matrix.scaling(0.3, 0.3, 0.3)),
        Pattern.new_radial_gradient(Pattern.new_solid((1.0, 1.0, 0.0)), Pattern.new_solid((0.0, 1.0, 1.0))))
    grad = Pattern.new_gradient(Pattern.new_solid((1.0, 0.0, 0.0)),
                                Pattern.new_ring(Pattern.new_solid(WHITE), Pattern.new_solid((0.2, 0.2, 0.2)))
                                .with_transform(Matrix.scaling(0.2, 0.2, 0.2)))
    perturbed = Pattern.new_perturb(checker.with_transform(Matrix.scaling(0.4, 0.4, 0.4)))
    floor = Primitive.new_plane().with_transform(Matrix.translation(0.0, -2.0, 0.0)).with_material(
        Material(pattern=checker, reflective=0.2))
    glass = Primitive.new_glass_sphere().with_transform(Matrix.translation(-1.5, 0.0, 0.0)).with_material(
        Material(transparency=0.8, reflective=0.6, refractive_index=1.5, diffuse=0.1))
    cube = Primitive.new_cube().with_transform(
        Matrix.chain(Matrix.translation(1.5, 0.0, 0.5), Matrix.rotation_y(0.6), Matrix.scaling(0.7, 0.7, 0.7))
    ).with_material(Material(pattern=blend))
    cyl = Primitive.new_cylinder(rt.Conic(-1.0, 1.0, True)).with_transform(
        Matrix.chain(Matrix.translation(0.0, 0.0, 2.5), Matrix.rotation_x(0.4))).with_material(Material(pattern=grad))
    cone = Primitive.new_cone(rt.Conic(-1.0, 0.5, True)).with_transform(
        Matrix.chain(Matrix.translation(0.3, 1.5, 0.0), Matrix.scaling(0.6, 0.8, 0.6))).with_material(
        Material(color=(0.8, 0.5, 0.2), pattern=perturbed))
    open_cyl = Primitive.new_cylinder(rt.Conic(-0.5, 0.5, False)).with_transform(Matrix.translation(-3.0, 1.0, 1.0))
    tri = Primitive.new_triangle((-1.0, -1.0, -2.0), (1.0, -1.0, -2.0), (0.0, 1.0, -2.5)).with_material(
        Material(color=(0.2, 0.9, 0.4)))
    stri = Primitive.new_smooth_triangle((2.0, 1.0, -1.0), (3.0, 1.0, -1.5), (2.5, 2.5, -1.2),
                                         (0.0, 0.2, -1.0), (0.3, 0.0, -1.0), (0.0, 0.0, -1.0))
    inner = rt.Compound.new_group([tri, stri, open_cyl]).with_transform(
        Matrix.chain(Matrix.rotation_y(0.3), Matrix.scaling(0.9, 1.1, 0.9)))
    outer = rt.Compound.new_group([inner, cone]).with_transform(Matrix.translation(0.2, 0.1, -0.3))
    lens = rt.Compound.new_csg(
        rt.Operation.Difference,
        Primitive.new_cube().with_material(Material(color=(0.9, 0.9, 0.2))),
        rt.Compound.new_csg(rt.Operation.Union,
                            Primitive.new_sphere().with_transform(Matrix.scaling(1.2, 1.2, 1.2)).with_material(
                                Material(color=(0.2, 0.2, 0.9), transparency=0.5, refractive_index=1.3)),
                            Primitive.new_cylinder(rt.Conic(-2.0, 2.0, True)).with_transform(
                                Matrix.scaling(0.4, 1.0, 0.4)))
    ).with_transform(Matrix.chain(Matrix.translation(0.0, 0.0, -4.0), Matrix.rotation_y(0.5), Matrix.scaling(0.8, 0.8, 0.8)))
    ghost = Primitive.new_sphere().with_shadow(False).with_transform(Matrix.translation(3.0, 3.0, 3.0))
    lights = [PointLight((-10.0, 10.0, -10.0), WHITE), PointLight((8.0, 6.0, -3.0), (0.4, 0.4, 0.5))]
    with_ghost = World([floor, glass, cube, cyl, outer, lens, ghost], lights)
    without_ghost = World([floor, glass, cube, cyl, outer, lens], lights)
    return with_ghost, without_ghost
