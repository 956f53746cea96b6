"""Counted FLOP budget of one render.

The counting variant of the render kernel tallies events (RtStats); each event has a fixed
FP64 cost derived from the reference source (add/sub/mul/div/sqrt = 1; neg/abs/min/max/compare
= 0).  `pow` is tallied separately and folded in at a nominal 60.  Acceleration-structure box
tests are work this implementation adds and are priced like the reference's slab test.

Two numerators, kept apart (most box tests of the large-scene kernels are f32):
  algorithmic_flops(stats)   FP64 work -- everything the reference's arithmetic asks for plus the f64 box
                             tests of the skip-link hierarchies (`bvh_box_tests`); priced against the FP64
                             vector peak;
  accel_f32_flops(stats)     the f32 tests of the 4-wide hierarchy (`accel_f32_tests`: four child boxes per
                             node visited, one pre-test per leaf), at the same 12 per test; priced against
                             the FP32 vector peak.  They are NOT part of the FP64 fraction.
"""

COST = {
    "ray_for_pixel": 54,       # camera.rs:68-80
    "ray_transform": 33,       # ray.rs:19-24 (M*P 18 + M*V 15)
    "prim": (31, 1, 12, 47, 53, 46, 46),  # sphere plane cube cylinder cone triangle smooth-triangle (primitive.rs:269-440)
    "slab_test": 12,           # bounds.rs:53-77
    "prepare": 37,             # intersection.rs:76-99
    "normal_level": 21,        # (M*P 18 + Mt*V 15 + normalize 9) / 2 per level walked (primitive.rs:153-192)
    "shadow_setup": 18,        # world.rs:86-89
    "lighting": 60,            # primitive.rs:111-151
    "pow": 60,                 # nominal
    "pattern_node": 18,        # pattern.rs:87-91
    "refract": 26,             # world.rs:109-127
    "schlick": 20,             # intersection.rs:182-200
    "combine": 6,              # world.rs:76-82
    "light_sum": 3,            # world.rs:58-70
}


def total_rays(stats):
    return stats["rays_primary"] + stats["rays_shadow"] + stats["rays_reflect"] + stats["rays_refract"]


def algorithmic_flops(stats):
    """FP64 operations one launch performs, from its event counts (a dict as RtStats.as_dict())."""
    f = COST["ray_for_pixel"] * stats["rays_primary"]
    f += COST["ray_transform"] * stats["ray_transforms"]
    f += sum(c * n for c, n in zip(COST["prim"], stats["prim_tests"]))
    f += COST["slab_test"] * (stats["group_gates"] + stats["bvh_box_tests"])
    f += COST["prepare"] * stats["hits_shaded"]
    f += COST["normal_level"] * stats["normal_levels"]
    f += COST["shadow_setup"] * stats["rays_shadow"]
    f += (COST["lighting"] + COST["light_sum"]) * stats["lightings"]
    f += COST["pow"] * stats["pow_calls"]
    f += COST["pattern_node"] * stats["pattern_nodes"]
    f += COST["refract"] * stats["rays_refract"]
    f += COST["schlick"] * stats["schlicks"]
    f += COST["combine"] * stats["hits_shaded"]
    return f


def accel_f32_flops(stats):
    """f32 operations of the 4-wide hierarchy's box tests and leaf pre-tests (acceleration only)."""
    return COST["slab_test"] * stats.get("accel_f32_tests", 0)


# MI355X FP64 vector peak: 256 CU x 4 SIMD x 16 FP64 lanes/clk x 2 (FMA) x 2.4 GHz
# (= half the 157.3 TFLOP/s FP32 vector figure of MI355X_MICROARCH.md).  The path is
# compiled with -ffp-contract=off (the reference never fuses a*b+c), so MUL and ADD issue
# separately and the attainable ceiling is half of that.
PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_FP64_VALU_NO_FMA_TFLOPS = 39.3
PEAK_FP32_VALU_TFLOPS = 157.3
