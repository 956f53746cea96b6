"""SURVEY.md §7 step 2's gate: the oracle rendering from the FLATTENED buffers == the oracle rendering from the object
tree.  The flattener (host/rt_flatten.cpp) is otherwise validated only through GPU == oracle; here its tables -- indices,
skip links, parents, kinds and flags, conics, triangles, materials, pattern trees, CSG operations, lights -- are read back
into the oracle's own scene format (tests/unflatten.py) and rendered on the CPU beside the object tree the user built.
No GPU.  The tables keep inverse matrices only, so the comparison is to rounding: colours within 1e-9 relative; a pixel whose
discrete decision flips on the ~1e-16 difference of a re-inverted matrix (a stripe's edge, a grazing feeler) is allowed for
one pixel in a thousand."""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import orc
from unflatten import description_from_flat


@pytest.mark.parametrize("name", ["scene", "pattern", "hexagon", "cover", "room", "csg", "glass_bubble"])
@pytest.mark.parametrize("hierarchy", [True, False])
def test_oracle_from_flat_tables_matches_oracle_from_tree(name, hierarchy):
    w, h = 64, 36
    world, cam = rt.exercise(name, w, h)
    flat = world.flatten(build_bvh=hierarchy)
    tree_img, tree_counts = orc.OracleWorld(world.describe()).render(cam)
    flat_img, flat_counts = orc.OracleWorld(description_from_flat(flat.desc.contents)).render(cam)
    close = np.isclose(flat_img, tree_img, rtol=1e-9, atol=1e-12).all(axis=2)
    assert close.mean() >= 0.999, f"{name}: {(~close).sum()} of {close.size} pixels differ"
    assert abs(sum(flat_counts) - sum(tree_counts)) <= max(4, sum(tree_counts) // 1000)


def test_oracle_from_flat_tables_synthetic_and_teapot():
    for world, cam in (rt.synthetic(64, 36, 600), rt.exercise("teapot", 48, 27)):
        tree_img, _ = orc.OracleWorld(world.describe()).render(cam)
        for mode in ("nested", "global"):
            flat = world.flatten(build_bvh=mode)
            flat_img, _ = orc.OracleWorld(description_from_flat(flat.desc.contents)).render(cam)
            close = np.isclose(flat_img, tree_img, rtol=1e-9, atol=1e-12).all(axis=2)
            assert close.mean() >= 0.999, f"{mode}: {(~close).sum()} of {close.size} pixels differ"
