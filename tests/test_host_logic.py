"""Host side of the boundary, on CPU: scene flattening, table validation, PPM writer, OBJ
parser, row-band tiling and the FLOP budget."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as rt
import scenarios
from oracle import orc
from raytracer_amd import _native as N
from raytracer_amd import flops, tiling
from raytracer_amd import Conic, Material, Matrix, Operation, Primitive, World


def create(desc):
    p = C.c_void_p()
    rc = N.hip.rt_hip_scene_create(desc, C.byref(p))
    if rc == 0:
        N.hip.rt_hip_scene_destroy(p)
    return rc


def test_flatten_is_preorder_with_skip_links():
    world, _ = scenarios.primitive_zoo()
    flat = world.flatten()
    d = flat.desc.contents
    shapes = [d.shapes[i] for i in range(d.n_shapes)]
    kinds = [s.kind_flags & 0xff for s in shapes]
    assert kinds.count(N.RT_GROUP) == 2 and kinds.count(N.RT_CSG) == 2
    for i, s in enumerate(shapes):
        assert i < s.skip <= d.n_shapes
        if kinds[i] <= N.RT_SMOOTH_TRIANGLE:
            assert s.skip == i + 1 and s.material < d.n_materials
        for c in range(i + 1, s.skip):
            assert shapes[c].depth > s.depth
    # oracle numbers its nodes in the same depth-first pre-order
    assert orc.OracleWorld(world.describe()).num_nodes == d.n_shapes
    # a valid scene passes validation; with no GPU the next step reports NO_DEVICE, never "ok"
    assert create(flat.desc) in (N.RT_OK, N.RT_ERR_NO_DEVICE)


def test_lhit_table_matches_oracle_includes():
    world, _ = scenarios.primitive_zoo()
    flat = world.flatten()
    d = flat.desc.contents
    ow = orc.OracleWorld(world.describe())
    for i in range(d.n_shapes):
        s = d.shapes[i]
        if (s.kind_flags & 0xff) == N.RT_CSG:
            c = d.csgs[s.param]
            for p in range(i + 1, s.skip):
                is_prim = (d.shapes[p].kind_flags & 0xff) <= N.RT_SMOOTH_TRIANGLE
                want = ow.csg_includes_left(i, p) if is_prim else 0
                assert d.lhit[c.lhit_offset + p - i - 1] == want


def test_group_gate_box_matches_oracle_bounds():
    s = Primitive.new_sphere().with_transform(Matrix.translation(2, 5, -3))
    g = rt.Compound.new_group([s, Primitive.new_cube()])          # untransformed group: gate == stored bounds
    world = World([g], [])
    flat = world.flatten()
    d = flat.desc.contents
    mn, mx = orc.OracleWorld(world.describe()).bounds(0)
    assert np.array_equal(np.array(list(d.groups[0].bmin)), mn)
    assert np.array_equal(np.array(list(d.groups[0].bmax)), mx)


@pytest.mark.parametrize("mutate,code", [
    (lambda d: setattr(d.shapes[0], "skip", 0), N.RT_ERR_INVALID_SCENE),
    (lambda d: setattr(d.shapes[0], "kind_flags", 77), N.RT_ERR_INVALID_SCENE),
    (lambda d: setattr(d.shapes[1], "material", 10 ** 6), N.RT_ERR_INVALID_SCENE),
    (lambda d: setattr(d.shapes[1], "parent", 5), N.RT_ERR_INVALID_SCENE),
    (lambda d: setattr(d.materials[0], "pattern", 10 ** 6), N.RT_ERR_INVALID_SCENE),
    (lambda d: setattr(d, "world_bvh_root", 10 ** 6), N.RT_ERR_INVALID_SCENE),
])
def test_malformed_tables_are_rejected(mutate, code):
    world, _ = scenarios.primitive_zoo()
    flat = world.flatten()
    mutate(flat.desc.contents)
    assert create(flat.desc) == code


def test_bvh_links_only_go_forward():
    world, _ = rt.synthetic(64, 36, 500)
    flat = world.flatten(build_bvh=True)
    d = flat.desc.contents
    assert d.n_bvh_nodes > 0
    leaves = set()
    for k in range(d.n_bvh_nodes):
        nd = d.bvh_nodes[k]
        assert nd.miss == -1 or k < nd.miss < d.n_bvh_nodes
        if nd.shape >= 0:
            leaves.add(nd.shape)
            bmin, bmax = np.array(list(nd.bmin)), np.array(list(nd.bmax))
            assert (bmin <= bmax).all()
    # every shape that sits directly under a BVH-accelerated level is named by exactly one leaf
    assert len(leaves) > 400
    assert create(flat.desc) in (N.RT_OK, N.RT_ERR_NO_DEVICE)


def test_nesting_deeper_than_the_stack_is_rejected():
    s = Primitive.new_sphere()
    for _ in range(N.RT_MAX_NEST + 1):
        s = rt.Compound.new_group([s])
    flat = World([s], []).flatten()
    assert create(flat.desc) == N.RT_ERR_NEST_TOO_DEEP


def test_depth_and_argument_checks():
    cd = rt.Camera(8, 8, 1.0).desc()
    flat = World.default().flatten()
    out = np.zeros(8 * 8 * 3)
    rc = N.hip.rt_hip_render(flat.desc, C.byref(cd), N.RT_MAX_DEPTH + 1, out.ctypes.data_as(N.c_double_p), None)
    assert rc == N.RT_ERR_DEPTH
    assert N.hip.rt_hip_render(flat.desc, None, 5, out.ctypes.data_as(N.c_double_p), None) == N.RT_ERR_INVALID_ARG
    assert N.hip.rt_hip_scene_create(None, None) == N.RT_ERR_INVALID_ARG


def test_singular_transform_raises_like_the_reference_asserts():  # matrix.rs:94
    with pytest.raises(RuntimeError):
        Primitive.new_sphere().with_transform(np.zeros((4, 4)))


def test_canvas_ppm_layout():  # canvas.rs:125-203
    c = rt.Canvas(5, 3)
    c.write_pixel(0, 0, (1.5, 0, 0)); c.write_pixel(2, 1, (0, 0.5, 0)); c.write_pixel(4, 2, (-0.5, 0, 1))
    lines = c.to_ppm().decode().split("\n")
    assert lines[:3] == ["P3", "5 3", "255"]
    assert lines[3] == "255 0 0 0 0 0 0 0 0 0 0 0 0 0 0"
    assert lines[4] == "0 0 0 0 0 0 0 128 0 0 0 0 0 0 0"
    assert lines[5] == "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"
    wide = rt.Canvas(10, 2, np.tile(np.array([1.0, 0.8, 0.6]), (2, 10, 1)))
    text = wide.to_ppm().decode()
    assert all(len(l) <= 70 for l in text.split("\n")) and text.endswith("\n")
    assert text == orc.to_ppm(wide.pixels).decode()
    rng = np.random.default_rng(3)
    px = rng.uniform(-0.2, 1.2, size=(7, 33, 3))
    assert rt.Canvas(33, 7, px).to_ppm() == orc.to_ppm(px)
    assert np.array_equal(rt.Canvas(33, 7, px).quantized(), orc.quantize(px))


@pytest.mark.parametrize("w,h", [(0, 0), (1, 1), (5, 0), (6, 1), (17, 3), (24, 5), (33, 7), (701, 611)])
def test_ppm_from_rgb8_matches_the_oracle_writer(w, h):  # canvas.rs:46-83, incl. the threaded two-pass path
    rng = np.random.default_rng(w * 1000 + h)
    # mixed token widths so the 70-column breaks fall everywhere: runs of 1-, 2- and 3-digit samples
    q = rng.choice(np.array([0, 7, 9, 10, 42, 99, 100, 128, 255], dtype=np.uint8), size=(h, w, 3))
    text = rt.Canvas.ppm_from_rgb8(q)
    assert text == orc.to_ppm(q.astype(np.float64) / 255.0)
    assert all(len(line) <= 70 for line in text.split(b"\n"))
    if w and h:
        assert np.array_equal(np.array(text.split()[4:], dtype=np.int64), q.reshape(-1))


def test_obj_parser():  # obj.rs:187-350
    with pytest.raises(RuntimeError, match="no shapes found"):   # obj.rs:47: gibberish alone yields no shape
        rt.obj.parse_string("There was a young lady named Bright\nwho traveled much faster than light.\n")
    shape, ignored = rt.obj.parse_string("gibberish line\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nanother one\n")
    assert ignored == 2
    text = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\n\nf 1 2 3 4 5\n"
    shape, _ = rt.obj.parse_string(text)
    keep = World([shape], []).flatten()
    flat = keep.desc.contents
    assert flat.n_triangles == 3                                     # fan triangulation (obj.rs:132-140)
    t = flat.triangles[1]
    assert list(t.p1) == [-1, 1, 0] and list(t.e1) == [2, -1, 0] and list(t.e2) == [2, 0, 0]
    text = "v 0 1 0\nv -1 0 0\nv 1 0 0\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\nf 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2\n"
    shape, _ = rt.obj.parse_string(text)
    keep = World([shape], []).flatten()
    flat = keep.desc.contents
    assert flat.n_triangles == 2
    assert list(flat.triangles[0].n1) == [0, 1, 0] and list(flat.triangles[0].n2) == [-1, 0, 0]
    text = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4\n"
    shape, _ = rt.obj.parse_string(text)
    assert shape.kind == N.RT_GROUP and len(shape.children()) == 2


def test_teapot_obj_loads():
    world, _ = rt.exercise("teapot", 32, 18)
    flat = world.flatten()
    d = flat.desc.contents
    assert d.n_triangles == 9216 and d.n_shapes == 9216 + 2


def test_tiling_round_trip():
    for vsize, n_parts, band in [(2160, 1, 16), (2160, 8, 16), (1080, 3, 16), (37, 4, 8), (5, 8, 16)]:
        rows = [tiling.image_rows(vsize, p, n_parts, band) for p in range(n_parts)]
        seen = np.concatenate([r[r >= 0] for r in rows])
        assert sorted(seen) == list(range(vsize))
        slots = tiling.padded_local_rows(vsize, n_parts, band)
        frame = np.arange(vsize * 4 * 3, dtype=float).reshape(vsize, 4, 3)
        shards = []
        for p in range(n_parts):
            r = tiling.image_rows(vsize, p, n_parts, band, n_slots=slots)
            s = np.full((slots, 4, 3), np.nan)
            s[r >= 0] = frame[r[r >= 0]]
            shards.append(s)
        assert np.array_equal(tiling.assemble(shards, vsize, n_parts, band), frame)


def test_flop_budget():
    stats = {k: 0 for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "ray_transforms",
                            "group_gates", "bvh_box_tests", "hits_shaded", "normal_levels", "lightings", "pow_calls",
                            "pattern_nodes", "container_passes", "list_entries", "schlicks")}
    stats["prim_tests"] = [0] * 7
    assert flops.algorithmic_flops(stats) == 0
    # SURVEY.md 8(d) worked example: a config-1 primary ray that hits a sphere and is lit, about 685 FLOPs
    stats.update(rays_primary=1, rays_shadow=1, ray_transforms=8, hits_shaded=1, normal_levels=2, lightings=1, pow_calls=1)
    stats["prim_tests"] = [6, 2, 0, 0, 0, 0, 0]
    assert 600 < flops.algorithmic_flops(stats) < 800
    assert flops.total_rays(stats) == 2


def test_direction_index_cells_left_to_the_device_from_64_candidates():
    """Flattener contract of RtSceneDesc.dir_cell_start: scenes with fewer than 64 candidate conics get their direction
    cells from the host; from 64 on only the candidates are listed (the library rasterises the cells on the device) unless
    the caller asks for host cells.  The candidate records are the same either way, and host-built tables are well formed
    (offsets monotone from 0 to n_dir_entries, entries name candidates)."""
    world, _ = rt.synthetic(64, 36, 1500)
    dev, host = world.flatten(), world.flatten(host_cells=True)
    a, b = dev.desc.contents, host.desc.contents
    assert a.n_dir_cands >= 64 and a.n_dir_cands == b.n_dir_cands and a.dir_res == b.dir_res != 0
    assert not a.dir_cell_start and not a.dir_entries and a.n_dir_entries == 0
    n_cells = 6 * b.dir_res * b.dir_res
    cells = np.ctypeslib.as_array(b.dir_cell_start, shape=(n_cells + 1,))
    entries = np.ctypeslib.as_array(b.dir_entries, shape=(b.n_dir_entries,))
    assert cells[0] == 0 and cells[-1] == b.n_dir_entries and (np.diff(cells.astype(np.int64)) >= 0).all()
    assert entries.max() < b.n_dir_cands
    import ctypes as C
    size = C.sizeof(a.dir_cands.contents) * a.n_dir_cands
    assert C.string_at(a.dir_cands, size) == C.string_at(b.dir_cands, size)
    small, _ = rt.exercise("cover", 64, 36)
    s = small.flatten().desc.contents
    assert s.n_dir_cands < 64 and (s.n_dir_cands == 0 or bool(s.dir_cell_start))


def _world_leaf_box(d, shape):
    for k in range(d.n_bvh_nodes):
        nd = d.bvh_nodes[k]
        if nd.shape == shape:
            return np.array(list(nd.bmin)), np.array(list(nd.bmax))
    raise AssertionError("no leaf names shape %d" % shape)


def test_csg_box_is_the_operand_whose_absence_empties_the_node():  # compound.rs:142-181
    # difference: nothing passes the filter without left entries -> the left operand's box; intersection: the smaller
    # operand's; union: both.  (Boxes are padded by 1e-4 of their diagonal and rounded outwards to f32.)
    small = Primitive.new_cube().with_transform(Matrix.scaling(0.5, 0.5, 0.5))
    big = Primitive.new_sphere().with_transform(Matrix.chain(Matrix.translation(3, 0, 0), Matrix.scaling(2, 2, 2)))
    other = Primitive.new_sphere().with_transform(Matrix.translation(0, 9, 0))  # (a second top-level shape: a hierarchy is built)
    for op, lo, hi in ((Operation.Difference, (-0.5,) * 3, (0.5,) * 3),
                       (Operation.Intersection, (-0.5,) * 3, (0.5,) * 3),
                       (Operation.Union, (-0.5, -2, -2), (5, 2, 2))):
        for left, right in ((small, big), (big, small)):
            if op == Operation.Difference and left is big:
                lo_, hi_ = (1, -2, -2), (5, 2, 2)
            else:
                lo_, hi_ = lo, hi
            flat = World([rt.Compound.new_csg(op, left, right), other], []).flatten()
            d = flat.desc.contents
            assert d.world_bvh_root >= 0
            bmin, bmax = _world_leaf_box(d, 0)
            assert (bmin <= np.array(lo_)).all() and (bmin >= np.array(lo_) - 0.01).all()
            assert (bmax >= np.array(hi_)).all() and (bmax <= np.array(hi_) + 0.01).all()
            assert create(flat.desc) in (N.RT_OK, N.RT_ERR_NO_DEVICE)


def test_csg_box_keeps_both_operands_when_left_hits_are_not_the_left_operand():  # compound.rs:154, shape.rs:18-23
    # l_hit is structural equality: a copy of the left sphere inside the right operand counts as a LEFT hit (its entries
    # pass a difference on a ray that misses the left sphere), a plane or an unbounded cylinder in the left operand (never
    # equal to itself, Q7) as a RIGHT one (its entries toggle in_r).  Such nodes keep the union of both operands' boxes.
    ball = Primitive.new_sphere()
    other = Primitive.new_sphere().with_transform(Matrix.translation(0, 9, 0))
    moved = rt.Compound.new_group([Primitive.new_sphere()]).with_transform(Matrix.translation(5, 0, 0))
    d = World([rt.Compound.new_csg(Operation.Difference, ball, moved), other], []).flatten().desc.contents
    bmin, bmax = _world_leaf_box(d, 0)
    assert bmin[0] <= -1 and bmax[0] >= 6, (bmin, bmax)
    # (an intersection may still take the RIGHT operand's box here: a line that misses it meets left entries only, all of
    # them left hits, and nothing toggles in_r)
    d = World([rt.Compound.new_csg(Operation.Intersection, ball, moved), other], []).flatten().desc.contents
    bmin, bmax = _world_leaf_box(d, 0)
    assert bmin[0] >= 3.9 and bmax[0] >= 6, (bmin, bmax)
    # a cylinder without limits in the left operand: the (smaller) right operand's box does not bound an intersection
    left = rt.Compound.new_group([Primitive.new_sphere().with_transform(Matrix.scaling(3, 3, 3)),
                                  Primitive.new_cylinder(rt.Conic(-np.inf, 2.0, False))])
    cube = Primitive.new_cube().with_transform(Matrix.chain(Matrix.translation(2.5, 0, 0), Matrix.scaling(0.5, 0.5, 0.5)))
    d = World([rt.Compound.new_csg(Operation.Intersection, left, cube), other], []).flatten().desc.contents
    found = [k for k in range(d.n_bvh_nodes) if d.bvh_nodes[k].shape == 0]
    assert found and not np.isfinite(list(d.bvh_nodes[found[0]].bmin)).all()  # unbounded: always visited
    # the same node with a bounded left operand keeps the smaller operand's box
    left2 = rt.Compound.new_group([Primitive.new_sphere().with_transform(Matrix.scaling(3, 3, 3))])
    d = World([rt.Compound.new_csg(Operation.Intersection, left2, cube), other], []).flatten().desc.contents
    bmin, bmax = _world_leaf_box(d, 0)
    assert bmin[0] >= 1.9 and bmax[0] <= 3.1


def test_group_inside_csg_gets_a_hierarchy_only_without_conics():
    # a cylinder or cone can return a hit outside its box (primitive.rs:486-490); the direction index makes up for
    # that for whole CSG nodes, not inside one
    def wedges(with_cylinder):
        kids = [Primitive.new_cube().with_transform(Matrix.translation(3.0 * i, 0, 0)) for i in range(9)]
        if with_cylinder:
            kids.append(Primitive.new_cylinder(Conic(-1.0, 1.0, True)))
        return rt.Compound.new_group(kids)
    for with_cylinder in (False, True):
        csg = rt.Compound.new_csg(Operation.Intersection, Primitive.new_sphere(), wedges(with_cylinder))
        flat = World([csg, Primitive.new_sphere().with_transform(Matrix.translation(0, 9, 0))], []).flatten()
        d = flat.desc.contents
        assert (d.groups[0].bvh_root >= 0) == (not with_cylinder)
        assert create(flat.desc) in (N.RT_OK, N.RT_ERR_NO_DEVICE)
    # ... and the library refuses tables that break the rule: graft the tree onto the group that holds the cylinder
    ok = World([rt.Compound.new_csg(Operation.Intersection, Primitive.new_sphere(), wedges(False)),
                Primitive.new_sphere().with_transform(Matrix.translation(0, 9, 0))], []).flatten()
    d = ok.desc.contents
    group = next(i for i in range(d.n_shapes) if (d.shapes[i].kind_flags & 0xff) == N.RT_GROUP)
    victim = d.shapes[group + 1]
    victim.kind_flags = (victim.kind_flags & ~0xff) | N.RT_CYLINDER
    victim.param = 0
    assert d.n_conics == 0
    conic = (N.RtConic * 1)(N.RtConic(-1.0, 1.0))
    d.conics, d.n_conics = conic, 1
    assert create(ok.desc) == N.RT_ERR_INVALID_SCENE


def test_hierarchy_does_not_depend_on_the_builder_threads(monkeypatch):
    # the upper levels of a large hierarchy are built by threads of their own (disjoint ranges of items and nodes)
    world, _ = rt.synthetic(64, 36, 6000)
    fa = world.flatten()
    a = fa.desc.contents
    assert a.n_bvh_nodes > 8000
    nodes_a = C.string_at(a.bvh_nodes, a.n_bvh_nodes * C.sizeof(a.bvh_nodes._type_))
    monkeypatch.setenv("RT_FLATTEN_THREADS", "1")
    fb = world.flatten()
    b = fb.desc.contents
    assert b.n_bvh_nodes == a.n_bvh_nodes and b.world_bvh_root == a.world_bvh_root
    assert C.string_at(b.bvh_nodes, b.n_bvh_nodes * C.sizeof(b.bvh_nodes._type_)) == nodes_a


def test_noise_permutation_table_in_the_kernels_is_the_derived_one():
    """noise::SuperSimplex::new() = seed 0: the oracle derives PermutationTable::new(0) from rand_xorshift 0.2.0 and
    rand 0.7.3's shuffle (and pattern.png pins the result); the kernels carry the table as a constant."""
    import os, re
    src = open(os.path.join(os.path.dirname(__file__), "..", "raytracer_amd", "csrc", "rt_device.h")).read()
    body = re.search(r"SS_PERM\[256\] = \{([^}]*)\}", src).group(1)
    table = [int(v) for v in body.replace("\n", " ").split(",")]
    derived = (C.c_uint8 * 256)()
    orc.lib.orc_noise_permutation(derived)
    assert table == list(derived) and sorted(table) == list(range(256))
