"""The flat record tables of include/rt_hip.h (RtSceneDesc) turned back into the object-tree description the oracle
consumes (the format of World.describe(), host/rt_scene.cpp:598-685).  TEST INFRASTRUCTURE: the gate of SURVEY.md §7 step 2,
"the oracle rendering from the flattened buffers == the oracle rendering from the object tree", needs the flattener's
output in a form the oracle reads.  The tables keep inverse transforms only, so the forward matrices here are numpy
inverses of them: equal to the originals to rounding, not bit for bit -- the gate compares with a tolerance and counts
the pixels whose discrete decisions flipped."""
import numpy as np

RT_GROUP, RT_CSG = 7, 8
RT_FLAG_HAS_SHADOW, RT_FLAG_CLOSED, RT_FLAG_CSG_OP_SHIFT = 1 << 8, 1 << 9, 10
PAT_SOLID, PAT_PERTURB, PAT_TEST = 0, 1, 8


def _forward(inv12):
    m = np.eye(4)
    m[:3, :] = np.asarray(inv12[:12], dtype=np.float64).reshape(3, 4)
    return np.linalg.inv(m)


def _pattern(d, out, idx):
    p = d.patterns[idx]
    out.append(float(p.kind))
    out.append(1.0)
    out.extend(_forward(p.inv).reshape(-1))
    if p.kind == PAT_SOLID:
        out.extend(p.color[:3])
    elif p.kind == PAT_TEST:
        pass
    elif p.kind == PAT_PERTURB:
        _pattern(d, out, p.a)
    else:
        _pattern(d, out, p.a)
        _pattern(d, out, p.b)


def _material(d, out, idx):
    if idx is None:  # compounds: a material is accepted and ignored by rendering (Q8)
        out.extend([1.0, 1.0, 1.0, 0.1, 0.9, 0.9, 200.0, 0.0, 0.0, 1.0, 0.0])
        return
    m = d.materials[idx]
    out.extend(m.color[:3])
    out.extend([m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency, m.refractive_index])
    out.append(1.0 if m.pattern >= 0 else 0.0)
    if m.pattern >= 0:
        _pattern(d, out, m.pattern)


def _shape(d, out, i):
    s = d.shapes[i]
    kind = s.kind_flags & 0xff
    out.append(float(kind))
    out.append(1.0 if kind >= RT_GROUP or (s.kind_flags & RT_FLAG_HAS_SHADOW) else 0.0)
    out.append(1.0)
    out.extend(_forward(s.inv).reshape(-1))
    if kind < RT_GROUP:
        _material(d, out, s.material)
        if kind in (3, 4):
            c = d.conics[s.param]
            out.extend([c.minimum, c.maximum, 1.0 if s.kind_flags & RT_FLAG_CLOSED else 0.0])
        elif kind in (5, 6):
            t = d.triangles[s.param]
            p1, e1, e2 = np.array(t.p1[:3]), np.array(t.e1[:3]), np.array(t.e2[:3])
            out.extend(p1), out.extend(p1 + e1), out.extend(p1 + e2)
            if kind == 6:
                out.extend(t.n1[:3]), out.extend(t.n2[:3]), out.extend(t.n3[:3])
        return
    _material(d, out, None)
    children = []
    k = i + 1
    while k < s.skip:
        children.append(k)
        k = d.shapes[k].skip
    if kind == RT_CSG:
        assert len(children) == 2
        out.append(float((s.kind_flags >> RT_FLAG_CSG_OP_SHIFT) & 3))
    else:
        assert len(children) == d.groups[s.param].n_children
        out.append(float(len(children)))
    for c in children:
        assert d.shapes[c].parent == i
        _shape(d, out, c)


def description_from_flat(desc):
    """desc: a _native.RtSceneDesc (FlatScene.desc.contents) -> the float64 array OracleWorld() takes."""
    d = desc
    out = [float(d.n_lights)]
    for k in range(d.n_lights):
        out.extend(d.lights[k].position[:3]), out.extend(d.lights[k].intensity[:3])
    top = []
    i = 0
    while i < d.n_shapes:
        assert d.shapes[i].parent == -1
        top.append(i)
        i = d.shapes[i].skip
    out.append(float(len(top)))
    for i in top:
        _shape(d, out, i)
    return np.asarray(out, dtype=np.float64)
