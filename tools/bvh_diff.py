import sys, numpy as np, torch
sys.path.insert(0, '.')
import raytracer_amd as rt
from raytracer_amd import _native as N
world, cam = rt.synthetic(192, 108, 3000)
cd = cam.desc()
frames = []
flats = []
for bvh in (False, True):
    flat = world.flatten(build_bvh=bvh); flats.append(flat)
    scene = rt.DeviceScene(flat)
    buf = torch.zeros((108, 192, 3), dtype=torch.float64, device="cuda")
    scene.render_rows(cd, 5, 0, 1, 108, 108, buf.data_ptr())
    torch.cuda.synchronize(); scene.status()
    frames.append(buf.cpu().numpy())
d = (frames[0] != frames[1]).any(axis=2)
print("differing pixels:", d.sum(), "of", d.size)
ys, xs = np.nonzero(d)
from oracle import orc
ow = orc.OracleWorld(world.describe())
desc = flats[0].desc.contents
for y, x in list(zip(ys, xs))[:12]:
    ray = orc.ray_for_pixel(192, 108, cam.field_of_view, cam.transform, int(x), int(y))
    p = ow.probe(ray)[0]
    kind = desc.shapes[int(p[2])].kind_flags & 0xff
    print(y, x, "oracle hit t", p[1], "shape", int(p[2]), "kind", kind, frames[0][y, x], frames[1][y, x])
