"""8K debugging: one band subset of the 10k scene through every mode against the oracle."""
import numpy as np, torch, raytracer_amd as rt
from raytracer_amd import tiling
from oracle import orc
import sys
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
world, cam = rt.synthetic(W, H, 10000)
scene = rt.DeviceScene(world.flatten())
cd = cam.desc()
band, n_sub, part = 16, 32, 5
sub_rows = tiling.n_local_rows(H, part, n_sub, band)
rows = tiling.image_rows(H, part, n_sub, band)
ow = orc.OracleWorld(world.describe())
ref = np.zeros((H, W, 3))
for y in rows[::16][:3]:
    ow.render(cam, rows=(int(y), int(y) + 1, 1), out=ref)
res = {}
for mode in (0, 1, 2):
    scene.set_wavefront(mode)
    out = torch.full((sub_rows, W, 3), -1.0, dtype=torch.float64, device="cuda")
    scene.render_rows(cd, 5, part, n_sub, band, sub_rows, out.data_ptr())
    torch.cuda.synchronize(); scene.status()
    res[mode] = out.cpu().numpy()
    for j, y in enumerate(rows[::16][:3]):
        d = np.abs(res[mode][j * 16] - ref[y]).max()
        print("mode", mode, "row", y, "max |gpu - oracle|", d)
full = torch.full((H, W, 3), -1.0, dtype=torch.float64, device="cuda")
scene.set_wavefront(2)
scene.render_rows(cd, 5, 0, 1, H, H, full.data_ptr())
torch.cuda.synchronize(); scene.status()
f = full.cpu().numpy()
for j, y in enumerate(rows[::16][:3]):
    print("full frame mode 2 row", y, np.abs(f[y] - ref[y]).max())
print("modes equal:", np.array_equal(res[0], res[1]), np.array_equal(res[0], res[2]))
print("full vs sub (mode2):", np.array_equal(f[rows], res[2]))
bad = np.argwhere((f[rows] != res[2]).any(axis=(1, 2)))
print("rows differing", bad[:10].ravel(), len(bad))
diff = (res[0] != res[2]).any(axis=2)
ys, xs = np.nonzero(diff)
print("pixels differing mode0 vs mode2:", len(ys), "of", diff.size)
for j in range(min(12, len(ys))):
    y, x = ys[j], xs[j]
    print("  local row", y, "image row", rows[y], "x", x, "mode0", res[0][y, x], "mode2", res[2][y, x])
if len(ys):
    # which is right? oracle on the first differing image row
    y0 = int(rows[ys[0]])
    ow.render(cam, rows=(y0, y0 + 1, 1), out=ref)
    x0 = xs[0]
    print("oracle", ref[y0, x0])
    import collections
    print("rows with differences:", collections.Counter(rows[ys]).most_common(8))
