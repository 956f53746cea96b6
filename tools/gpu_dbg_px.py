"""One row of the 8K frame of the 10k scene through modes 0 and 2 (debugging a pixel)."""
import numpy as np, torch, raytracer_amd as rt, os
W, H, Y, X = 7680, 4320, 83, 2187
world, cam = rt.synthetic(W, H, 10000)
scene = rt.DeviceScene(world.flatten())
cd = cam.desc()
res = {}
for mode in (0, 2):
    scene.set_wavefront(mode)
    out = torch.full((1, W, 3), -1.0, dtype=torch.float64, device="cuda")
    scene.render_rows(cd, 5, Y, H, 1, 1, out.data_ptr())
    torch.cuda.synchronize(); scene.status()
    res[mode] = out.cpu().numpy()[0]
d = (res[0] != res[2]).any(axis=1)
print(os.environ.get("RT_HIP_WF2_SLACK"), os.environ.get("RT_HIP_WF2_SEG_NEAR"), "differing x:", np.nonzero(d)[0][:10], res[0][X], res[2][X])
