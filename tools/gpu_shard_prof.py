import sys, time
import torch
import raytracer_amd as rt
from raytracer_amd import tiling
W, H = 4096, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
world, cam = rt.synthetic(W, H, 10000)
scene = rt.DeviceScene(world.flatten())
cd = cam.desc()
stream = torch.cuda.current_stream().cuda_stream
rows = tiling.n_local_rows(H, 0, n, 16)
buf = torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda")
for _ in range(6):
    scene.render_rows(cd, 5, 0, n, 16, rows, buf.data_ptr(), stream)
torch.cuda.synchronize()
