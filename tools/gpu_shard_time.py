"""Time of ONE rank's share of the default workload when the frame is dealt to N ranks (what strong scaling can reach on
N GPUs if the gather is hidden): part 0 of N, 16-row bands, on this one GPU."""
import sys, time
import torch
import raytracer_amd as rt
from raytracer_amd import tiling

W, H = 4096, 2160
world, cam = rt.synthetic(W, H, 10000)
scene = rt.DeviceScene(world.flatten())
cd = cam.desc()
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    rows = tiling.n_local_rows(H, 0, n, 16)
    buf = torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda")
    for _ in range(2):
        scene.render_rows(cd, 5, 0, n, 16, rows, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    k = 10
    for _ in range(k):
        scene.render_rows(cd, 5, 0, n, 16, rows, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / k * 1e3
    whole = ms if n == 1 else whole
    print(f"N={n}: part 0 has {rows} rows, {ms:.2f} ms per frame share; an even split of N=1's time {whole / n:.2f}", flush=True)
