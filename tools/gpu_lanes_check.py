"""Would two independent pipelines over halves of a small share overlap?  One scene rendering part 0 of N against
two (four) scenes of the same world rendering parts of 2N (4N) concurrently on streams of their own."""
import sys, time
import torch
import raytracer_amd as rt
from raytracer_amd import tiling
W, H = 4096, 2160
world, cam = rt.synthetic(W, H, 10000)
cd = cam.desc()
flat = world.flatten()
scenes = [rt.DeviceScene(flat) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
def run(n, lanes, k=10):
    parts = n * lanes
    rows = [tiling.n_local_rows(H, p, parts, 16) for p in range(lanes)]
    bufs = [torch.zeros((r, W, 3), dtype=torch.float64, device="cuda") for r in rows]
    def frame():
        for l in range(lanes):
            scenes[l].render_rows(cd, 5, l, parts, 16, rows[l], bufs[l].data_ptr(), streams[l].cuda_stream)
    for _ in range(2):
        frame()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        frame()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3
for n in (1, 2, 4, 8):
    print(f"N={n}: one pipeline {run(n, 1):.2f} ms, two {run(n, 2):.2f} ms, four {run(n, 4):.2f} ms", flush=True)
