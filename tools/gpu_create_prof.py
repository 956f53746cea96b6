import time, sys
import torch, raytracer_amd as rt
torch.zeros(1, device="cuda")
w, cam = rt.synthetic(4096, 2160, 10000)
for k in range(2):
    t = time.time(); f = w.flatten(); t1 = time.time(); s = rt.DeviceScene(f); torch.cuda.synchronize(); t2 = time.time()
    print("flatten %.1f ms scene_create %.1f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3), flush=True)
    s.close()
