"""Host flatten + rt_hip_scene_create of the 10k scene and of teapot.rs, steady state (the third of three rounds is the one to read).
RT_FLATTEN_TIMING=1 / RT_HIP_CREATE_TIMING=1 in the environment print the stages."""
import time, sys
import torch, raytracer_amd as rt
torch.zeros(1, device="cuda")
for name in ("synthetic", "teapot"):
    w, cam = rt.synthetic(4096, 2160, 10000) if name == "synthetic" else rt.exercise("teapot", 4096, 2160)
    for k in range(3):
        t = time.time(); f = w.flatten(); t1 = time.time(); s = rt.DeviceScene(f); torch.cuda.synchronize(); t2 = time.time()
        print("%s: flatten %.1f ms scene_create %.1f ms" % (name, (t1 - t) * 1e3, (t2 - t1) * 1e3), flush=True)
        s.close()
