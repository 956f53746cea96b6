#!/bin/bash
# device-built direction cells: the test, then the bench at several resolutions, with scene-creation time
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "direction_cells or conic_parallel or generation_per_launch" 2>&1 | tail -5 || exit 1
for r in 256 512 1024 2048; do
  RT_HIP_DIR_RES=$r timeout -k 10 200 python - <<PY || exit 1
import time, json, subprocess, sys, os
import torch, raytracer_amd as rt
torch.zeros(1, device="cuda")
w, cam = rt.synthetic(4096, 2160, 10000)
t = time.time(); f = w.flatten(); t1 = time.time(); s = rt.DeviceScene(f); torch.cuda.synchronize(); t2 = time.time()
t = time.time(); f = w.flatten(); t1 = time.time(); s = rt.DeviceScene(f); torch.cuda.synchronize(); t2 = time.time()
print("res $r flatten %.1f ms scene_create %.1f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3), flush=True)
PY
  RT_HIP_DIR_RES=$r timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print('res $r ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
done
