#!/usr/bin/env python3
"""Register / spill / scratch / LDS table of every kernel of one .hip file, from hipcc's resource-usage remarks.
usage: tools/kernel_resources.py raytracer_amd/csrc/rt_wf2.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?: (Function Name|[A-Za-z ]+(?:\[[^\]]*\])?): (.*?)( \[-Rpass|$)", line)
    if not m:
        m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*([A-Za-z ]+(?: \[[^\]]*\])?): (\S+)", line)
        if not m:
            continue
    key, val = m.group(1).strip(), m.group(2).strip()
    if key in ("Function Name", "Name"):
        cur = {"name": val}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s} {'SGPR':>5s}")
for r in rows:
    name = subprocess.run(["c++filt", "-p", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
    get = lambda k: next((v for kk, v in r.items() if kk.startswith(k)), "?")
    print(f"{name[:70]:70s} {get('VGPRs:') if False else r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {get('VGPRs Spill'):>6s} {get('SGPRs Spill'):>6s} "
          f"{get('ScratchSize'):>8s} {get('LDS Size'):>7s} {get('Occupancy'):>4s} {get('TotalSGPRs'):>5s}")
