#!/bin/bash
# Round-3 artefacts of the bench workload (10k scene): bench line, kernel-trace summary, counter passes, HBM traffic.
# usage (GPU box, repo root): tools/gpu_r03_artefacts.sh <tag> <4k|8k> [full]
#   4k: 4096x2160 (the default bench line); 8k: 7680x4320.  "full" adds the SQ/TCP/TCC passes of tools/pmc_memory.sh.
set -o pipefail
tag=$1; size=$2; full=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
args=""; [ "$size" = "8k" ] && args="--width 7680 --height 4320"
name=${tag}_synthetic${size}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py $args > gpurun_out/${name}_bench.json 2> gpurun_out/${name}_bench.err || { tail -5 gpurun_out/${name}_bench.err; exit 1; }
bash tools/gpu_trace_stats.sh $name $args > gpurun_out/${name}_trace_table.txt 2>&1 || { tail gpurun_out/${name}_trace_table.txt; exit 1; }
out=$root/gpurun_out/pmcmem_$name
mkdir -p $out
if [ "$full" = "full" ]; then
  bash tools/pmc_memory.sh $name $args > gpurun_out/${name}_pmc_raw.txt 2>&1 || { tail gpurun_out/${name}_pmc_raw.txt; exit 1; }
  python3 tools/pmc_table.py $out > gpurun_out/${name}_pmc_table.txt
else
  cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $out/pmc_fetch.log 2>&1 || { tail -3 $out/pmc_fetch.log; exit 1; }
  cd $root
fi
# HBM traffic per frame: FETCH_SIZE (KB, x2 on gfx950) and WRITE_SIZE (KB), summed over one frame's dispatches
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $out/pmc_write.log 2>&1 || { tail -3 $out/pmc_write.log; exit 1; }
cd $root
python3 - "$out" gpurun_out/${name}_bench.json <<'PY' > gpurun_out/${name}_traffic.txt
import csv, glob, sys, collections, json, re
tot = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and ("wf2_" in k or "wf_combine" in k) and "Lb1E" not in k and "<true" not in k:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            m = re.search(r"\w+_kernel(<[^>]*>)?", k)
            per[m.group(0) if m else k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
frames = 4.0  # --steps 3 --warmup 1 of the non-counting kernels (the counting launch uses the <true> variants)
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rays = line["config"].get("rays_per_frame") or line.get("rays_per_frame")
hbm = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / frames * 1024
print("FETCH_SIZE KB per frame (as counted)", tot["FETCH_SIZE"] / frames, " x2 (gfx950) =", 2 * tot["FETCH_SIZE"] / frames)
print("WRITE_SIZE KB per frame", tot["WRITE_SIZE"] / frames)
print("HBM bytes per frame", hbm)
if rays: print("rays per frame", rays, " HBM bytes per ray", hbm / rays)
print("per kernel, bytes per frame (2 x FETCH + WRITE):")
for k in sorted(per, key=lambda k: -(2 * per[k]["FETCH_SIZE"] + per[k]["WRITE_SIZE"])):
    print(f"  {k:60s} {(2 * per[k]['FETCH_SIZE'] + per[k]['WRITE_SIZE']) / frames * 1024 / 1e9:8.3f} GB  (read {2 * per[k]['FETCH_SIZE'] / frames * 1024 / 1e9:.3f}, written {per[k]['WRITE_SIZE'] / frames * 1024 / 1e9:.3f})")
PY
cat gpurun_out/${name}_bench.json; cat gpurun_out/${name}_traffic.txt; head -14 gpurun_out/${name}_trace_table.txt
