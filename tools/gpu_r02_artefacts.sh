#!/bin/bash
# Round-2 artefacts of the default bench workload (10k scene at 4096x2160): bench line, kernel-trace summary, counter passes.
# usage (GPU box, repo root): tools/gpu_r02_artefacts.sh <tag>
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || { tail -5 gpurun_out/${tag}_bench_default.err; exit 1; }
bash tools/gpu_trace_stats.sh ${tag}_synthetic4k > gpurun_out/${tag}_trace_table.txt 2>&1 || { tail gpurun_out/${tag}_trace_table.txt; exit 1; }
bash tools/pmc_memory.sh ${tag}_synthetic4k > gpurun_out/${tag}_pmc_raw.txt 2>&1 || { tail gpurun_out/${tag}_pmc_raw.txt; exit 1; }
python3 tools/pmc_table.py gpurun_out/pmcmem_${tag}_synthetic4k > gpurun_out/${tag}_pmc_table.txt
# HBM traffic per frame: FETCH_SIZE (KB, x2 on gfx950) and WRITE_SIZE (KB), summed over one frame's dispatches
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/pmcmem_${tag}_synthetic4k/pmc_write -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_pmc_write.log 2>&1
cd $root
python3 - "$root/gpurun_out/pmcmem_${tag}_synthetic4k" <<'PY' > gpurun_out/${tag}_traffic.txt
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and ("wf2_" in row["Kernel_Name"] or "wf_combine" in row["Kernel_Name"]) and "Lb1E" not in row["Kernel_Name"] and "<true" not in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
frames = 4.0  # --steps 3 --warmup 1 of the non-counting kernels (the counting launch uses the <true> variants)
print("FETCH_SIZE KB per frame (as counted)", tot["FETCH_SIZE"] / frames, " x2 (gfx950) =", 2 * tot["FETCH_SIZE"] / frames)
print("WRITE_SIZE KB per frame", tot["WRITE_SIZE"] / frames)
print("HBM bytes per frame", (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / frames * 1024)
PY
cat gpurun_out/${tag}_bench_default.json; cat gpurun_out/${tag}_traffic.txt; head -12 gpurun_out/${tag}_pmc_table.txt
