#!/bin/bash
# One bench line per workload at 4096x2160 (+ the 10k scene at 7680x4320) -> gpurun_out/<tag>_all_workloads.txt
tag=${1:-all}
out=gpurun_out/${tag}_all_workloads.txt
mkdir -p gpurun_out
: > $out
for w in plane scene pattern hexagon cover room csg teapot; do
  timeout -k 10 120 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline >> $out 2>>gpurun_out/${tag}_all_workloads.err || exit 1
done
timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline >> $out 2>>gpurun_out/${tag}_all_workloads.err || exit 1
timeout -k 10 200 python bench.py --width 7680 --height 4320 --steps 5 --warmup 1 --no-cpu-baseline >> $out 2>>gpurun_out/${tag}_all_workloads.err || exit 1
python - <<PY
import json
for l in open("$out"):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel"))
PY
