#!/bin/bash
# usage: tools/gpu_small_sweep.sh <tag> "<ENV=..>" ... : ms per frame of the small compound workloads at 4096x2160 per environment setting
tag=$1; shift
mkdir -p gpurun_out
: > gpurun_out/${tag}_small.txt
for setting in "$@"; do
  for w in ${SMALL_WORKLOADS:-csg cover room hexagon}; do
    line=$(env $setting timeout -k 10 120 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null) || { echo "$setting $w FAILED" | tee -a gpurun_out/${tag}_small.txt; exit 1; }
    ms=$(echo "$line" | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$setting $w ms_per_step $ms" | tee -a gpurun_out/${tag}_small.txt
  done
done
