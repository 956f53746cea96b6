#!/bin/bash
# usage: tools/gpu_sweep.sh <tag> "<ENV=..;ENV=..>" ... : one bench line per environment setting (mode 2, 10k scene 4K)
tag=$1; shift
mkdir -p gpurun_out
: > gpurun_out/${tag}_sweep.txt
for setting in "$@"; do
  line=$(env $setting timeout -k 10 200 python bench.py --steps 5 --warmup 2 --wavefront 2 --no-cpu-baseline $SWEEP_ARGS 2>/dev/null) || { echo "$setting FAILED" | tee -a gpurun_out/${tag}_sweep.txt; exit 1; }
  ms=$(echo "$line" | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$setting ms_per_step $ms" | tee -a gpurun_out/${tag}_sweep.txt
done
