#!/bin/bash
# usage: tools/gpu_modes.sh <tag> <workload>... : ms per frame of each workload under wavefront modes -1, 0, 2 (4096x2160)
tag=$1; shift
mkdir -p gpurun_out
: > gpurun_out/${tag}_modes.txt
for w in "$@"; do
  for m in -1 0 2; do
    line=$(timeout -k 10 200 python bench.py --workload $w --steps 5 --warmup 2 --wavefront $m --no-cpu-baseline 2>/dev/null) || { echo "$w $m FAILED" | tee -a gpurun_out/${tag}_modes.txt; exit 1; }
    echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w mode $m', d['ms_per_step'], d['roofline'].get('kernel','')[:60])" | tee -a gpurun_out/${tag}_modes.txt
  done
done
