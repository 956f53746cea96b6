#!/bin/bash
# usage: tools/workloads_bench.sh "<workloads>" [extra bench args]   (run on the GPU box from the repo root)
w=$1; shift
for x in $w; do
  timeout -k 10 300 python bench.py --workload $x --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:28], 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'rays', d['config']['rays_per_frame'], 'TFLOP/s', d['roofline']['achieved'], 'frac_no_fma', d['roofline']['frac_no_fma'])" || exit 1
done
