#!/bin/bash
# usage: tools/variant_bench.sh "<variants>" "<workloads>"   (run on the GPU box from the repo root)
set -e
cp raytracer_amd/lib/librt_hip.so /tmp/librt_hip_orig.so
for v in $1; do
  cp build/variants/librt_hip_$v.so raytracer_amd/lib/librt_hip.so
  for w in $2; do
    timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'][:28], 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'frac_no_fma', d['roofline']['frac_no_fma'])"
  done
done
cp /tmp/librt_hip_orig.so raytracer_amd/lib/librt_hip.so
