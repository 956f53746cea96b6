#!/bin/bash
for lib in "" variants/lds4 variants/lds2; do
for w in teapot; do
  line=$(RT_HIP_LIB_DIR=$lib timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null) || { echo "$w FAILED"; exit 1; }
  echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=$lib $w', d['ms_per_step'], d['roofline'].get('kernel','')[:70])"
done; done
