#!/bin/bash
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -4 || exit 1
for w in teapot hexagon; do for h in nested global; do
  line=$(timeout -k 10 200 python bench.py --workload $w --hierarchy $h --steps 10 --warmup 3 --wavefront 0 --no-cpu-baseline 2>/dev/null) || { echo "$h $m FAILED"; exit 1; }
  echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $h', d['ms_per_step'], d['roofline'].get('kernel','')[:70])"
done; done
for m in 0 1; do
line=$(timeout -k 10 200 python bench.py --steps 3 --warmup 1 --wavefront $m --no-cpu-baseline 2>/dev/null) || { echo "synthetic mode $m FAILED"; exit 1; }
echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('synthetic mode $m', d['ms_per_step'])"
done
