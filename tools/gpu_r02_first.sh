#!/bin/bash
# first GPU pass of round 2: the wavefront tests (incl. the kernel pipeline), then the 10k scene in modes 1 and 2
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "generation or pipeline or synthetic" > gpurun_out/r02_first_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r02_first_tests.log
[ $rc -ne 0 ] && exit $rc
for mode in 1 2; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --wavefront $mode --no-cpu-baseline --print-stats > gpurun_out/r02_first_bench_m$mode.json 2> gpurun_out/r02_first_bench_m$mode.err || exit $?
  cat gpurun_out/r02_first_bench_m$mode.json
done
