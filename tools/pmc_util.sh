#!/bin/bash
# lane-utilisation PMC pass for several workloads (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/util_$w; mkdir -p $out
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1
done
