#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel-trace summary of bench.py for each named workload.
# usage: tools/trace_all.sh "<workloads>"
set -e
cd /tmp && export TMPDIR=/tmp
for w in $1; do
  out=$GRAFT_REPO_ROOT/gpurun_out/trace_$w
  mkdir -p $out
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/err.log
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$w.csv
  cp $out/bench.json $GRAFT_REPO_ROOT/gpurun_out/bench_under_trace_$w.json
done
