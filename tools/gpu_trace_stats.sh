#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_trace_stats.sh <tag> [bench args...]
# kernel-trace summary of bench.py -> gpurun_out/<tag>_kernel_stats.csv
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {float(r['Percentage']):6.2f}")
PY
tail -1 $out/trace.log | cut -c1-300
