#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_create
mkdir -p $out
export PYTHONPATH=$root
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/gpu_create_prof.py > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
grep "flatten" $out/trace.log
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {float(r['Percentage']):6.2f}")
PY
