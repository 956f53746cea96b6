#!/bin/bash
# per-kernel durations of one workload (run on the GPU box from the repo root): tools/wf_trace.sh <workload>
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/wftrace_$1
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --steps 2 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = [r for r in rows][-40:]
for r in last:
    print(r['Kernel_Name'][:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us grid', r.get('Grid_Size_X', r.get('Grid_Size')))
PY
