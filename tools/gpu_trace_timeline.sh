#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_trace_timeline.sh <tag> [bench args...]
# per-dispatch kernel timeline of ONE timed frame of bench.py -> gpurun_out/<tag>_timeline.txt (start offset, duration, kernel, stream)
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $root/gpurun_out/${tag}_timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last PRIMARY trace launch of the non-counting variant on
starts = [i for i, r in enumerate(rows) if "wf2_trace_kernel<false, 0>" in r["Kernel_Name"] or "wf2_packet_kernel<false, 0>" in r["Kernel_Name"]]
first = starts[-1] if starts else 0
t0 = int(rows[first]["Start_Timestamp"])
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("rtk::", "").replace("(anonymous namespace)::", "")
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}")
PY
tail -1 $out/trace.log | cut -c1-200
