#!/bin/bash
# usage (GPU box, repo root): tools/gpu_share_timeline.sh <tag>  -> gpurun_out/<tag>_share8_timeline.txt
# per-dispatch timeline of the LAST frame tools/gpu_shard_time.py renders: rank 0's share of the 4K frame dealt to 8 ranks
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$root timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/tools/gpu_shard_time.py > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $root/gpurun_out/${tag}_share8_timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "wf2_packet_kernel<false, 0>" in r["Kernel_Name"] or "wf2_trace_kernel<false, 0>" in r["Kernel_Name"]]
first = starts[-1] if starts else 0
t0 = int(rows[first]["Start_Timestamp"])
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
    return n.replace("rtk::", "")
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}")
PY
tail -2 $out/trace.log
