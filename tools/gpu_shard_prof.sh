#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_shard
rm -rf $out; mkdir -p $out
export PYTHONPATH=$root
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/gpu_shard_prof.py ${1:-8} > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
t=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms_per_frame {float(r['TotalDurationNs'])/1e6/6:7.3f}")
print("sum of kernel time per frame (ms):", sum(float(r['TotalDurationNs']) for r in rows if 'wf' in r['Name'] or 'fill' in r['Name'].lower()) / 1e6 / 6)
tr = list(csv.DictReader(open(sys.argv[2])))
tr = [r for r in tr if 'wf' in r['Kernel_Name']]
tr.sort(key=lambda r: int(r['Start_Timestamp']))
# last frame: the final 1/6 of the launches
k = len(tr) // 6
last = tr[-k:]
span = (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6
print(f"last frame: {k} launches, span {span:.3f} ms, kernel time {busy:.3f} ms")
gaps = sorted(((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3, a['Kernel_Name'][:40], b['Kernel_Name'][:40]) for a, b in zip(last, last[1:]))
print("largest gaps (us):", gaps[-6:])
print("mean gap (us):", sum(g[0] for g in gaps) / len(gaps))
PY
