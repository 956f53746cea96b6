#!/bin/bash
# usage (GPU box, repo root): tools/pmc_icache.sh <tag> : instruction-cache counters of the bench command, per kernel
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcic_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/pass -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pass.log 2>&1 || { tail -5 $out/pass.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/pass/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"\w+_kernel(<[^>]*>)?", row["Kernel_Name"])
        k = m.group(0) if m else row["Kernel_Name"][:50]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
print(f"{'kernel':40s} {'icache req':>12s} {'miss share':>10s} {'ifetch':>12s} {'wait inst / wave cycles':>24s}")
for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if a.get("SQ_WAVE_CYCLES", 0) < 1e8: continue
    req = a.get("SQC_ICACHE_REQ", 0)
    print(f"{k:40s} {req:12.4g} {a.get('SQC_ICACHE_MISSES', 0) / req if req else float('nan'):10.4f} {a.get('SQ_IFETCH', 0):12.4g} {a.get('SQ_WAIT_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:24.3f}")
PY
