#!/bin/bash
# Run on the GPU box from the repo root: memory-side counters of one bench workload, one pass per run.
# usage: tools/pmc_memory.sh <tag> [bench args...]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $@"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mem -- $B > $out/pmc_mem.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_TA_BUSY_sum TA_BUSY_avr --output-format csv -d $out/pmc_tcp -- $B > $out/pmc_tcp.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_tcc -- $B > $out/pmc_tcc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        import re
        m = re.search(r"\w+_kernel(<[^>]*>)?", row["Kernel_Name"])
        k = m.group(0) if m else row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in acc:
    a = acc[k]
    if a.get("SQ_WAVES", 0) < 1000: continue
    print(k)
    for c in sorted(a): print(f"  {c:34s} total {a[c]:.6g}  dispatches {cnt[k][c]}")
    d = lambda x, y: a[x] / a[y] if a.get(y) else float("nan")
    print(f"  -> VALU lane utilisation       {d('SQ_THREAD_CYCLES_VALU', 'SQ_ACTIVE_INST_VALU') / 64:.3f}   (thread cycles / (active VALU cycles x 64))")
    print(f"  -> waves waiting               {d('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'):.3f}   (SQ_WAIT_ANY / SQ_WAVE_CYCLES)")
    print(f"  -> VALU issue share of a wave  {d('SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES'):.3f}   (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)")
    print(f"  -> VALU per SALU instruction   {d('SQ_INSTS_VALU', 'SQ_INSTS_SALU'):.2f}")
    print(f"  -> VALU per vector memory read {d('SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD'):.1f}")
    print(f"  -> L1 lines per read instr     {d('TCP_TOTAL_CACHE_ACCESSES_sum', 'SQ_INSTS_VMEM_RD'):.1f}")
    print(f"  -> vector L1 miss share        {d('TCP_TCC_READ_REQ_sum', 'TCP_TOTAL_CACHE_ACCESSES_sum'):.3f}")
    print(f"  -> L2 hit share                {d('TCC_HIT_sum', 'TCC_REQ_sum'):.3f}")
    print(f"  -> TA busy share               {a.get('TA_TA_BUSY_sum', 0) / 256 / (a.get('GRBM_GUI_ACTIVE', 0) / 8) if a.get('GRBM_GUI_ACTIVE') else float('nan'):.3f}   (TA_TA_BUSY_sum / 256 TAs / (GRBM_GUI_ACTIVE / 8 XCDs))")
PY
