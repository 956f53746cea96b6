#!/usr/bin/env python3
"""One line per kernel from the counter passes of tools/pmc_memory.sh.  usage: tools/pmc_table.py gpurun_out/pmcmem_<tag>"""
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", row["Kernel_Name"])
        k = m.group(1) if m else row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
print(f"{'kernel':36s} {'waves':>8s} {'lanes':>6s} {'wait':>6s} {'valu':>6s} {'TA':>6s} {'valu/vmem':>9s} {'lines/rd':>8s} {'L1miss':>6s} {'L2hit':>6s} {'GUI Mcyc':>9s}")
for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if a.get("SQ_WAVES", 0) < 1000:
        continue
    d = lambda x, y: a[x] / a[y] if a.get(y) else float("nan")
    ta = a.get("TA_TA_BUSY_sum", 0) / 256 / (a["GRBM_GUI_ACTIVE"] / 8) if a.get("GRBM_GUI_ACTIVE") else float("nan")
    print(f"{k[:36]:36s} {a['SQ_WAVES']:8.0f} {d('SQ_THREAD_CYCLES_VALU', 'SQ_ACTIVE_INST_VALU') / 64:6.3f} {d('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'):6.3f} "
          f"{d('SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES'):6.3f} {ta:6.3f} {d('SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD'):9.1f} "
          f"{d('TCP_TOTAL_CACHE_ACCESSES_sum', 'SQ_INSTS_VMEM_RD'):8.1f} {d('TCP_TCC_READ_REQ_sum', 'TCP_TOTAL_CACHE_ACCESSES_sum'):6.3f} {d('TCC_HIT_sum', 'TCC_REQ_sum'):6.3f} {a.get('GRBM_GUI_ACTIVE', 0) / 8e6:9.2f}")
