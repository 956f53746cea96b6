#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv files: per-kernel mean of each counter per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        import re
        m = re.search(r"render_kernel<[a-z0-9, ]+>", name)
        k = m.group(0) if m else name[:40]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "render_kernel<false" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} mean/dispatch {sum(v)/len(v):.6g}  (n={len(v)})")
