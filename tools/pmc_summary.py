#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csv files: per kernel, mean of each counter per dispatch.
Usage: tools/pmc_summary.py <dir-or-csv> [...]   (looks for *counter_collection.csv)"""
import csv, glob, os, sys, collections
files = []
for a in sys.argv[1:]:
    files += [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0]
        acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
names = sorted({c for k in acc for c in acc[k]})
print("| kernel | dispatches | " + " | ".join(names) + " |")
print("|---|---|" + "|".join(["---"] * len(names)) + "|")
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    print("| %s | %d | " % (k[:48], n) + " | ".join(("%.4g" % (sum(acc[k][c]) / len(acc[k][c]))) if acc[k][c] else "-" for c in names) + " |")
