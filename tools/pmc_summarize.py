#!/usr/bin/env python3
"""Summarises the counter CSVs written by tools/pmc_conv.sh: mean per-dispatch value of each counter for the
dominant (longest) kernel."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not acc:
    print("no counter files under", root); sys.exit(0)
# dominant kernel = the one with the largest SQ_WAVE_CYCLES (fallback: most rows)
def weight(k):
    v = acc[k].get("SQ_WAVE_CYCLES") or acc[k].get("GRBM_GUI_ACTIVE") or [0]
    return sum(v) / len(v)
for k in sorted(acc, key=weight, reverse=True)[:2]:
    print("==", k[:110])
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-34s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
