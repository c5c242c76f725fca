#!/usr/bin/env python
"""mean counter value per (kernel, counter) from rocprofv3 --pmc csv output dirs: pmc_summary.py DIR [DIR ...]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "conv" in k or "wgrad" in k:
        print("%-72s %-30s %16.0f  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
