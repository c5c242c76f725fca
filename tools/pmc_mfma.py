#!/usr/bin/env python
"""matrix-pipe occupancy per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
pmc_mfma.py DIR STEPS.   busy fraction = sum MFMA_BUSY / (sum GUI_ACTIVE / 8 XCDs x 1024 SIMDs)  (GRBM_GUI_ACTIVE is reported as the sum over
the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over all SIMDs: MI355X_MICROARCH.md "rocprofv3 PMC slots")"""
import csv, glob, os, re, sys
from collections import defaultdict

d, steps = sys.argv[1], float(sys.argv[2])
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[name] += 1
if steps <= 0:
    steps = max(1.0, cnt.get("sgd_multi_kernel", 6) / 6.0)
print("| kernel | launches/step | MFMA-busy | MFMA instructions/step (M) | GUI_ACTIVE/8 per launch (k cycles) |\n|---|---|---|---|---|")
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0:
        continue
    rows.append((c["SQ_VALU_MFMA_BUSY_CYCLES"], k, cnt[k] / steps, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0), c.get("SQ_INSTS_MFMA", 0.0) / steps / 1e6,
                 gui / 8.0 / max(cnt[k], 1) / 1e3))
for _, k, n, frac, mf, g in sorted(rows, reverse=True)[:16]:
    print("| `%s` | %.1f | %.3f | %.2f | %.1f |" % (k[:80], n, frac, mf, g))
