#!/usr/bin/env python
"""HBM-side bytes per kernel from two rocprofv3 --pmc passes over the same command:  pmc_bytes.py FETCH_DIR WRITE_DIR STEPS
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 kB per count by rocprofv3 (checked against r03: 136.3 MB of exact
output writes); on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so reads are DOUBLED (MI355X_MICROARCH.md "HBM").
Per kernel name: launches per step, corrected read + write bytes per step, and -- from the kernel-trace timestamps of the same
runs -- time per step and bytes / time."""
import csv, glob, os, re, sys
from collections import defaultdict

fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
json_out = sys.argv[4] if len(sys.argv) > 4 else None     # {engine: {kernel symbol: HBM bytes per launch}} for bench.py's roofline.traffic
engine = sys.argv[5] if len(sys.argv) > 5 else "f16x3"


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            e = acc[name]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return acc


def times(d):
    acc = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            acc[name] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
    return acc


fe, wr, tm = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE"), times(fetch_dir)
if steps <= 0:  # every train step ends in 6 sgd_multi_kernel launches (two param groups of R101: 64 tensors per launch), eager or replayed
    steps = max(1.0, fe.get("sgd_multi_kernel", [6, 0.0])[0] / 6.0)
print("counter units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kB; reads x2 (gfx950 correction); %g steps in the run (warm-up included)" % steps)
print("| kernel | launches/step | read MB/step (x2) | write MB/step | ms/step (under PMC) | TB/s |\n|---|---|---|---|---|---|")
rows = []
for k in set(fe) | set(wr):
    rd = 2.0 * fe[k][1] * 1024 / 1e6 / steps if k in fe else 0.0
    ww = wr[k][1] * 1024 / 1e6 / steps if k in wr else 0.0
    n = (fe[k][0] if k in fe else wr[k][0]) / steps
    t = tm.get(k, 0.0) / steps
    rows.append((rd + ww, k, n, rd, ww, t))
for tot, k, n, rd, ww, t in sorted(rows, reverse=True)[:24]:
    print("| `%s` | %.1f | %.1f | %.1f | %.3f | %s |" % (k[:80], n, rd, ww, t, ("%.2f" % (tot / 1e3 / t)) if t > 0 else "-"))

# ---- per family against SURVEY 8(d)'s algorithmic bytes of config A (R101 513^2 batch 8, f32 tensors): conv 3 x (in + out) x 4 B + weights
# = 22.55 GB, train-mode BN 4 passes x 4 B x 108 M elements x 8 images = 13.79 GB (bench.py:mixed_roofline computes both)
fam = defaultdict(float)
for tot, k, n, rd, ww, t in rows:
    f = ("conv + weight gradient" if ("conv_x3" in k or "wgrad" in k or "conv_igemm" in k or "rowtap" in k or "conv_wgrad" in k) else
         "BN family" if (k.startswith("bn_") or "colstat" in k or "sum_n" in k) else "other")
    fam[f] += tot
total = sum(fam.values())
print("\nper family (GB / step): " + ", ".join("%s %.1f" % (k, v / 1e3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])) + "; total %.1f GB" % (total / 1e3))
print("against the algorithmic bytes of config A: conv %.2fx of 22.55 GB, BN %.2fx of 13.79 GB, step %.2fx of 36.34 GB"
      % (fam["conv + weight gradient"] / 22550.0, fam["BN family"] / 13790.0, total / 36340.0))
if json_out:
    import json
    per = {}
    for tot, k, n, rd, ww, t in rows:
        if n > 0 and not k.startswith("at::") and not k.startswith("__amd"):
            per[k.replace(", ", ",")] = round(tot * 1e6 / n, 1)
    old = {}
    if os.path.exists(json_out):
        old = json.load(open(json_out))
    old[engine] = per
    json.dump(old, open(json_out, "w"), indent=1, sort_keys=True)
