#!/bin/bash
# wave-state counters of ONE pre-split conv shape per forced tile (run on the GPU box):  bash tools/x3_pmc.sh OUTDIR "N H W C K ks st pad dil" tile...
# pass A: SQ wave states (parked on s_waitcnt / barrier vs issue-stalled vs issuing) + MFMA busy; pass B: texture-addresser / L1 busy
set -e
OUT=$1; SHAPE=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$OUT"
[ -f "$OUT/avail.txt" ] || rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || true
for T in "$@"; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES \
      --kernel-trace --output-format csv -d "$OUT/A_t$T" -- python3 tools/x3_one.py $SHAPE 6 $T > "$OUT/A_t$T.log" 2>&1
  rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN2_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_max \
      --kernel-trace --output-format csv -d "$OUT/B_t$T" -- python3 tools/x3_one.py $SHAPE 6 $T > "$OUT/B_t$T.log" 2>&1 || echo "pass B failed for tile $T"
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
for T in sys.argv[2:]:
    for P in "AB":
        acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(int)
        for f in glob.glob(os.path.join(out, "%s_t%s" % (P, T), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])).split("(")[0]
                if "conv_x3_kernel" not in k:
                    continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(k, r["Counter_Name"])] += 1
        for k, c in acc.items():
            launches = max(n[(k, next(iter(c)))], 1)
            print("tile %s pass %s %s (%d launches): " % (T, P, k, launches) + "  ".join("%s=%.4g" % (name, v / launches) for name, v in sorted(c.items())))
            if "SQ_WAVE_CYCLES" in c:
                w = c["SQ_WAVE_CYCLES"]
                print("      of wave-cycles: parked (waitcnt/barrier) %.2f, issue-stalled %.2f (of which LDS %.2f), issuing %.2f; MFMA busy of GUI x 1024 SIMDs: %.3f" % (
                    c["SQ_WAIT_ANY"] / w, c["SQ_WAIT_INST_ANY"] / w, c.get("SQ_WAIT_INST_LDS", 0) / w, c["SQ_ACTIVE_INST_ANY"] / w,
                    c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
PY
