#!/bin/bash
# MC-dropout leg alone: rocprofv3 kernel statistics + matrix-pipe occupancy per kernel (run on the GPU box):  bash tools/profile_mc.sh OUTDIR
set -e
OUT=${1:-gpurun_out/prof_mc}
FLAGS="--only mc --no-cpu-baseline --no-roofline --no-second-dtype --no-coreset --no-pool-reader"
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 bench.py $FLAGS > "$OUT/trace_bench.json" 2> "$OUT/trace_bench.err"
DB=$(find "$OUT/trace" -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" 1 28 > "$OUT/mc_summary.md"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_MFMA" -- python3 bench.py $FLAGS --mc-batches 12 > "$OUT/pmc_MFMA.json" 2> "$OUT/pmc_MFMA.err"
python3 tools/pmc_mfma.py "$OUT/pmc_MFMA" 1 > "$OUT/pmc_mfma.txt"
find "$OUT" -name "*.db" -size +20M -delete; find "$OUT" -name "*kernel_trace.csv" -size +20M -delete; find "$OUT" -name "*counter_collection.csv" -size +30M -delete
