#!/bin/bash
# Profiling recipe (rounds 4-5) (run on the GPU box from the repo root):  bash tools/profile_step.sh OUTDIR [extra bench flags]
#   1. rocprofv3 --kernel-trace --stats of the train leg (the command bench.py's in-step `roofline` is checked against)
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) over a short train leg, summarised per kernel
# rocprofv3 gets `python3 bench.py` itself after `--` (no env / bash -c hop: the profiler's library initialises the GPU first).
set -e
OUT=${1:-gpurun_out/prof}; shift || true
FLAGS="--no-cpu-baseline --no-mc --no-roofline --no-second-dtype --no-coreset --no-pool-reader $*"
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 3 $FLAGS > "$OUT/trace_bench.json" 2> "$OUT/trace_bench.err"
DB=$(find "$OUT/trace" -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" 0 48 --after sgd_multi 18 > "$OUT/train_summary.md"
echo "trace done: $DB"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py --steps 2 --warmup 2 $FLAGS > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err"
  echo "pmc $C done"
done
python3 tools/pmc_bytes.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" 0 "$OUT/pmc_dominant.json" f16x3 > "$OUT/pmc_bytes.txt"
#   3. matrix-pipe occupancy per kernel of the step: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_MFMA" -- python3 bench.py --steps 2 --warmup 2 $FLAGS > "$OUT/pmc_MFMA.json" 2> "$OUT/pmc_MFMA.err"
python3 tools/pmc_mfma.py "$OUT/pmc_MFMA" 0 > "$OUT/pmc_mfma.txt"
echo "pmc MFMA done"
# keep the merge-back small: the raw csv / db files stay on the box
find "$OUT" -name "*.db" -size +20M -delete; find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
