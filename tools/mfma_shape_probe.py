#!/usr/bin/env python
"""bf16 MFMA rate and shader clock with all CUs busy, 32x32x16 against 16x16x32 (dass_clock_probe modes 0/1 and 2/3: operands
in registers / re-read from LDS every iteration).  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from clock_probe import probe  # noqa: E402

cus = torch.cuda.get_device_properties(0).multi_processor_count
for blocks in (cus, 2 * cus):
    for mode, name in ((0, "32x32x16 regs"), (1, "32x32x16 lds"), (2, "16x16x32 regs"), (3, "16x16x32 lds")):
        r = probe(blocks, mode)
        print("%4d blocks  %-14s clock %.3f GHz  %7.1f TFLOP/s  %.2f ms" % (blocks, name, r["clock_ghz_median"], r["mfma_tflops"], r["ms"]), flush=True)
