#!/usr/bin/env python
"""Shader clock MI355X sustains under a bf16 MFMA load, by the number of busy CUs (dass_clock_probe): the measured basis of
DESIGN.md's "sustained MFMA ceiling".  The nominal 2.5 PFLOP/s dense bf16 peak is 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4
GHz; under an MFMA-dense load on all CUs the chip holds a lower clock (power), so the ceiling a kernel can reach scales with
it.  Prints one JSON line; GPU only.   python tools/clock_probe.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import check, lib  # noqa: E402


def probe(blocks, use_lds, iters=60000, warm=40):
    out = torch.zeros((blocks, 2), dtype=torch.int64, device="cuda")
    for _ in range(warm):  # ~2 s of back-to-back load before the measured launch: DVFS settles
        check(lib.dass_clock_probe(ops._p(out), blocks, iters, use_lds, ops._stream()), "dass_clock_probe")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.dass_clock_probe(ops._p(out), blocks, iters, use_lds, ops._stream()), "dass_clock_probe")
    e1.record()
    torch.cuda.synchronize()
    o = out.cpu().double()
    ghz = (o[:, 0] / o[:, 1] * 0.1)
    ms = e0.elapsed_time(e1)
    # 4 waves x 4 MFMAs of 32x32x16 (32768 FLOP each... 2*32*32*16) per iteration and block
    tflops = blocks * iters * 16 * (2 * 32 * 32 * 16) / (ms * 1e-3) / 1e12
    return {"blocks": blocks, "lds_reads": bool(use_lds), "clock_ghz_median": round(float(ghz.median()), 3),
            "clock_ghz_min": round(float(ghz.min()), 3), "clock_ghz_max": round(float(ghz.max()), 3), "ms": round(ms, 3),
            "mfma_tflops": round(tflops, 1)}


def main():
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rows = []
    for blocks in (cus // 8, cus // 2, cus, 2 * cus):
        for use_lds in (0, 1):
            rows.append(probe(blocks, use_lds))
            print(rows[-1], flush=True)
    full = [r for r in rows if r["blocks"] == 2 * cus and r["lds_reads"]][0]
    line = {"device": torch.cuda.get_device_name(0), "cus": cus, "nominal_clock_ghz": 2.4, "probe": rows,
            "sustained_clock_ghz_all_cus_lds_fed": full["clock_ghz_median"],
            "sustained_bf16_mfma_peak_tflops": round(2500.0 * full["clock_ghz_median"] / 2.4, 1)}
    print(json.dumps(line))
    if len(sys.argv) > 1:
        json.dump(line, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
