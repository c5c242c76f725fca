./tools/probes/wg_placement_probe 548 33792
./tools/probes/wg_placement_probe 548 54000
./tools/probes/wg_placement_probe 276 49408
./tools/probes/wg_placement_probe 2192 33792
for PAD in 0 20000 46000; do echo "== DASS_X3_LDS_PAD=$PAD"; DASS_X3_LDS_PAD=$PAD DASS_F32_MMA=f16x3 python tools/x3_time.py 0 2>&1 | grep "^l3\|^l2.c\|^aspp1\|count-w"; done
