"""resident workgroups per CU of the whole-tile conv kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor through the C-ABI)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch
torch.zeros(1, device="cuda")
from dass_hip._lib import lib
for which, name, ring in ((0, "64x64", 32), (1, "128x64", 48), (2, "256x128", 96)):
    n = lib.dass_x3_resident_workgroups(which)
    print("whole %s tiles: %d workgroups per CU resident = %d KB of ring in flight" % (name, n, n * ring))
