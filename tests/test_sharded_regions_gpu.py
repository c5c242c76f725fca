"""region scoring sharded over ranks.  Two ranks (gloo; both on the box's one GPU) each run the
T-pass vote-entropy forwards and the box-sum kernels on THEIR shard of the pool, exchange the global min / max, all-gather
the normalised score maps and run the device NMS: the regions must equal ONE process scoring the whole pool
(active_selection/mc_dropout.py:82-108,123-171), on every rank."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from dass_hip import ops
from models.deeplab import DeepLab
from active_selection.mc_dropout import ActiveSelectionMCDropout
from oracle import deeplab_cpu as O
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
torch.cuda.set_device(0)
ops.set_compute_dtype(torch.float32)
ncls, hw, T, region = 19, 65, 4, 17
om = O.ODeepLab("mobilenet", 16, ncls)
O.fill_state_dict(om, seed=12)
pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
pm.load_state_dict(om.state_dict())
pm = pm.cuda().eval()
keys = [("img_%03d" % i).encode("ascii") for i in range(5)]          # 5 images over 2 ranks: shards of 3 and 2
pool = {k: O.synthetic_batch(1, hw, hw, ncls, first_index=700 + i) for i, k in enumerate(keys)}

def factory(images, include_labels, bs=2):
    for i in range(0, len(images), bs):
        chunk = images[i:i + bs]
        yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}

class FixedMasks(ActiveSelectionMCDropout):
    # the same dropout masks for every image on every rank: votes do not depend on who scores an image
    def _votes(self, model, image_batch, steps, masks=None):
        n = image_batch.shape[0]
        m1, m2 = O.dropout_masks(1, steps, seed=33)
        return super()._votes(model, image_batch, steps, masks=(m1.expand(steps, n, 256), m2.expand(steps, n, 256)))

existing = [[], [(5, 5, region, region)], [], [(30, 40, region, region)], []]
sharded = FixedMasks(ncls, None, hw, 2, loader_factory=factory)
whole = FixedMasks(ncls, None, hw, 2, loader_factory=factory, shard=False)
local, start = sharded.local_slice(keys)
assert (start, len(local)) == ((0, 3) if rank == 0 else (3, 2))
got, got_count = sharded.create_region_maps(pm, keys, existing, region, 2, steps=T)
want, want_count = whole.create_region_maps(pm, keys, existing, region, 2, steps=T)
assert got_count == want_count and got == want, (got, want)
assert got_count > 2 and len(got) >= 2
flat = sorted((k.decode(), r) for k, rs in got.items() for r in rs)
print("rank %d ok %s" % (rank, flat))
dist.destroy_process_group()
"""


def test_region_scoring_two_ranks_equals_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), os.path.join(ROOT, "deep-active-semantic-segmentation_amd"), ROOT],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]
    assert outs[0].split("ok", 1)[1].strip() == outs[1].split("ok", 1)[1].strip(), "both ranks must return the same regions"


WORKER_ONE_IMAGE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from dass_hip import ops
from models.deeplab import DeepLab
from active_selection.max_subset import ActiveSelectionMaxSubset
from oracle import deeplab_cpu as O
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
torch.cuda.set_device(0)
ops.set_compute_dtype(torch.float32)
ncls, hw, region = 19, 129, 32
om = O.ODeepLab("mobilenet", 16, ncls)
O.fill_state_dict(om, seed=12)
pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
pm.load_state_dict(om.state_dict())
pm = pm.cuda().eval()
keys = [b"img_000"]                                   # ONE image over two ranks: rank 1's shard is empty
pool = {keys[0]: O.synthetic_batch(1, hw, hw, ncls, first_index=900)}

def factory(images, include_labels, bs=2):
    for i in range(0, len(images), bs):
        chunk = images[i:i + bs]
        yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}

sharded = ActiveSelectionMaxSubset(None, hw, 2, loader_factory=factory)
whole = ActiveSelectionMaxSubset(None, hw, 2, loader_factory=factory, shard=False)
local, start = sharded.local_slice(keys)
assert len(local) == (1 if rank == 0 else 0)
got = sharded._get_features_for_image_regions(pm, keys, region)
want = whole._get_features_for_image_regions(pm, keys, region)
assert got.shape == want.shape and got.shape[0] > 1 and got.shape[1] == 304, (got.shape, want.shape)
assert torch.equal(got, want)
print("rank %d ok %s" % (rank, tuple(got.shape)))
dist.destroy_process_group()
"""


def test_image_region_features_with_an_empty_shard(tmp_path):
    """max_subset._get_features_for_image_regions with fewer images than ranks: the rank whose shard is empty never
    sees a feature map, so the cells-per-image count is agreed on by all ranks before the gather; both ranks end with the
    features one process computes"""
    script = tmp_path / "worker1.py"
    script.write_text(WORKER_ONE_IMAGE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), os.path.join(ROOT, "deep-active-semantic-segmentation_amd"), ROOT],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]
