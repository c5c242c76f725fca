"""SynchronizedBatchNorm2d over torch.distributed.  Two ranks (gloo here: the box has one GPU
and RCCL refuses two ranks on one device; the collective calls are backend-agnostic) each hold half of a batch;
forward activations, running statistics and every gradient must equal ONE process normalising the whole batch with
the reference's SyncBN formula (clamp(biased_var, eps)^-1/2, unbiased running_var; batchnorm.py:113-125)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch, torch.nn as nn, torch.nn.functional as F, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from dass_hip import ops
from models.sync_batchnorm import SynchronizedBatchNorm2d
from models.aspp import ASPP
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(7)
N, C, K, H = 4, 32, 64, 9
x = torch.randn(N, C, H, H, generator=g)
w = torch.randn(K, C, 3, 3, generator=g) * 0.1
gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
go = torch.randn(N, K, H, H, generator=g)

# ---- single-process truth on the whole batch, reference SyncBN formula
xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
y = F.conv2d(xr, wr, padding=1)
mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False)
eps = 1e-5
inv = var.clamp(min=eps) ** -0.5
out = F.relu((y - mean[None, :, None, None]) * (inv * gr)[None, :, None, None] + br[None, :, None, None])
out.backward(go)
cnt = N * H * H
run_var = 0.9 * 1.0 + 0.1 * var.detach() * cnt / (cnt - 1)

# ---- this rank's half through the HIP path
conv = nn.Conv2d(C, K, 3, 1, 1, bias=False).cuda(); bn = SynchronizedBatchNorm2d(K).cuda()
with torch.no_grad():
    conv.weight.copy_(w); bn.weight.copy_(gamma); bn.bias.copy_(beta)
conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
sl = slice(rank * 2, rank * 2 + 2)
xd = x[sl].cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
od = ops.conv_bn_act(xd, conv, bn, ops.ACT_RELU)
od.backward(go[sl].cuda().contiguous(memory_format=torch.channels_last))
def close(a, b, tol, what):
    err = (a.detach().float().cpu() - b.detach().float()).abs().max().item() / max(b.abs().max().item(), 1e-9)
    assert err <= tol, (what, err)
close(od, out[sl], 2e-4, "forward")
close(xd.grad, xr.grad[sl], 1e-3, "dx")
close(bn.running_var, run_var, 1e-4, "running_var")
close(bn.running_mean, 0.1 * mean, 1e-4, "running_mean")
# weight / gamma / beta gradients are per-rank partial sums of the global-batch gradient: sum over ranks
for t in (conv.weight.grad, bn.weight.grad, bn.bias.grad):
    pass
wg = conv.weight.grad.clone(); dist.all_reduce(wg); close(wg, wr.grad, 1e-3, "dw (summed over ranks)")
# dgamma / dbeta: the global sums are used inside the BN backward (dx needs them), but the PARAMETER gradients handed to
# autograd are this rank's local sums, like conv weights (torch.nn.SyncBatchNorm does the same): sum over ranks
dgs = bn.weight.grad.clone(); dist.all_reduce(dgs); close(dgs, gr.grad, 1e-3, "dgamma (summed over ranks)")
dbs = bn.bias.grad.clone(); dist.all_reduce(dbs); close(dbs, br.grad, 1e-3, "dbeta (summed over ranks)")
assert (bn.weight.grad.cpu() - gr.grad).abs().max().item() > 1e-3 * gr.grad.abs().max().item(), "local, not global, sums"

# ---- SyncBN + GradientAverager + the global-batch loss == ONE process stepping the whole batch (SURVEY 8e row 4):
# every parameter (conv weight, BN gamma / beta) must come out of the averager equal to the single-process gradient
from dass_hip.dist import GradientAverager
from utils.loss import SegmentationLosses
lab = torch.randint(0, K, (N, H, H), generator=g).float()
lab[0, :6] = 255; lab[3, :, :2] = 255                      # unequal valid-pixel counts on the two ranks
for t_ in (xr, wr, gr, br):
    t_.grad = None
y = F.conv2d(xr, wr, padding=1)
mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False)
logits = (y - mean[None, :, None, None]) * (var.clamp(min=eps) ** -0.5 * gr)[None, :, None, None] + br[None, :, None, None]
ref_loss = F.cross_entropy(logits, lab.long(), ignore_index=255) / N      # utils/loss.py:39-51 on the gathered batch
ref_loss.backward()
conv.zero_grad(); bn.zero_grad()
params = [conv.weight, bn.weight, bn.bias]
avg = GradientAverager(params, bucket_bytes=4096)
xd2 = x[sl].cuda().contiguous(memory_format=torch.channels_last)
lg = ops.conv_bn_act(xd2, conv, bn, ops.ACT_NONE)
loss = SegmentationLosses(cuda=True, global_batch=True).build_loss("ce")(lg.float().contiguous(), lab[sl].cuda())
assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
loss.backward()
assert avg.finish() >= 1
close(conv.weight.grad, wr.grad, 2e-3, "averaged dw vs single process")
close(bn.weight.grad, gr.grad, 2e-3, "averaged dgamma vs single process")
close(bn.bias.grad, br.grad, 2e-3, "averaged dbeta vs single process")
# plain BatchNorm2d must NOT synchronise
bn2 = nn.BatchNorm2d(K).cuda()
assert ops.sync_bn_world(bn2) == 1 and ops.sync_bn_world(bn) == 2
# the ASPP image-pool branch (BN after a broadcast) under SyncBN: just has to run and agree across ranks
aspp = ASPP("mobilenet", 16, SynchronizedBatchNorm2d).cuda()
torch.manual_seed(3)
for p_ in aspp.parameters():
    dist.broadcast(p_.data, 0)
o = aspp(torch.randn(2, 320, 9, 9, generator=torch.Generator().manual_seed(100 + rank)).cuda(), apply_dropout=False)
o.float().sum().backward()
rv = aspp.bn_global_average_pool.running_var.clone(); rv0 = rv.clone(); dist.broadcast(rv0, 0)
assert torch.allclose(rv, rv0), "running stats must be identical on every rank"
print("rank %d ok" % rank)
dist.destroy_process_group()
"""


def test_syncbn_two_ranks_equals_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), os.path.join(ROOT, "deep-active-semantic-segmentation_amd"), ROOT],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]
