"""eager vs graphed losses of tests/test_round5_gpu.py's step under variations of the warm-up (diagnostic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dass_hip import ops  # noqa: E402
from dass_hip.graph import GraphedStep  # noqa: E402
from dass_hip.optim import SGD  # noqa: E402
from models.deeplab import DeepLab  # noqa: E402
from oracle import deeplab_cpu as O  # noqa: E402
from utils.loss import SegmentationLosses  # noqa: E402

ops.set_f32_mma("f16x3")
ops.set_compute_dtype(torch.float32)


def run(mode):
    om = O.ODeepLab("resnet", 16, 19)
    O.fill_state_dict(om, seed=8, randomize_bn_stats=False)
    x, lab = O.synthetic_batch(4, 129, 129, 19, first_index=800)
    xd, ld = x.cuda(), lab.cuda()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=19, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().train()
    opt = SGD([{"params": pm.get_1x_lr_params(), "lr": 0.01}, {"params": pm.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    masks = O.dropout_masks(4, 1, seed=9)
    dm = (masks[0][0].cuda(), masks[1][0].cuda())

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(pm(xd, dropout_masks=dm), ld)
        loss.backward()
        opt.step()
        return loss

    losses = []
    if mode.endswith("_prealloc"):
        mode = mode[:-9]
        for gi in range(2):
            opt._hyper_tensor(gi, xd.device)
    if mode == "eager":
        for i in range(6):
            losses.append(float(step().detach()))
        return losses
    if mode == "inside":
        gs = GraphedStep(step, warmup=2)
        losses += [None, None]
    elif mode == "manual":
        for i in range(2):
            losses.append(float(step().detach()))
        gs = GraphedStep(step, warmup=0)
    elif mode == "manual_side":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(2):
                losses.append(float(step().detach()))
        torch.cuda.current_stream().wait_stream(side)
        gs = GraphedStep(step, warmup=0)
    elif mode == "manual1_inside1":
        losses.append(float(step().detach()))
        gs = GraphedStep(step, warmup=1)
        losses.append(None)
    for i in range(4):
        losses.append(float(gs().detach()))
        print("   hyper", {gi: ent["dev"].cpu().tolist() for gi, ent in opt.__dict__.get("_dass_hyper", {}).items()})
    gs.release()
    return losses


for det in (False, True):
    ops.set_deterministic(det)
    for mode in sys.argv[1:] or ["eager", "inside", "manual", "manual_side", "manual1_inside1"]:
        print("det", det, "hyper", os.environ.get("DASS_SGD_DEV_HYPER", "1"), mode, ["%.6f" % v if v is not None else None for v in run(mode)], flush=True)
