"""Round 4: all T stochastic passes of a scoring batch as one launch per conv (decoder.head_mc_all), against the per-pass tail and
against T full forwards with the same masks."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(backbone, seed):
    from dass_hip import ops
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O

    ops.set_compute_dtype(torch.float32)
    om = O.ODeepLab(backbone, 16, 19)
    O.fill_state_dict(om, seed=seed)
    pm = DeepLab(backbone=backbone, output_stride=16, num_classes=19, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    return pm.cuda().eval(), O


@pytest.mark.parametrize("backbone,n,hw,T", [("resnet", 3, 129, 4), ("mobilenet", 2, 97, 5), ("resnet", 8, 257, 10)])
def test_batched_mc_tail_equals_per_pass_tail(backbone, n, hw, T):
    """DASS_MC_BATCHED (default): N x T (image, pass) pairs ride as the "images" of ONE per-image conv launch, sharing the batch's
    deterministic residual (dass_conv2d_x3_per_image_rep, dass_split3_rows_packed_rep).  Same products per output as the per-pass
    launches, summed in a different order where the schedules differ (whole tiles vs stream-K): votes may move only at exact
    near-ties; against T full forwards with the same masks the argmax agrees wherever the top-2 margin exceeds 1e-3."""
    from dass_hip import ops

    keep_mma, keep_env = ops.f32_mma(), os.environ.get("DASS_MC_BATCHED")
    try:
        ops.set_f32_mma("f16x3")
        pm, O = _model(backbone, seed=11)
        x, _ = O.synthetic_batch(n, hw, hw, 19, first_index=300)
        m1, m2 = O.dropout_masks(n, T, seed=12)
        xd = x.cuda()
        state = pm.mc_prefix(xd)
        os.environ["DASS_MC_BATCHED"] = "1"
        vb = pm.mc_tail(state, T, masks=(m1, m2))
        assert torch.equal(vb, pm.mc_tail(state, T, masks=(m1, m2)))   # deterministic
        os.environ["DASS_MC_BATCHED"] = "0"
        vs = pm.mc_tail(state, T, masks=(m1, m2))
        diff = int((vb != vs).sum())
        print("%s %dx%d T=%d: batched vs per-pass vote differences %d of %d" % (backbone, n, hw, T, diff, vb.numel()))
        assert vb.shape == (n, T, hw, hw) and diff <= 1e-5 * vb.numel() + 2
        with torch.no_grad():
            for t in (0, T - 1):
                full = pm(xd, dropout_masks=(m1[t].cuda(), m2[t].cuda()))
                top = full.topk(2, dim=1)[0]
                safe = (top[:, 0] - top[:, 1]) > 1e-3
                assert torch.equal(full.argmax(1)[safe].to(torch.uint8), vb[:, t][safe])
    finally:
        ops.set_f32_mma(keep_mma)
        if keep_env is None:
            os.environ.pop("DASS_MC_BATCHED", None)
        else:
            os.environ["DASS_MC_BATCHED"] = keep_env
