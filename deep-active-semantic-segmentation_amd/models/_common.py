"""helpers shared by the HIP-backed model mirrors"""
import os
import warnings

import torch
import torch.nn as nn

from models.sync_batchnorm import SynchronizedBatchNorm2d


def channels_last_weights(module):
    """keep every 4-D conv weight in channels_last (= KRSC) storage so the forward operand is zero-copy;
    state_dict round trips and optimizers are layout-agnostic."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


def init_weights(module):
    """kaiming-normal convs, BN gamma=1 beta=0 (aspp.py:91-101, decoder.py:50-60, resnet.py:135-144)"""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, (SynchronizedBatchNorm2d, nn.BatchNorm2d)):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def load_local_pretrained(module, filename):
    """the reference downloads ImageNet weights at construction (resnet.py:147-157, mobilenet.py:146-157);
    this build never touches the network: a file under $DASS_PRETRAINED_DIR is used if present."""
    root = os.environ.get("DASS_PRETRAINED_DIR")
    path = os.path.join(root, filename) if root else None
    if not path or not os.path.exists(path):
        warnings.warn("pretrained=True but %s is not available locally (set DASS_PRETRAINED_DIR); "
                      "keeping random initialisation -- nothing is downloaded" % filename)
        return
    pretrain_dict = torch.load(path, map_location="cpu")
    state_dict = module.state_dict()
    state_dict.update({k: v for k, v in pretrain_dict.items() if k in state_dict})
    module.load_state_dict(state_dict)


def dropout_mask_for(drop, n, c, device):
    """[N,C] multipliers of an nn.Dropout2d in train mode, None otherwise"""
    if drop is None or not drop.training or drop.p == 0:
        return None
    from dass_hip import ops

    return ops.dropout2d_mask(n, c, drop.p, device)
