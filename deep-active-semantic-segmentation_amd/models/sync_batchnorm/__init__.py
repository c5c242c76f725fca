"""Stand-in for the reference's vendored models/sync_batchnorm (batchnorm.py:48-125, replicate.py:65-88).

The reference synchronises BN statistics between DataParallel threads of ONE process.  The MI355X design
is one process per GPU: SynchronizedBatchNorm2d is an nn.BatchNorm2d whose batch sums the HIP BN path
all-reduces over torch.distributed (RCCL over xGMI) in forward ([2K] floats) and backward ([2K] floats), one
small collective per layer and direction.  `patch_replication_callback` / `DataParallelWithCallback` are kept
as no-op names so `active_train.py:82-85`-style code keeps importing.
"""
import torch.nn as nn


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    # The HIP BN path all-reduces the [2K] batch sums of instances carrying this mark over the ranks of
    # torch.distributed (RCCL) and normalises with clamp(var, eps)^-1/2 like the vendored original
    # (batchnorm.py:113-125); without an initialised process group it is a plain BatchNorm2d.
    _dass_sync = True


class SynchronizedBatchNorm1d(nn.BatchNorm1d):
    pass


def patch_replication_callback(data_parallel):
    return data_parallel


class DataParallelWithCallback(nn.DataParallel):
    pass
