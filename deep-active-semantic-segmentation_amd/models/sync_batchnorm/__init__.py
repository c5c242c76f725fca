"""Stand-in for the reference's vendored models/sync_batchnorm (batchnorm.py:48-125, replicate.py:65-88).

The reference synchronises BN statistics between DataParallel threads of ONE process.  The MI355X design
is one process per GPU, so SynchronizedBatchNorm2d is an nn.BatchNorm2d that the HIP BN path treats
like any other BN (per-GPU statistics; the per-GPU batch of 8 at 513^2 makes that the default, see
DESIGN.md) and `patch_replication_callback` / `DataParallelWithCallback` are kept as no-op names so
`active_train.py:82-85`-style code keeps importing.
"""
import torch.nn as nn


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    pass


class SynchronizedBatchNorm1d(nn.BatchNorm1d):
    pass


def patch_replication_callback(data_parallel):
    return data_parallel


class DataParallelWithCallback(nn.DataParallel):
    pass
