from . import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d  # noqa: F401
