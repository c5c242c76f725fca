from . import DataParallelWithCallback, patch_replication_callback  # noqa: F401
