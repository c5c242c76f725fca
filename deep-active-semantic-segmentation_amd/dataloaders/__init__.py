"""Pool reader of the acquisition-scoring loops -- the part of the reference's `dataloaders` package
that sits directly in front of the hot path: `dataset.paths_dataset.PathsDataset` and the transforms it composes.  The
training datasets / augmentation pipeline of the reference (cityscapes.py, pascal.py, active_*.py, make_dataloader) are
outside this build's scope."""
