"""Factory with the reference's signature (active_selection/__init__.py:9-21).  mode strings map to
the three selector families on this build's path; the noise / accuracy / max-subset families are the
"next" rows of SURVEY.md 8f and raise NotImplementedError here."""
from active_selection.ceal import ActiveSelectionCEAL
from active_selection.core_set import ActiveSelectionCoreSet
from active_selection.mc_dropout import ActiveSelectionMCDropout


def get_active_selection_class(active_selection_method, dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size):
    if active_selection_method == 'coreset':
        return ActiveSelectionCoreSet(dataset_lmdb_env, crop_size, dataloader_batch_size)
    elif active_selection_method in ('ceal_confidence', 'ceal_margin', 'ceal_entropy', 'ceal_fusion', 'ceal_entropy_weakly_labeled'):
        return ActiveSelectionCEAL(dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size)
    elif active_selection_method in ('variance', 'variance_representative', 'random'):
        return ActiveSelectionMCDropout(dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size)
    else:
        raise NotImplementedError
