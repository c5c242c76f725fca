"""Factory with the reference's signature (active_selection/__init__.py:9-25).  The MC-dropout, CEAL, core-set,
noise and max-subset families run on this build's kernels; the accuracy-predictor family (a second network,
SURVEY.md 2 #11/#14) is outside the path and raises NotImplementedError."""
from active_selection.ceal import ActiveSelectionCEAL
from active_selection.core_set import ActiveSelectionCoreSet
from active_selection.max_subset import ActiveSelectionMaxSubset
from active_selection.mc_dropout import ActiveSelectionMCDropout
from active_selection.mc_noise import ActiveSelectionMCNoise


def get_active_selection_class(active_selection_method, dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size):
    if active_selection_method == 'coreset':
        return ActiveSelectionCoreSet(dataset_lmdb_env, crop_size, dataloader_batch_size)
    elif active_selection_method in ('ceal_confidence', 'ceal_margin', 'ceal_entropy', 'ceal_fusion', 'ceal_entropy_weakly_labeled'):
        return ActiveSelectionCEAL(dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size)
    elif active_selection_method in ('noise_image', 'noise_feature', 'noise_variance'):
        return ActiveSelectionMCNoise(dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size)
    elif active_selection_method in ('variance', 'variance_representative', 'random'):
        return ActiveSelectionMCDropout(dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size)
    else:
        raise NotImplementedError


def get_max_subset_active_selector(dataset_lmdb_env, crop_size, dataloader_batch_size):
    return ActiveSelectionMaxSubset(dataset_lmdb_env, crop_size, dataloader_batch_size)
