"""Factory with the reference's signature (active_selection/__init__.py:9-25).  The MC-dropout, CEAL, core-set,
noise and max-subset families run on this build's kernels; the accuracy-predictor family (a second network,
SURVEY.md 2 #11/#14) is outside the path and raises NotImplementedError."""
from active_selection.ceal import ActiveSelectionCEAL
from active_selection.core_set import ActiveSelectionCoreSet
from active_selection.max_subset import ActiveSelectionMaxSubset
from active_selection.mc_dropout import ActiveSelectionMCDropout
from active_selection.mc_noise import ActiveSelectionMCNoise

# --active-selection-mode -> (selector class, does its constructor take the class count first?)   (active_train.py:445-514)
_SELECTORS = {}
for _cls, _with_classes, _modes in (
        (ActiveSelectionCoreSet, False, ('coreset',)),
        (ActiveSelectionCEAL, True, ('ceal_confidence', 'ceal_margin', 'ceal_entropy', 'ceal_fusion', 'ceal_entropy_weakly_labeled')),
        (ActiveSelectionMCNoise, True, ('noise_image', 'noise_feature', 'noise_variance')),
        (ActiveSelectionMCDropout, True, ('variance', 'variance_representative', 'random'))):
    for _m in _modes:
        _SELECTORS[_m] = (_cls, _with_classes)


def get_active_selection_class(active_selection_method, dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size):
    if active_selection_method not in _SELECTORS:   # (incl. 'accuracy_labels' / 'accuracy_eval': the predictor network is out of scope)
        raise NotImplementedError
    cls, with_classes = _SELECTORS[active_selection_method]
    args = (dataset_lmdb_env, crop_size, dataloader_batch_size)
    return cls(dataset_num_classes, *args) if with_classes else cls(*args)


def get_max_subset_active_selector(dataset_lmdb_env, crop_size, dataloader_batch_size):
    return ActiveSelectionMaxSubset(dataset_lmdb_env, crop_size, dataloader_batch_size)
