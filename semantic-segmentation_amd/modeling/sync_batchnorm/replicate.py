"""Replica notification for torch's thread-based DataParallel, with the public names of the reference's
modeling/sync_batchnorm/replicate.py:17-48.

A module may define `__data_parallel_replicate__(ctx, copy_id)`.  Whenever a DataParallel wrapper clones its model
onto the devices, the clones of one sub-module are told which copy they are (0 = the master, visited first) and are
handed one shared `CallbackContext` to find each other through.  The SNws models do not need this - SwitchNorm2d
synchronises through torch.distributed, one process per GPU - but the reference's train.py:210-214 applies
`patch_replication_callback` to the wrapped model, so the hook protocol is provided and behaves as documented there.
"""
from torch.nn.parallel.data_parallel import DataParallel

__all__ = ['CallbackContext', 'execute_replication_callbacks', 'DataParallelWithCallback',
           'patch_replication_callback']

_HOOK = '__data_parallel_replicate__'


class CallbackContext(object):
    """An empty namespace; the clones of one sub-module hang their rendez-vous state on it."""


def execute_replication_callbacks(modules):
    """modules: the clones of one model, master first.  Clone k's j-th sub-module (in `.modules()` order) is called
    as `sub.__data_parallel_replicate__(context_j, k)`; context_j is the same object for every k."""
    shared = []
    for copy_id, clone in enumerate(modules):
        for j, sub in enumerate(clone.modules()):
            if copy_id == 0:
                shared.append(CallbackContext())
            notify = getattr(sub, _HOOK, None)
            if notify is not None:
                notify(shared[j], copy_id)


def _notifying(replicate):
    """Wrap a DataParallel.replicate bound method so that the clones it returns have been notified."""
    def replicate_and_notify(module, device_ids):
        clones = replicate(module, device_ids)
        execute_replication_callbacks(clones)
        return clones
    replicate_and_notify.__doc__ = replicate.__doc__
    replicate_and_notify.__name__ = getattr(replicate, '__name__', 'replicate')
    return replicate_and_notify


def patch_replication_callback(data_parallel):
    """Give an existing DataParallel object the notifying `replicate` (instance attribute; the class is untouched)."""
    if not isinstance(data_parallel, DataParallel):
        raise AssertionError("patch_replication_callback expects a torch.nn.DataParallel instance")
    data_parallel.replicate = _notifying(data_parallel.replicate)


class DataParallelWithCallback(DataParallel):
    """DataParallel whose every instance is patched at construction."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        patch_replication_callback(self)
