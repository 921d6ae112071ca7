"""DataParallel replication callbacks with the reference's API (modeling/sync_batchnorm/replicate.py:17-48).

`patch_replication_callback(dp)` / `DataParallelWithCallback` make every module that defines
`__data_parallel_replicate__(ctx, copy_id)` receive a per-module context shared by its replicas.  The SNws models
do not rely on it (SwitchNorm2d synchronises through torch.distributed), but `train.py:210-214` calls it on the
wrapped model, so it is kept and works as documented."""
import functools

from torch.nn.parallel.data_parallel import DataParallel

__all__ = ['CallbackContext', 'execute_replication_callbacks', 'DataParallelWithCallback',
           'patch_replication_callback']


class CallbackContext(object):
    pass


def execute_replication_callbacks(modules):
    """modules: the replicas of one model.  Sub-module j of every replica gets the same context object; replica 0
    (the master copy) is called first."""
    n_sub = sum(1 for _ in modules[0].modules())
    contexts = [CallbackContext() for _ in range(n_sub)]
    for copy_id, replica in enumerate(modules):
        for j, sub in enumerate(replica.modules()):
            hook = getattr(sub, '__data_parallel_replicate__', None)
            if hook is not None:
                hook(contexts[j], copy_id)


class DataParallelWithCallback(DataParallel):
    def replicate(self, module, device_ids):
        replicas = super(DataParallelWithCallback, self).replicate(module, device_ids)
        execute_replication_callbacks(replicas)
        return replicas


def patch_replication_callback(data_parallel):
    """Monkey-patch an existing DataParallel object so that its replicate() runs the callbacks."""
    assert isinstance(data_parallel, DataParallel)
    inner = data_parallel.replicate

    @functools.wraps(inner)
    def replicate_with_callbacks(module, device_ids):
        replicas = inner(module, device_ids)
        execute_replication_callbacks(replicas)
        return replicas

    data_parallel.replicate = replicate_with_callbacks
