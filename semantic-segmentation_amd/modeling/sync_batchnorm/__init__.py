"""Cross-GPU batch statistics, import-compatible with the reference's modeling/sync_batchnorm package
(`SynchronizedBatchNorm{1,2,3}d`, `patch_replication_callback`, `DataParallelWithCallback`).

The reference exchanged per-channel sum / square-sum between nn.DataParallel replica THREADS through a
master/slave queue protocol (comm.py) and torch's reduce-add / broadcast (batchnorm.py:80-101).  Here every GPU has
its own process, and the same payload - packed per-channel sums plus the element count - travels in ONE
all-reduce over RCCL (torch.distributed backend "nccl"), after which every rank derives identical statistics.
SwitchNorm2d uses that mechanism directly inside its forward/backward (snws.ops); this package keeps the reference's
public names working for code that imports them.
"""
from .batchnorm import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d, SynchronizedBatchNorm3d
from .replicate import DataParallelWithCallback, patch_replication_callback

__all__ = ['SynchronizedBatchNorm1d', 'SynchronizedBatchNorm2d', 'SynchronizedBatchNorm3d',
           'DataParallelWithCallback', 'patch_replication_callback']
