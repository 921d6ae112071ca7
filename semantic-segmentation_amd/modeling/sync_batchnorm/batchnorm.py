"""SynchronizedBatchNorm{1,2,3}d with the reference's constructor and statistics
(modeling/sync_batchnorm/batchnorm.py:28-141): per-channel sum and square-sum over (N, spatial), mean = sum/size,
normalisation with the BIASED variance CLAMPED at eps (not +eps) on the synchronised path, running buffers updated
with the UNBIASED variance and PyTorch's momentum convention.

Synchronisation: one process per GPU; in training mode under an initialised torch.distributed group the packed
per-channel moments are all-reduced (RCCL), forward and backward.  Every input runs on the library's streaming
kernels (snws_sn_stats / snws_bn_* / snws_sn_apply and their backward counterparts): (N, C), (N, C, L) and
(N, C, D, H, W) inputs are viewed as (N, C, positions, 1), one-position-per-sample inputs as one image of N positions.  Outside a process group, and in eval mode, the inverse
standard deviation is (var + eps)^-1/2 like the reference's F.batch_norm fallback (batchnorm.py:40-43).  There is no
CPU path: CPU tensors raise.  Only DeepLab-style models use this class (modeling/deeplab.py:16-19)."""
import torch.distributed as dist
from torch.nn.modules.batchnorm import _BatchNorm

__all__ = ['SynchronizedBatchNorm1d', 'SynchronizedBatchNorm2d', 'SynchronizedBatchNorm3d']


class _SynchronizedBatchNorm(_BatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super(_SynchronizedBatchNorm, self).__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.process_group = None

    def _synced(self):
        return self.training and dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(self.process_group) > 1

    def forward(self, input):
        self._check_input_dim(input)
        from snws import ops
        ops.require_gpu(input, type(self).__name__)
        if self.momentum is None:
            raise ValueError("%s: momentum=None (cumulative average) is not supported" % type(self).__name__)
        shape, C = input.shape, self.num_features
        per_sample = input.numel() // max(shape[0] * C, 1)
        if input.dim() == 4:
            x = input
        elif per_sample == 1:     # (N, C) or (N, C, 1): the batch axis becomes the pixel axis of ONE image - the kernels
            x = input.reshape(shape[0], C).t().reshape(1, C, shape[0], 1)    # take per-image moments over >= 2 positions
        else:
            x = input.reshape(shape[0], C, -1, 1)
        if self.training and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        # the reference clamps only on its synchronised path; its fallback is F.batch_norm (var + eps)
        y = ops.BatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum,
                                  self.eps, self.training, False, True, self.process_group, self._synced())
        if input.dim() == 4:
            return y
        return y.reshape(C, shape[0]).t().reshape(shape) if per_sample == 1 else y.reshape(shape)


class SynchronizedBatchNorm1d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 2 and input.dim() != 3:
            raise ValueError('expected 2D or 3D input (got {}D input)'.format(input.dim()))


class SynchronizedBatchNorm2d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(input.dim()))


class SynchronizedBatchNorm3d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 5:
            raise ValueError('expected 5D input (got {}D input)'.format(input.dim()))
