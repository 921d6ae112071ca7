"""SynchronizedBatchNorm{1,2,3}d with the reference's constructor and statistics
(modeling/sync_batchnorm/batchnorm.py:28-141): per-channel sum and square-sum over (N, spatial), mean = sum/size,
normalisation with the BIASED variance CLAMPED at eps (not +eps), running buffers updated with the UNBIASED
variance and PyTorch's momentum convention.

Synchronisation: one process per GPU; in training mode under an initialised torch.distributed group the packed
per-channel moments are all-reduced (RCCL), forward and backward.  4-D GPU inputs (SynchronizedBatchNorm2d) run on
the library's streaming kernels (snws_sn_stats / snws_bn_* / snws_sn_apply and their backward counterparts) with
the clamp-at-eps inverse standard deviation in training and evaluation alike; 1-D / 3-D inputs and CPU tensors keep
the composition of stock torch ops below (outside a process group, or in eval mode, F.batch_norm like the
reference's fallback, batchnorm.py:40-43).  Only DeepLab-style models use this class (modeling/deeplab.py:16-19)."""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

__all__ = ['SynchronizedBatchNorm1d', 'SynchronizedBatchNorm2d', 'SynchronizedBatchNorm3d']


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, group):
        ctx.group = group
        t = t.clone()
        dist.all_reduce(t, group=group)
        return t

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


class _SynchronizedBatchNorm(_BatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super(_SynchronizedBatchNorm, self).__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.process_group = None

    def _synced(self):
        return self.training and dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(self.process_group) > 1

    def forward(self, input):
        self._check_input_dim(input)
        if input.dim() == 4 and input.is_cuda and self.track_running_stats and self.momentum is not None:
            from snws import ops
            if self.training and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            # the reference clamps only on its synchronised path; its fallback is F.batch_norm (var + eps)
            clamp = self._synced()
            return ops.BatchNormFn.apply(input, self.weight, self.bias, self.running_mean, self.running_var,
                                         self.momentum, self.eps, self.training, False, True, self.process_group, clamp)
        if not self._synced():
            return F.batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias, self.training,
                                self.momentum, self.eps)
        shape = input.size()
        x = input.reshape(shape[0], self.num_features, -1)
        local = torch.cat([x.sum((0, 2)), (x * x).sum((0, 2)),
                           x.new_tensor([float(x.size(0) * x.size(2))])])
        total = _AllReduceSum.apply(local, self.process_group)
        C = self.num_features
        size = total[2 * C].detach()
        assert float(size) > 1, 'BatchNorm computes unbiased standard-deviation, which requires size > 1.'
        mean = total[:C] / size
        sumvar = total[C:2 * C] - total[:C] * mean
        with torch.no_grad():
            self.running_mean.mul_(1 - self.momentum).add_(self.momentum * mean)
            self.running_var.mul_(1 - self.momentum).add_(self.momentum * sumvar / (size - 1))
        inv_std = (sumvar / size).clamp(self.eps) ** -0.5
        if self.affine:
            y = (x - mean.view(1, C, 1)) * (inv_std * self.weight).view(1, C, 1) + self.bias.view(1, C, 1)
        else:
            y = (x - mean.view(1, C, 1)) * inv_std.view(1, C, 1)
        return y.view(shape)


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
