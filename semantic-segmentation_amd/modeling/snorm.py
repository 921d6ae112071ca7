"""Switchable Normalization, same module API as the reference's modeling/snorm.py:5-81.

Same constructor, parameters (`weight`, `bias` of shape (1,C,1,1); `mean_weight`, `var_weight` of 3 or 2
entries), buffers (`running_mean`, `running_var` of shape (1,C,1), starting at zero) and the same arithmetic:
IN / LN / BN moments mixed by two softmaxes, running statistics updated with the 0.997 "keep" momentum or
accumulated, affine output.  The work is done by HIP kernels (one statistics pass, a few tiny per-(n,c)
kernels, one normalise pass; likewise two passes backward).

Under torch.distributed (one process per GPU) the BN branch sums its per-channel statistics over all ranks
with one small all-reduce per call forward and one backward (RCCL), so N ranks with batch B each compute
exactly what one device would on the batch N*B; set `sync_bn_stats = False` on a module to keep the
statistics local like the reference's nn.DataParallel replicas did.
"""
import torch
import torch.nn as nn

from snws import ops


class SwitchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.997, using_movavg=True, using_bn=True, last_gamma=False):
        super(SwitchNorm2d, self).__init__()
        self.eps = eps
        self.momentum = momentum
        self.using_movavg = using_movavg
        self.using_bn = using_bn
        self.last_gamma = last_gamma
        self.sync_bn_stats = True
        self.process_group = None
        self.fuse_relu = False      # set by ConvSnRelu-style containers that fold the following ReLU in
        n_branch = 3 if using_bn else 2
        self.weight = nn.Parameter(torch.ones(1, num_features, 1, 1))
        self.bias = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.mean_weight = nn.Parameter(torch.ones(n_branch))
        self.var_weight = nn.Parameter(torch.ones(n_branch))
        if using_bn:
            self.register_buffer('running_mean', torch.zeros(1, num_features, 1))
            self.register_buffer('running_var', torch.zeros(1, num_features, 1))
        self.reset_parameters()

    def reset_parameters(self):
        if self.using_bn:
            self.running_mean.zero_()
            self.running_var.zero_()
        self.weight.data.fill_(0 if self.last_gamma else 1)
        self.bias.data.zero_()

    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(input.dim()))

    def forward(self, x):
        self._check_input_dim(x)
        ops.require_gpu(x, "SwitchNorm2d")
        return ops.SwitchNormFn.apply(x, self.weight, self.bias, self.mean_weight, self.var_weight,
                                      ops.SNParams(self), self.fuse_relu)
