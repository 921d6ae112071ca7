"""Building blocks shared by the SNws model graphs."""
import torch
import torch.nn as nn

import modeling.snorm as sn
import modeling.weightstd as ws
from snws import ops


class ConvSnRelu(nn.Sequential):
    """Conv2dws -> SwitchNorm2d -> ReLU with the reference's layout (Sequential indices 0/1/2, so state_dict
    keys read '<name>.0.weight', '<name>.1.running_mean', ... as produced by ConvSnRelu at
    modeling/UNet3p_SNws.py:29-33), executed as ONE fused op: the conv epilogue emits the per-channel partial
    moments SwitchNorm needs, SwitchNorm applies the ReLU, and backward recomputes the ReLU mask from the conv
    output instead of storing the activation twice."""

    def __init__(self, in_channels, out_channels, using_movavg=True, using_bn=True, last_gamma=False, kernel_size=3,
                 stride=1, padding=1):
        super(ConvSnRelu, self).__init__(
            ws.Conv2dws(in_channels, out_channels, kernel_size, stride, padding),
            sn.SwitchNorm2d(out_channels, using_movavg=using_movavg, using_bn=using_bn, last_gamma=last_gamma),
            nn.ReLU())

    def forward(self, x, out=None):
        """out: optional channels-last destination (e.g. a channel slice of a concat buffer) for the block output."""
        conv, norm = self[0], self[1]
        if x.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(x.dim()))
        ops.require_gpu(x, "ConvSnRelu")
        prep = conv.prepared(x.dtype == torch.bfloat16 and not conv.force_direct)
        if conv.force_direct:
            y = ops.Conv2dwsFn.apply(x, conv.weight, conv.bias, prep, True)
            y = ops.SwitchNormFn.apply(y, norm.weight, norm.bias, norm.mean_weight, norm.var_weight,
                                       ops.SNParams(norm), True)
            if out is not None:
                raise NotImplementedError("out= is not available with force_direct")
            return y
        return ops.ConvSnReluFn.apply(x, conv.weight, conv.bias, norm.weight, norm.bias, norm.mean_weight,
                                      norm.var_weight, prep, ops.SNParams(norm), True, out)


    # ---- the block applied to nn.Upsample(scale_factor=k, mode='bilinear')(x), without building the up-sampled map
    def taps(self, x):
        """Z = W_taps x at x's resolution; feed it to up(Z, k) for every scale k that needs conv(upsample_k(x))."""
        conv = self[0]
        ops.require_gpu(x, "ConvSnRelu")
        tp = conv.prepared_taps(x.dtype == torch.bfloat16 and not conv.force_direct)
        return ops.TapConvFn.apply(x, conv.weight, tp)

    def up(self, z, k, out=None):
        conv, norm = self[0], self[1]
        return ops.UpTapSnReluFn.apply(z, conv.bias, norm.weight, norm.bias, norm.mean_weight, norm.var_weight,
                                       ops.SNParams(norm), int(k), conv.kernel_size[0], conv.out_channels, True, out)

    def on_upsampled(self, x, k):
        """== self(nn.Upsample(scale_factor=k, mode='bilinear', align_corners=False)(x))."""
        return self.up(self.taps(x), k)


def crop_concat(a, b):
    """Centre-crop skip `a` to b's spatial size (diff//2 from the top/left) and stack (a, b) on the channel axis
    (CropConcat, modeling/UNet_SNws.py:35-39)."""
    dy = a.size(2) - b.size(2)
    dx = a.size(3) - b.size(3)
    if dy or dx:
        a = a[:, :, dy // 2: a.size(2) - (dy - dy // 2), dx // 2: a.size(3) - (dx - dx // 2)]
    return torch.cat((a, b), dim=1)


def to_device_layout(x, dtype=None):
    """Input images (N,C,H,W) -> channels-last tensor of the compute dtype."""
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous(memory_format=torch.channels_last)


def head_conv(conv, x):
    """The plain nn.Conv2d 1x1 classification head (modeling/UNet3p_SNws.py:93,158) through the library's own conv
    kernels; returns float32 logits.  For bf16 activations the filter is zero-padded to a multiple of 16 output
    channels so the MFMA kernels take it; the padding is sliced off again (autograd sees plain pad/slice ops)."""
    ops.require_gpu(x, "classification head")
    w, b = conv.weight, conv.bias
    n_out = w.shape[0]
    if conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 or \
            tuple(conv.padding) != ((w.shape[2] - 1) // 2, (w.shape[3] - 1) // 2):
        raise NotImplementedError("head_conv handles stride-1 'same' convolutions")
    pad = (-n_out) % 16 if x.dtype == torch.bfloat16 else 0
    if pad:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, pad))
        b = None if b is None else torch.nn.functional.pad(b, (0, pad))
    y = ops.PlainConvFn.apply(x, w, b)
    return y[:, :n_out].float()
