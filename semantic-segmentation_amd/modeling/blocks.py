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
        return ops.tag_sn_output(ops.ConvSnReluFn.apply(x, conv.weight, conv.bias, norm.weight, norm.bias,
                                                        norm.mean_weight, norm.var_weight, prep, ops.SNParams(norm),
                                                        True, out))


    def with_head(self, x, head):
        """head(self(x)) for a 1x1 nn.Conv2d classification head as ONE fused op (no normalised activation, no
        gradient of it, ever stored): the raw logits (N, 16, H, W) bf16, channels >= head.out_channels are padding.
        Call only when ops.head_fusable(x, cfg, head)."""
        conv, norm = self[0], self[1]
        prep = conv.prepared(True)
        return ops.ConvSnReluHeadFn.apply(x, conv.weight, conv.bias, norm.weight, norm.bias, norm.mean_weight,
                                          norm.var_weight, head.weight, head.bias, prep, ops.SNParams(norm))

    def head_fusable(self, x, head):
        conv = self[0]
        cfg = ops.ConvCfg(conv.weight.shape, conv.stride, conv.padding, conv.dilation, conv.groups)
        return (not conv.force_direct) and x.dim() == 4 and x.is_cuda and ops.head_fusable(x, cfg, head)

    # ---- the block applied to nn.Upsample(scale_factor=k, mode='bilinear')(x), without building the up-sampled map
    def taps(self, x):
        """Z = W_taps x at x's resolution; feed it to up(Z, k) for every scale k that needs conv(upsample_k(x))."""
        conv = self[0]
        ops.require_gpu(x, "ConvSnRelu")
        tp = conv.prepared_taps(x.dtype == torch.bfloat16 and not conv.force_direct)
        return ops.TapConvFn.apply(x, conv.weight, tp)

    def up(self, z, k, out=None):
        conv, norm = self[0], self[1]
        return ops.tag_sn_output(ops.UpTapSnReluFn.apply(z, conv.bias, norm.weight, norm.bias, norm.mean_weight,
                                                         norm.var_weight, ops.SNParams(norm), int(k),
                                                         conv.kernel_size[0], conv.out_channels, True, out))

    def on_upsampled(self, x, k):
        """== self(nn.Upsample(scale_factor=k, mode='bilinear', align_corners=False)(x))."""
        return self.up(self.taps(x), k)


def crop_concat(a, b):
    """Fit skip `a` to b's spatial size and stack (a, b) on the channel axis (CropConcat, modeling/UNet_SNws.py:35-39:
    F.pad with the negated size differences).  Where `a` is larger it is centre-cropped (diff//2 from the top/left);
    where it is SMALLER - a ceil-mode pool followed by the x2 up-sampling leaves the decoder map one pixel larger than
    the skip on odd sizes, e.g. 75 -> 38 -> 76 - it is zero-padded by the same rule (Python floor division: -1//2 = -1,
    so the single extra row/column goes to the top/left)."""
    dy = a.size(2) - b.size(2)
    dx = a.size(3) - b.size(3)
    if dy or dx:
        a = torch.nn.functional.pad(a, (-(dx // 2), -(dx - dx // 2), -(dy // 2), -(dy - dy // 2)))
    return torch.cat((a, b), dim=1)


def to_device_layout(x, dtype=None):
    """Input images (N,C,H,W) -> channels-last tensor of the compute dtype.  A planar batch headed for the bf16 matrix
    kernels, whose channel count is not a multiple of 16 (the 14 bands), takes ONE pass of the library's layout kernel:
    cast, transposition and zero-padding to 16 channels at once; the result is the logical (N,C,H,W) slice of that
    padded buffer and carries the buffer itself for the first convolution (ops.conv_forward)."""
    if dtype == torch.bfloat16 and x.is_cuda and x.dim() == 4 and not x.requires_grad and x.is_contiguous() and \
            x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % 16 != 0 and x.shape[1] <= 48 and x.shape[0] <= 65535:
        return ops.planar_to_padded_nhwc(x, dtype)
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous(memory_format=torch.channels_last)


def plain_conv(conv, x, raw=False):
    """A plain (not standardized) stride-1 'same' nn.Conv2d through the library's own conv kernels, in x's dtype.
    For bf16 activations the filter is zero-padded to a multiple of 16 output channels so the MFMA kernels take
    it; the padding is sliced off again (autograd sees plain pad/slice ops)."""
    ops.require_gpu(x, "nn.Conv2d")
    w, b = conv.weight, conv.bias
    n_out = w.shape[0]
    if conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 or \
            tuple(conv.padding) != ((w.shape[2] - 1) // 2, (w.shape[3] - 1) // 2):
        raise NotImplementedError("plain_conv handles stride-1 'same' convolutions")
    pad = (-n_out) % 16 if x.dtype == torch.bfloat16 else 0
    if pad:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, pad))
        b = None if b is None else torch.nn.functional.pad(b, (0, pad))
    want_stats = bool(getattr(conv, "emit_stats", False)) and conv.training and not pad
    y = ops.PlainConvFn.apply(x, w, b, want_stats)
    if want_stats:
        ops.attach_stat(y)
    return y if (raw or not pad) else y[:, :n_out]


def head_conv(conv, x, raw=False):
    """The plain nn.Conv2d 1x1 classification head (modeling/UNet3p_SNws.py:93,158); returns float32 logits, or
    with raw=True the kernel's own output (compute dtype, channels padded to 16 for bf16) for the fused loss."""
    if raw:
        return plain_conv(conv, x, raw=True)
    return plain_conv(conv, x).float()


def finish_logits(raw16, n_class, raw=False):
    """What head_conv returns, from the fused op's raw logits (N, 16, H, W): as they are (raw=True) or the n_class
    real channels as float32."""
    return raw16 if raw else raw16[:, :n_class].float()


class PlainConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose forward runs the library's conv kernels.  emit_stats
    (set by link_conv_bn for a convolution whose only reader is a HipBatchNorm2d): the kernel's epilogue also sums the
    output's per-tile moments, so the normalisation does not re-read the tensor for them."""
    emit_stats = False

    def forward(self, x):
        return plain_conv(self, x)


class HipBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state_dict keys) on the library's streaming kernels; with
    fuse_relu the ReLU that follows it in the reference's Sequential runs in the same pass (put FusedAway() at the
    ReLU's index to keep the Sequential numbering).  Statistics are those of the local batch (nn.BatchNorm2d under
    the reference's DataParallel); sync_stats=True all-reduces them over the process group."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, fuse_relu=False,
                 sync_stats=False, clamp_eps=False):
        super(HipBatchNorm2d, self).__init__(num_features, eps, momentum, affine, track_running_stats)
        self.fuse_relu, self.sync_stats, self.clamp_eps = fuse_relu, sync_stats, clamp_eps
        self.process_group = None

    def forward(self, x, post=None):
        """post: optional (N, C) non-negative factors on the output per (image, channel): the mask of a Dropout2d that
        follows, applied in the same pass (see with_dropout)."""
        self._check_input_dim(x)
        ops.require_gpu(x, "BatchNorm2d")
        momentum = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:       # cumulative moving average
                momentum = 1.0 / float(self.num_batches_tracked)
        return ops.BatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, momentum, self.eps,
                                     self.training, self.fuse_relu, self.sync_stats, self.process_group, self.clamp_eps, post)

    def with_dropout(self, x, drop):
        """drop(self(x)) for an nn.Dropout2d `drop` (modeling/UNet3p_res_ocr_SNws.py:118-124) in one pass per direction:
        the mask is drawn as F.dropout2d draws it (one Bernoulli(1-p) per (image, channel) in x's dtype, divided by
        1-p, so a seeded run sees the same mask) and rides on the normalisation coefficients."""
        if not (drop.training and drop.p > 0.0):
            return self(x)
        if drop.p >= 1.0 or not ops.DROPOUT_FOLD:
            return drop(self(x))
        noise = x.new_empty((x.shape[0], x.shape[1], 1, 1)).bernoulli_(1.0 - drop.p).div_(1.0 - drop.p)
        return self(x, post=noise)


def link_conv_bn(root):
    """Marks every PlainConv2d that an nn.Sequential feeds straight into a HipBatchNorm2d (conv -> BN -> ReLU triples
    of the OCR head, modeling/UNet3p_res_ocr_SNws.py:39-73,118-123,269-274) so that it emits the moments the
    normalisation needs."""
    for m in root.modules():
        if isinstance(m, nn.Sequential):
            for a, b in zip(list(m)[:-1], list(m)[1:]):
                if isinstance(a, PlainConv2d) and isinstance(b, HipBatchNorm2d):
                    a.emit_stats = True
    return root


class FusedAway(nn.Identity):
    """Placeholder at the index of an activation that the preceding module already applied."""


class Resnet_BasicBlock(nn.Module):
    """conv1-bn1-relu-conv2-bn2, += block input, relu (modeling/UNet3p_res_ocr_SNws.py:134-161) as two fused ops:
    ConvSnRelu and conv + SwitchNorm + residual add + ReLU."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Resnet_BasicBlock, self).__init__()
        self.conv1 = ws.Conv2dws(inplanes, planes, kernel_size=3, stride=1, padding=1)
        self.bn1 = sn.SwitchNorm2d(planes)
        self.conv2 = ws.Conv2dws(planes, planes, kernel_size=3, stride=1, padding=1)
        self.bn2 = sn.SwitchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        ops.require_gpu(x, "Resnet_BasicBlock")
        mf = x.dtype == torch.bfloat16
        c1, n1, c2, n2 = self.conv1, self.bn1, self.conv2, self.bn2
        if self.downsample is None:
            x = ops.share_grad(x)       # read by conv1 and by the residual add: one gradient tensor, summed in place
        t = ops.tag_sn_output(ops.ConvSnReluFn.apply(x, c1.weight, c1.bias, n1.weight, n1.bias, n1.mean_weight,
                                                     n1.var_weight, c1.prepared(mf and not c1.force_direct),
                                                     ops.SNParams(n1), True, None))
        if self.downsample is not None:
            # the reference feeds conv2's OUTPUT to downsample (:156); kept, unfused
            y = ops.Conv2dwsFn.apply(t, c2.weight, c2.bias, c2.prepared(mf and not c2.force_direct), c2.force_direct)
            out = ops.SwitchNormFn.apply(y, n2.weight, n2.bias, n2.mean_weight, n2.var_weight, ops.SNParams(n2), False)
            return torch.relu(out + self.downsample(y))
        return ops.ConvSnAddReluFn.apply(t, c2.weight, c2.bias, n2.weight, n2.bias, n2.mean_weight, n2.var_weight,
                                         c2.prepared(mf and not c2.force_direct), ops.SNParams(n2), x)


class PoolPyramid:
    """Max-pooled copies of the encoder maps, each computed once and by halving: pool_2k(x) = pool_2(pool_k(x))
    (the maximum over a window is the maximum over its sub-windows' maxima, ceil-mode borders included), so the
    factor-4 and factor-8 skips read a map 4x / 16x smaller than the source, the encoder's own down-sampling and
    the decoder's factor-2 skip share one tensor, and in backward every source receives ONE pooled gradient instead
    of one per consumer.  Gradient routing: where a window holds several equal maxima the chain of 2x2 choices may
    pick a different (equal) element than a single k x k scan; all-zero post-ReLU windows route to the first
    element either way."""

    def __init__(self, enc):
        self.enc = enc
        self.maps = {}

    def get(self, src, k):
        if k == 1:
            return self.enc[src]
        key = (src, k)
        if key not in self.maps:
            # (several consumers: the next pool, a decoder skip block, an encoder stage - their gradients meet in one tensor)
            self.maps[key] = ops.share_grad(ops.max_pool_ceil(self.get(src, k // 2), 2))
        return self.maps[key]


def full_scale_decoder(m, enc, pools=None, head=None):
    """The UNet3+ full-scale skip decoder (modeling/UNet3p_SNws.py:127-156, UNet3p_res_ocr_SNws.py:327-361) over
    the encoder maps enc[1..5]; m holds convd{1..4}_cat, conv5_cat, convu_cat, conv_fusion.  Returns the finest
    decoder map - or, with `head` (the 1x1 classification conv that consumes it), (raw logits, True) when the last
    fusion block and the head can run as one fused op, else (finest decoder map, False).  Every up-sampled source goes through conv(upsample(x)) = sum_t up(W_t x): the 1x1 half
    Z = W_taps x is computed once per source at ITS resolution and shared by all the scales that consume it."""
    enc_cat = {1: m.convd1_cat, 2: m.convd2_cat, 3: m.convd3_cat, 4: m.convd4_cat}
    if pools is None:
        pools = PoolPyramid(enc)
    z5 = ops.share_grad(m.conv5_cat.taps(enc[5]))      # one tap map, consumed by every decoder scale
    zdec = {}
    dec = {}
    f = m.Cat_channels
    for level in (4, 3, 2, 1):
        # the five 64-channel blocks write straight into their channel slice of the fusion input: no torch.cat
        ref = enc[level]
        buf = ops.new_nhwc(ref.shape[0], 5 * f, ref.shape[2], ref.shape[3], ref.dtype, ref.device)
        parts = []
        # data-parallel runs: the five branch blocks are independent, so their SwitchNorm statistics travel in ONE
        # all-reduce (their outputs are complete when the block ends; nothing reads them before the concatenation)
        with ops.stat_batch(getattr(m.conv_fusion[1], "process_group", None)):
            for src in (1, 2, 3, 4):
                out = buf[:, len(parts) * f:(len(parts) + 1) * f]
                if src < level:         # encoder map, max-pooled down to this level
                    parts.append(enc_cat[src](pools.get(src, 2 ** (level - src)), out=out))
                elif src == level:
                    parts.append(enc_cat[src](enc[src], out=out))
                else:                   # decoder map of a coarser level, up-sampled
                    if src not in zdec:
                        zdec[src] = ops.share_grad(m.convu_cat.taps(dec[src]))
                    parts.append(m.convu_cat.up(zdec[src], 2 ** (src - level), out=out))
            # concat order: encoder sources fine->coarse, then decoder sources fine->coarse, then the centre
            parts.append(m.conv5_cat.up(z5, 2 ** (5 - level), out=buf[:, 4 * f:5 * f]))
        fused_in = ops.assemble(buf, parts)
        if level == 1 and head is not None:
            if m.conv_fusion.head_fusable(fused_in, head):
                return m.conv_fusion.with_head(fused_in, head), True
            return m.conv_fusion(fused_in), False
        dec[level] = m.conv_fusion(fused_in)
    return dec[1]
