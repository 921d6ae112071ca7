"""UNet3+ with residual encoder blocks and an OCR (object-contextual representation) head - the build's counterpart
of the reference's modeling/UNet3p_res_ocr_SNws.py:10-377: same class names, constructors, parameter / buffer names,
shapes and registration order (a reference checkpoint loads with load_state_dict), same arithmetic.

What runs where: every Conv2dws -> SwitchNorm2d -> ReLU block, the residual blocks' conv + SwitchNorm + add + ReLU,
the full-scale skip decoder and all plain nn.Conv2d layers (1x1 and the 320 -> 512 3x3) go through the library's
HIP kernels, and so do the OCR head's BatchNorm2d + ReLU pairs (blocks.HipBatchNorm2d; statistics of the local
batch, as under the reference's DataParallel) and, in the bf16 tier, its two soft-max + batched-matmul attention
steps (ops.OcrGatherFn / ops.OcrAttendFn; ops.OcrGather32Fn / ops.OcrAttend32Fn in the float32 tier); the Dropout2d after the
last BatchNorm2d + ReLU rides on that pass's coefficients (HipBatchNorm2d.with_dropout).  No attention step falls back
to stock torch ops: unsupported inputs raise.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.conv import _ConvNd
from torch.nn.modules.utils import _pair

import modeling.snorm as sn
import modeling.weightstd as ws
from modeling.blocks import (ConvSnRelu, FusedAway, HipBatchNorm2d, PlainConv2d, PoolPyramid, Resnet_BasicBlock,  # noqa: F401
                             link_conv_bn,
                             crop_concat as CropConcat,
                             full_scale_decoder, head_conv, plain_conv, to_device_layout)
from snws import ops


class SpatialGather_Module(nn.Module):
    """Soft object regions: class-probability-weighted average of the pixel features
    (modeling/UNet3p_res_ocr_SNws.py:10-28).  The soft-max over pixels runs in float32."""

    def __init__(self, cls_num=0, scale=1):
        super(SpatialGather_Module, self).__init__()
        self.cls_num = cls_num
        self.scale = scale

    def forward(self, feats, probs):
        B, K = probs.size(0), probs.size(1)
        if ops.ocr_gather_ok(feats, probs):
            return ops.OcrGatherFn.apply(feats, probs, float(self.scale))               # B x C x K x 1
        if ops.ocr32_ok(feats, K):
            return ops.OcrGather32Fn.apply(feats, probs.float(), float(self.scale))
        raise RuntimeError("SpatialGather_Module runs on the HIP kernels only: bf16 features with %d channels and <= 16 "
                           "classes, or float32 GPU tensors with <= 64 classes (got %s %s, %d classes)"
                           % (ops.OCR_GATHER_C, feats.dtype, tuple(feats.shape), K))


class _ObjectAttentionBlock(nn.Module):
    """Pixel-to-object-region attention (modeling/UNet3p_res_ocr_SNws.py:31-104)."""

    def __init__(self, in_channels, key_channels, scale=1):
        super(_ObjectAttentionBlock, self).__init__()
        self.scale = scale
        self.in_channels = in_channels
        self.key_channels = key_channels
        self.pool = nn.MaxPool2d(kernel_size=(scale, scale))

        def cbr(cin, cout):
            return [PlainConv2d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False),
                    HipBatchNorm2d(cout, fuse_relu=True), FusedAway()]
        self.f_pixel = nn.Sequential(*(cbr(in_channels, key_channels) + cbr(key_channels, key_channels)))
        self.f_object = nn.Sequential(*(cbr(in_channels, key_channels) + cbr(key_channels, key_channels)))
        self.f_down = nn.Sequential(*cbr(in_channels, key_channels))
        self.f_up = nn.Sequential(*cbr(key_channels, in_channels))

    def forward(self, x, proxy):
        B, h, w = x.size(0), x.size(2), x.size(3)
        if self.scale > 1:
            x = self.pool(x)
        proxy = proxy.contiguous(memory_format=torch.channels_last)
        q4 = self.f_pixel(x)
        if ops.ocr_attend_ok(q4, proxy):
            context = ops.OcrAttendFn.apply(q4, self.f_object(proxy), self.f_down(proxy), self.key_channels ** -.5)
            context = self.f_up(context)
            if self.scale > 1:
                context = F.interpolate(input=context, size=(h, w), mode='bilinear', align_corners=True)
            return context
        if not ops.ocr32_ok(q4, proxy.shape[2]):
            raise RuntimeError("_ObjectAttentionBlock runs on the HIP kernels only: bf16 queries with %d channels and <= 16 "
                               "classes, or float32 GPU tensors with <= 64 classes (got %s %s, %d classes)"
                               % (ops.OCR_ATTEND_C, q4.dtype, tuple(q4.shape), proxy.shape[2]))
        context = ops.OcrAttend32Fn.apply(q4, self.f_object(proxy), self.f_down(proxy), self.key_channels ** -.5)
        context = self.f_up(context)
        if self.scale > 1:
            context = F.interpolate(input=context, size=(h, w), mode='bilinear', align_corners=True)
        return context


class ObjectAttentionBlock2D(_ObjectAttentionBlock):
    def __init__(self, in_channels, key_channels, scale=1):
        super(ObjectAttentionBlock2D, self).__init__(in_channels, key_channels, scale)


class SpatialOCR_Module(nn.Module):
    """Object context + pixel features -> 1x1 conv, BN, ReLU, Dropout2d (modeling/UNet3p_res_ocr_SNws.py:112-132)."""

    def __init__(self, in_channels, key_channels, out_channels, scale=1, dropout=0.1):
        super(SpatialOCR_Module, self).__init__()
        self.object_context_block = ObjectAttentionBlock2D(in_channels, key_channels, scale)
        self.conv_bn_dropout = nn.Sequential(
            PlainConv2d(2 * in_channels, out_channels, kernel_size=1, padding=0, bias=False),
            HipBatchNorm2d(out_channels, fuse_relu=True),
            FusedAway(),
            nn.Dropout2d(dropout))

    def grad_share_ok(self, feats):
        """True when every op of this module that reads `feats` takes part in shared-gradient accumulation
        (ops.share_grad): f_pixel's first convolution and the two-product form of the 1x1 convolution below."""
        return self.object_context_block.scale == 1 and ops.conv1x1_pair_ok(feats, feats, self.conv_bn_dropout[0])

    def forward(self, feats, proxy_feats):
        context = self.object_context_block(feats, proxy_feats)
        conv, bn, _, drop = self.conv_bn_dropout
        if ops.conv1x1_pair_ok(context, feats, conv):
            # conv(cat([context, feats])) = W[:, :C] context + W[:, C:] feats: the concatenation is never built
            want_stats = conv.emit_stats and conv.training
            y = ops.Conv1x1PairFn.apply(context, feats, conv.weight, want_stats)
            return bn.with_dropout(ops.attach_stat(y) if want_stats else y, drop)
        return bn.with_dropout(conv(torch.cat([context, feats], 1)), drop)


class GatedSpatialConv2d(_ConvNd):
    """Gated convolution of modeling/UNet3p_res_ocr_SNws.py:163-196 (defined by the reference, not used by its
    model).  The gate runs on the library's SwitchNorm / Conv2dws ops; the final convolution is a stride-1 'same'
    convolution through plain_conv."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, groups=1, bias=False):
        super(GatedSpatialConv2d, self).__init__(in_channels, out_channels, _pair(kernel_size), _pair(stride),
                                                 _pair(padding), _pair(dilation), False, _pair(0), groups, bias, 'zeros')
        self.gate_conv = nn.Sequential(
            sn.SwitchNorm2d(in_channels + 1),
            ws.Conv2dws(in_channels + 1, in_channels + 1, 1),
            nn.ReLU(),
            ws.Conv2dws(in_channels + 1, 1, 1),
            sn.SwitchNorm2d(1),
            nn.Sigmoid())

    def forward(self, input_features, gating_features):
        alphas = self.gate_conv(torch.cat([input_features, gating_features], dim=1))
        return plain_conv(self, input_features * (alphas + 1))

    def reset_parameters(self):
        nn.init.xavier_normal_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class UNet3p_res_ocr_SNws(nn.Module):
    def __init__(self, n_channels, n_filters, n_class, using_movavg, using_bn):
        super(UNet3p_res_ocr_SNws, self).__init__()
        f = n_filters
        kw = dict(using_movavg=using_movavg, using_bn=using_bn)
        widths = [n_channels, f, 2 * f, 4 * f, 8 * f, 16 * f]
        for i, name in enumerate(['convd1', 'convd2', 'convd3', 'convd4', 'conv5']):
            setattr(self, name + '_1', ConvSnRelu(widths[i], widths[i + 1], **kw))
            setattr(self, name + '_2', Resnet_BasicBlock(widths[i + 1], widths[i + 1], stride=1, downsample=None))

        ocr_mid_channels = 512
        ocr_key_channels = 256
        self.aux_head = nn.Sequential(
            PlainConv2d(5 * f, 5 * f, kernel_size=1, stride=1, padding=0),
            HipBatchNorm2d(5 * f, fuse_relu=True),
            FusedAway(),
            PlainConv2d(5 * f, n_class, kernel_size=1, stride=1, padding=0, bias=True))
        self.conv3x3_ocr = nn.Sequential(
            PlainConv2d(5 * f, ocr_mid_channels, kernel_size=3, stride=1, padding=1),
            HipBatchNorm2d(ocr_mid_channels, fuse_relu=True),
            FusedAway())
        self.ocr_gather_head = SpatialGather_Module(n_class)
        self.ocr_distri_head = SpatialOCR_Module(in_channels=ocr_mid_channels, key_channels=ocr_key_channels,
                                                 out_channels=ocr_mid_channels, scale=1, dropout=0.05)
        self.cls_head = nn.Conv2d(ocr_mid_channels, n_class, kernel_size=1, stride=1, padding=0, bias=True)

        self.Cat_channels = f
        self.Fusion_channels = 5 * f
        for name, cin in (('convd1_cat', f), ('convd2_cat', 2 * f), ('convd3_cat', 4 * f), ('convd4_cat', 8 * f),
                          ('conv5_cat', 16 * f), ('convu_cat', 5 * f)):
            setattr(self, name, ConvSnRelu(cin, f, **kw))
        self.conv_fusion = ConvSnRelu(5 * f, 5 * f, **kw)
        self.output_seg_map = nn.Conv2d(f, n_class, kernel_size=1, stride=1, padding=0)   # registered, unused (:294)
        self.compute_dtype = None      # None: keep the input's dtype; torch.bfloat16 for the MFMA tier
        self.raw_logits = False        # True: return the head kernel's own (padded, compute-dtype) output
        link_conv_bn(self)             # conv -> BatchNorm pairs: the conv's epilogue sums the moments

    def forward(self, x):
        x = to_device_layout(x, self.compute_dtype)
        enc = {}
        pools = PoolPyramid(enc)
        t = x
        for lvl, name in enumerate(['convd1', 'convd2', 'convd3', 'convd4', 'conv5'], start=1):
            if lvl > 1:
                t = pools.get(lvl - 1, 2)       # shared with the decoder's factor-2 skip of the same map
            t = getattr(self, name + '_2')(getattr(self, name + '_1')(t))
            # encoder maps 1..4 feed the next stage's pool AND the decoder's skips: one shared gradient tensor
            enc[lvl] = t = ops.share_grad(t) if lvl < 5 else t
        # the finest decoder map feeds the auxiliary head and the OCR branch; the OCR features feed the region
        # gather, the pixel branch of the attention and the final 1x1 convolution: one gradient tensor each
        convu1 = ops.share_grad(full_scale_decoder(self, enc, pools))
        out_aux = self.aux_head(convu1)
        feats = self.conv3x3_ocr(convu1)
        if ops.ocr_gather_ok(feats, out_aux) and self.ocr_distri_head.grad_share_ok(feats):
            feats = ops.share_grad(feats)
        context = self.ocr_gather_head(feats, out_aux)
        feats = self.ocr_distri_head(feats, context)
        return head_conv(self.cls_head, feats, raw=self.raw_logits)
