"""UNet3+ with weight-standardized convs and SwitchNorm - the build's counterpart of the reference's
modeling/UNet3p_SNws.py:42-160: same constructor, same parameter / buffer names and shapes (a reference
checkpoint loads with load_state_dict), same arithmetic, but every Conv2dws -> SwitchNorm2d -> ReLU block runs
as one fused HIP op and the full-scale skips (max-pool / bilinear up-sampling) run through the resample kernels.
The seven decoder blocks are shared by the four decoder scales, as in the reference (17 blocks, 34 calls).
"""
import torch
import torch.nn as nn

from modeling.blocks import ConvSnRelu, PoolPyramid, finish_logits, full_scale_decoder, head_conv, to_device_layout
from snws import ops


class UNet3p_SNws(nn.Module):
    def __init__(self, n_channels, n_filters, n_class, using_movavg, using_bn):
        super(UNet3p_SNws, self).__init__()
        f = n_filters
        kw = dict(using_movavg=using_movavg, using_bn=using_bn)
        # encoder: two blocks per scale, widths f, 2f, 4f, 8f, 16f
        widths = [n_channels, f, 2 * f, 4 * f, 8 * f, 16 * f]
        enc_names = ['convd1', 'convd2', 'convd3', 'convd4', 'conv5']
        for i, name in enumerate(enc_names):
            setattr(self, name + '_1', ConvSnRelu(widths[i], widths[i + 1], **kw))
            setattr(self, name + '_2', ConvSnRelu(widths[i + 1], widths[i + 1], **kw))
        # decoder: one "to 64 channels" block per source, one fusion block, all shared across scales
        self.Cat_channels = f
        self.Fusion_channels = 5 * f
        for name, cin in (('convd1_cat', f), ('convd2_cat', 2 * f), ('convd3_cat', 4 * f), ('convd4_cat', 8 * f),
                          ('conv5_cat', 16 * f), ('convu_cat', 5 * f)):
            setattr(self, name, ConvSnRelu(cin, f, **kw))
        self.conv_fusion = ConvSnRelu(5 * f, 5 * f, **kw)
        self.output_seg_map = nn.Conv2d(5 * f, n_class, kernel_size=1, stride=1, padding=0)
        self.compute_dtype = None      # None: keep the input's dtype; torch.bfloat16 for the MFMA tier
        self.raw_logits = False        # True: return the head kernel's own (padded, compute-dtype) output

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
        # the last fusion block and the classification head run as one fused op when they can (bf16, 64 / 320 channels)
        out, fused = full_scale_decoder(self, enc, pools, head=self.output_seg_map)
        if fused:
            return finish_logits(out, self.output_seg_map.out_channels, raw=self.raw_logits)
        return head_conv(self.output_seg_map, out, raw=self.raw_logits)
