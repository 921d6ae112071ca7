"""UNet with weight-standardized convs and SwitchNorm - counterpart of the reference's
modeling/UNet_SNws.py:42-152 (same constructor, state_dict names/shapes and arithmetic), built from the fused
ConvSnRelu block and the resample kernels."""
import torch
import torch.nn as nn

from modeling.blocks import ConvSnRelu, crop_concat, finish_logits, head_conv, to_device_layout
from snws import ops


class UNet_SNws(nn.Module):
    def __init__(self, n_channels, n_filters, n_class, using_movavg, using_bn):
        super(UNet_SNws, self).__init__()
        f = n_filters
        kw = dict(using_movavg=using_movavg, using_bn=using_bn)

        def block(cin, cout, k=3):
            return ConvSnRelu(cin, cout, kernel_size=(k, k), padding=(k - 1) // 2, **kw)
        self.conv1_1, self.conv1_2 = block(n_channels, f), block(f, f)
        self.conv2_1, self.conv2_2 = block(f, 2 * f), block(2 * f, 2 * f)
        self.conv3_1, self.conv3_2 = block(2 * f, 4 * f), block(4 * f, 4 * f)
        self.conv4_1, self.conv4_2 = block(4 * f, 8 * f), block(8 * f, 8 * f)
        self.conv_d1, self.conv_d2 = block(8 * f, 16 * f), block(16 * f, 16 * f)
        self.conv_u1, self.conv_u2, self.conv_u3 = block(16 * f, 8 * f, 1), block(16 * f, 8 * f), block(8 * f, 8 * f)
        self.conv5_1, self.conv5_2, self.conv5_3 = block(8 * f, 4 * f, 1), block(8 * f, 4 * f), block(4 * f, 4 * f)
        self.conv6_1, self.conv6_2, self.conv6_3 = block(4 * f, 2 * f, 1), block(4 * f, 2 * f), block(2 * f, 2 * f)
        self.conv7_1, self.conv7_2, self.conv7_3 = block(2 * f, f, 1), block(2 * f, f), block(f, f)
        self.output_seg_map = nn.Conv2d(f, n_class, kernel_size=(1, 1), stride=1, padding=0)
        self.compute_dtype = None
        self.raw_logits = False        # True: return the head kernel's own (padded, compute-dtype) output

    def forward(self, x):
        x = to_device_layout(x, self.compute_dtype)
        skips = []
        for a, b in ((self.conv1_1, self.conv1_2), (self.conv2_1, self.conv2_2), (self.conv3_1, self.conv3_2),
                     (self.conv4_1, self.conv4_2)):
            x = b(a(x))
            skips.append(x)
            x = ops.max_pool_ceil(x, 2)
        x = self.conv_d2(self.conv_d1(x))
        for (up1, c2, c3) in ((self.conv_u1, self.conv_u2, self.conv_u3), (self.conv5_1, self.conv5_2, self.conv5_3),
                              (self.conv6_1, self.conv6_2, self.conv6_3), (self.conv7_1, self.conv7_2, self.conv7_3)):
            x = up1.on_upsampled(x, 2)       # 1x1 conv of the x2 up-sampled map, computed at low resolution
            x = c2(crop_concat(skips.pop(), x))
            if c3 is self.conv7_3 and c3.head_fusable(x, self.output_seg_map):
                # the last block and the classification head as one fused op (bf16 tier, 64 channels)
                return finish_logits(c3.with_head(x, self.output_seg_map), self.output_seg_map.out_channels,
                                     raw=self.raw_logits)
            x = c3(x)
        return head_conv(self.output_seg_map, x, raw=self.raw_logits)
