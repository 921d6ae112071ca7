#!/usr/bin/env python3
"""Small target for rocprofv3 --pmc passes: runs ONE family of kernels a few times on the headline shapes.
    python tools/pmc_target.py conv|wgrad|sn|hist|upconv"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, _p)
import torch  # noqa: E402
from snws import ops  # noqa: E402
import modeling.weightstd as ws  # noqa: E402
from modeling.blocks import ConvSnRelu  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "conv"
N, DEV = 16, "cuda:0"
torch.manual_seed(0)
if what in ("conv", "wgrad"):
    conv = ws.Conv2dws(320, 320, 3, 1, 1).to(DEV)
    x = torch.randn(N, 512, 512, 320, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = torch.randn(N, 512, 512, 320, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
    prep = conv.prepared(True)
    y, _, saved = ops.conv_forward(x, prep, conv.bias, want_stats=True)
    for _ in range(3):
        if what == "conv":
            ops.conv_forward(x, prep, conv.bias, want_stats=True)
        else:
            ops.conv_backward(dy, prep, saved, conv.weight, False, True, False)   # no bias gradient, as inside a Conv2dws -> SwitchNorm block
elif what == "sn":
    blk = ConvSnRelu(64, 64).to(DEV).train()
    x = torch.randn(N, 512, 512, 64, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = torch.randn(N, 512, 512, 64, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
    p = ops.SNParams(blk[1])
    for _ in range(3):
        y, saved = ops.sn_forward(x, p, True)
        ops.sn_backward(dy, saved)
elif what == "upconv":
    # conv(upsample_k(x)) for every scale of the decoder, 512^2 outputs, forward and backward
    blk = ConvSnRelu(320, 64).to(DEV).train()
    for k in (2, 4, 8, 16):
        x = torch.randn(N, 512 // k, 512 // k, 320, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
        for _ in range(2):
            y = blk.on_upsampled(x, k)
            y.backward(torch.ones_like(y))
elif what == "hist":
    from utils.metrics import Evaluator
    gt = torch.randint(0, 10, (16, 2748, 2748), device=DEV).float()
    pr = torch.randint(0, 10, (16, 2748, 2748), device=DEV)
    ev = Evaluator(10)
    for _ in range(3):
        ev.add_batch(gt, pr)
torch.cuda.synchronize()
