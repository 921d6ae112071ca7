#!/usr/bin/env python3
"""Times the forward / backward of conv(upsample_k(x)) (tap map -> 64-channel output, batch 16) for every scale of the
decoder, at the output sizes the model uses.  SNWS_UPCONV_MFMA=0 selects the VALU form of the forward.
    python tools/upconv_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402
from modeling.blocks import ConvSnRelu  # noqa: E402

DEV, N = "cuda:0", 16
blk = ConvSnRelu(320, 64).to(DEV).train()
for k, sizes in ((2, (512, 256, 128, 64)), (4, (512, 256, 128)), (8, (512, 256)), (16, (512,))):
    for S in sizes:
        z = torch.randn(N, S // k, S // k, 576, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
        g = None
        res = []

        def fwd():
            return blk.up(z, k)

        def both():
            fwd().backward(g)
        g = torch.randn_like(fwd())
        for fn in (fwd, both):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 5 * 1e3)
        px = N * S * S
        floor = (px * 128 + px // (k * k) * 1152) / 5.3e12 * 1e6
        print("k=%2d out %3d^2: forward (incl. SwitchNorm + ReLU) %7.1f us, forward + backward %7.1f us; up-conv HBM floor per direction %6.1f us"
              % (k, S, res[0], res[1], floor), flush=True)
