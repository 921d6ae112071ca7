#!/usr/bin/env python3
"""Times snws_upconv_bwd (conv(upsample_k(x)) backward: dy (N, H, W, 64) -> dZ (N, h, w, 576)) on the decoder's shapes,
batch 16.  SNWS_UPCONV_BWD_ROWS=1/4 forces one / four source rows per thread in the separable kernel (k = 8, 16)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from snws import _lib as L

N, Co = 16, 64
for k, S in ((2, 512), (4, 512), (8, 512), (8, 256), (16, 512)):
    h = w = S // k
    dy = torch.randn(N, S, S, Co, device="cuda").to(torch.bfloat16)
    dZ = torch.empty(N, h, w, 9 * Co, device="cuda", dtype=torch.bfloat16)

    def run():
        L.call("snws_upconv_bwd", dy.data_ptr(), Co, dZ.data_ptr(), 9 * Co, 1, N, h, w, Co, k, 3,
               torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    alg = dy.numel() * 2 + dZ.numel() * 2
    print("upconv bwd k=%2d out %3d^2: %7.1f us  algorithmic %6.1f MB -> %5.2f TB/s" % (k, S, us, alg / 1e6, alg / us / 1e6), flush=True)
