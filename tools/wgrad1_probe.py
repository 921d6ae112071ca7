#!/usr/bin/env python3
"""Times snws_conv_mfma_wgrad on the 1x1 shapes of the two UNet3+ models (batch 16, 512^2 input)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from snws import _lib as L
from snws import ops

N = 16
for cin, cout, S in ((512, 256, 512), (256, 256, 512), (256, 512, 512), (1024, 512, 512), (320, 320, 512),
                     (320, 576, 256), (320, 576, 128), (320, 576, 64), (1024, 576, 32), (512, 16, 512), (320, 16, 512)):
    x = torch.randn(N, S, S, cin, device="cuda:0").to(torch.bfloat16)
    dy = torch.randn(N, S, S, cout, device="cuda:0").to(torch.bfloat16)
    g = torch.zeros(cout, 1, 1, cin, device="cuda:0")
    cfg = ops.ConvCfg((cout, cin, 1, 1), 1, 0, 1, 1)
    d = cfg.desc(N, S, S, cin, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        L.call("snws_conv_mfma_wgrad", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(), None, st)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    fl = 2.0 * N * S * S * cin * cout
    gb = 2.0 * N * S * S * (cin + cout)
    print("wgrad 1x1 %4d->%4d @%3d^2: %.3f ms  %6.0f TFLOP/s  operands at %.2f TB/s" % (cin, cout, S, t, fl / t / 1e9, gb / t / 1e9), flush=True)
