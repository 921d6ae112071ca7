#!/usr/bin/env python3
"""Times snws_sn_apply / snws_sn_bwd_apply on the SwitchNorm shapes of UNet3p_SNws (batch 16, 512^2 input, bf16): GB/s of
the algorithmic 2 / 3 tensor passes per shape.  Buffers are rotated so that no call finds its operands in the
256 MB Infinity Cache.   python tools/sn_probe.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from snws import _lib as L

N = 16
lib = L.lib()
dev = "cuda:0"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for C, S in ((64, 512), (320, 256), (64, 256), (128, 256), (320, 128), (64, 128), (256, 128), (320, 64), (64, 64), (512, 64), (1024, 32)):
    HW = S * S
    nbytes = N * HW * C * 2
    R = max(2, int(600e6 // (3 * nbytes)) + 1)      # rotate > 600 MB of operands
    xs = [torch.randn(N, S, S, C, device=dev).to(torch.bfloat16) for _ in range(R)]
    dys = [torch.randn(N, S, S, C, device=dev).to(torch.bfloat16) for _ in range(R)]
    ys = [torch.empty(N, S, S, C, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    ctx = torch.randn(lib.snws_sn_ctx_floats(N, C), device=dev).abs() + 0.5
    ctxb = torch.randn(lib.snws_sn_ctxb_floats(N, C), device=dev) * 0.01
    beta = torch.randn(C, device=dev)

    def fwd(i):
        L.call("snws_sn_apply", xs[i].data_ptr(), C, ys[i].data_ptr(), C, None, 0, L.SNWS_BF16, ctx.data_ptr(), beta.data_ptr(),
               N, HW, C, 1, st())

    def bwd(i):
        L.call("snws_sn_bwd_apply", dys[i].data_ptr(), C, xs[i].data_ptr(), C, None, 0, ys[i].data_ptr(), C, L.SNWS_BF16,
               ctx.data_ptr(), ctxb.data_ptr(), beta.data_ptr(), N, HW, C, 1, st())
    out = []
    for fn, passes in ((fwd, 2), (bwd, 3)):
        for i in range(R):
            fn(i)
        torch.cuda.synchronize()
        reps = 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for i in range(R):
                fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (reps * R) * 1e3
        out.append((us, passes * nbytes / us / 1e3))
    print("C %4d @%3d^2 (%6.1f MB/tensor, %2d rotating): apply %7.1f us %5.0f GB/s | bwd_apply %7.1f us %5.0f GB/s"
          % (C, S, nbytes / 1e6, R, out[0][0], out[0][1], out[1][0], out[1][1]), flush=True)
    del xs, dys, ys
