#!/usr/bin/env python3
"""Library GEMM rates (torch.matmul -> hipBLASLt/rocBLAS) on the 1x1-convolution shapes of the models, bf16."""
import torch
dev = "cuda:0"
def bench(fn, flops, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-46s %.3f ms  %.0f TFLOP/s" % (name, ms, flops / ms / 1e9))
for (P, ci, co, tag) in ((16 * 256 * 256, 320, 576, "taps 320->576 @256^2"), (16 * 128 * 128, 320, 576, "taps @128^2"),
                         (8 * 512 * 512, 512, 256, "ocr 512->256 @512^2 N=8"), (8 * 512 * 512, 1024, 512, "ocr 1024->512"),
                         (8 * 512 * 512, 320, 320, "aux 320->320")):
    x = torch.randn(P, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(P, co, device=dev).to(torch.bfloat16)
    fl = 2.0 * P * ci * co
    bench(lambda: torch.matmul(x, w.t()), fl, tag + " fwd  x@W^T")
    bench(lambda: torch.matmul(dy, w), fl, tag + " dgrad dy@W")
    bench(lambda: torch.matmul(dy.t(), x), fl, tag + " wgrad dy^T@x")
    bench(lambda: torch.matmul(dy.t().float() if False else dy.t(), x).float(), fl, tag + " wgrad (+f32 cast)")
