#!/usr/bin/env python3
"""What the vendor GEMM reaches on this box for large bf16 products (random and all-zero operands): the practical
matrix-core ceiling next to which the convolution kernels' TFLOP/s are read.   python tools/mm_ceiling.py"""
import torch

dev = torch.device("cuda", 0)


def rate(m, n, k, zero):
    a = torch.zeros(m, k, device=dev, dtype=torch.bfloat16) if zero else torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = torch.zeros(n, k, device=dev, dtype=torch.bfloat16) if zero else torch.randn(n, k, device=dev).to(torch.bfloat16)
    for _ in range(3):
        torch.mm(a, b.t())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    reps = 20
    e0.record()
    for _ in range(reps):
        torch.mm(a, b.t())
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * m * n * k * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


for m, n, k in ((8192, 8192, 8192), (16384, 8192, 4096), (4194304, 320, 2880), (65536, 2048, 2048)):
    print("%8d x %5d x %5d   random %7.1f TFLOP/s   zeros %7.1f TFLOP/s" % (m, n, k, rate(m, n, k, False), rate(m, n, k, True)), flush=True)

# the library's own 3x3 kernels on the 320 -> 320 layer at 512^2, batch 16, with the same two kinds of operands
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semantic-segmentation_amd"))
from snws import _lib as L  # noqa: E402
from snws import ops  # noqa: E402

N, S, C = 16, 512, 320
cfg = ops.ConvCfg((C, C, 3, 3), 1, 1, 1, 1)
d = cfg.desc(N, S, S, C, C, C)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nbytes = L.lib().snws_conv_mfma_wgrad_workspace_bytes(ctypes.byref(d))
ws = torch.empty(nbytes // 4, device=dev)
g = torch.zeros(C, 3, 3, C, device=dev)
db = torch.zeros(C, device=dev)
y = torch.empty(N, S, S, C, device=dev, dtype=torch.bfloat16)
for zero in (False, True):
    mk = (lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16)) if zero else (lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16))
    x, dy = mk(N, S, S, C), mk(N, S, S, C)
    prep = ops.PreparedWeight(mk(C, C, 3, 3).float(), cfg, True, standardize=False)
    out = []
    runs = {"wgrad": lambda: L.call("snws_conv_mfma_wgrad_det", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(),
                                    db.data_ptr(), ws.data_ptr(), nbytes, st)}
    runs["forward"] = lambda: L.call("snws_conv_mfma_fwd", ctypes.byref(d), x.data_ptr(), prep.pack_fwd.data_ptr(), None,
                                     y.data_ptr(), None, st)
    for name, run in runs.items():
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        out.append("%s %.0f TFLOP/s" % (name, 2.0 * 9 * N * S * S * C * C * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e12))
    print("libsnws 3x3 320->320 @512^2, %s operands: %s" % ("zero" if zero else "random", ", ".join(out)), flush=True)
