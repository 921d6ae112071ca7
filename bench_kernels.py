#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the BASELINE shapes (HIP events, random bf16 data):
    python bench_kernels.py [conv|sn|upconv|all] [--batch 16]
Prints TFLOP/s (conv) or GB/s of algorithmic bytes (memory-bound kernels) per shape."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

from snws import ops  # noqa: E402
import modeling.weightstd as ws  # noqa: E402
from modeling.blocks import ConvSnRelu  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def bench_conv(N):
    shapes = [(320, 320, 512, 3), (320, 320, 256, 3), (64, 64, 512, 3), (128, 128, 256, 3), (256, 256, 128, 3),
              (512, 512, 64, 3), (1024, 1024, 32, 3), (128, 64, 256, 3), (320, 576, 256, 1), (1024, 576, 32, 1)]
    for cin, cout, S, k in shapes:
        conv = ws.Conv2dws(cin, cout, k, 1, (k - 1) // 2).to(DEV)
        x = torch.randn(N, S, S, cin, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
        dy = torch.randn(N, S, S, cout, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
        prep = conv.prepared(True)
        y, _, saved = ops.conv_forward(x, prep, conv.bias, want_stats=True)
        fl = 2.0 * N * S * S * cin * cout * k * k
        tf = timed(lambda: ops.conv_forward(x, prep, conv.bias, want_stats=True))
        td = timed(lambda: ops.conv_backward(dy, prep, saved, conv.weight, True, False, False))
        tw = timed(lambda: ops.conv_backward(dy, prep, saved, conv.weight, False, True, True))
        print("conv %4d->%4d %dx%d @%3d^2 N=%d: fwd %7.1f TF/s (%.3f ms)  dgrad %7.1f  wgrad(+ws_bwd) %7.1f"
              % (cin, cout, k, k, S, N, fl / tf / 1e12, tf * 1e3, fl / td / 1e12, fl / tw / 1e12), flush=True)


def bench_sn(N):
    for C, S in [(64, 512), (320, 512), (128, 256), (320, 256), (256, 128), (512, 64), (1024, 32)]:
        blk = ConvSnRelu(C, C).to(DEV).train()
        x = torch.randn(N, S, S, C, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
        dy = torch.randn(N, S, S, C, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
        p = ops.SNParams(blk[1])
        y, saved = ops.sn_forward(x, p, True)
        nbytes = 2.0 * N * S * S * C
        ops.PROFILE = []
        for _ in range(4):
            ops.sn_forward(x, p, True)
            ops.sn_backward(dy, saved)
        torch.cuda.synchronize()
        acc = {}
        for tag, work, e0, e1 in ops.PROFILE[len(ops.PROFILE) // 2:]:
            a = acc.setdefault(tag, [0.0, 0.0])
            a[0] += work
            a[1] += e0.elapsed_time(e1) * 1e-3
        ops.PROFILE = None
        print("sn C=%4d @%3d^2 N=%d (%.0f MB): " % (C, S, N, nbytes / 1e6) +
              "  ".join("%s %.0f GB/s" % (t, w / s / 1e9) for t, (w, s) in sorted(acc.items())), flush=True)


def bench_upconv(N):
    for cin, s, k in [(1024, 32, 16), (1024, 32, 2), (320, 256, 2), (320, 128, 4), (320, 64, 8)]:
        blk = ConvSnRelu(cin, 64).to(DEV).train()
        x = torch.randn(N, s, s, cin, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
        z = blk.taps(x).detach().requires_grad_(True)
        dy = torch.randn(N, s * k, s * k, 64, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
        ops.PROFILE = []
        for _ in range(3):
            y = blk.up(z, k)
            torch.autograd.grad(y, z, dy)
        torch.cuda.synchronize()
        acc = {}
        for tag, work, e0, e1 in ops.PROFILE[len(ops.PROFILE) // 3:]:
            a = acc.setdefault(tag, [0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
        ops.PROFILE = None
        print("upconv %4d ch %3d^2 x%-2d N=%d: " % (cin, s, k, N) +
              "  ".join("%s %.3f ms" % (t, s_ / n * 1e3) for t, (n, s_) in sorted(acc.items())), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.what in ("conv", "all"):
        bench_conv(a.batch)
    if a.what in ("sn", "all"):
        bench_sn(a.batch)
    if a.what in ("upconv", "all"):
        bench_upconv(a.batch)
