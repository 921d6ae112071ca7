#!/usr/bin/env python3
"""Practical HBM ceiling of this device: device-to-device copy and read-only reduction rates at several sizes."""
import torch
dev = "cuda:0"
for mb in (256, 1024, 2048):
    n = mb * (1 << 20) // 2
    a = torch.randn(n, device=dev).to(torch.bfloat16)
    b = torch.empty_like(a)
    for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 2 * a.numel() * 2), ("relu", lambda: torch.relu_(b), 2 * a.numel() * 2),
                             ("add3", lambda: torch.add(a, b, out=b), 3 * a.numel() * 2), ("sum", lambda: a.sum(), a.numel() * 2)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%5d MB %-5s %.3f ms  %.2f TB/s" % (mb, name, ms, bytes_ / ms / 1e9))
