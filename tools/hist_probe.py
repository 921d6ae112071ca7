#!/usr/bin/env python3
"""Confusion-matrix kernel on one 2748 x 2748 disk (f32 labels + i64 predictions, 10 classes) for three label
statistics: uniformly random pixels (the benchmark's), large uniform regions (what real cloud masks look like: the
lanes of a wave then hit few histogram bins), and one constant class (the worst case for same-address atomics).
Six disk pairs are rotated so the operands come from HBM, not from the Infinity Cache."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from utils.metrics import Evaluator

S, C, DEV = 2748, 10, "cuda:0"
g = torch.Generator(device="cpu").manual_seed(0)


def regions():
    coarse = torch.randint(0, C, (1, 1, S // 64 + 1, S // 64 + 1), generator=g).float()
    return torch.nn.functional.interpolate(coarse, scale_factor=64, mode="nearest")[0, 0, :S, :S].contiguous()


cases = {
    "random pixels": lambda: (torch.randint(0, C, (S, S), generator=g).float(), torch.randint(0, C, (S, S), generator=g)),
    "64x64 uniform regions": lambda: (regions(), regions().long()),
    "one constant class": lambda: (torch.full((S, S), 3.0), torch.full((S, S), 3, dtype=torch.int64)),
}
for name, make in cases.items():
    pairs = [tuple(t.to(DEV) for t in make()) for _ in range(6)]
    ev = Evaluator(C)
    for gt, pr in pairs:
        ev.add_batch(gt, pr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        for gt, pr in pairs:
            ev.add_batch(gt, pr)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print("%-24s %7.1f us per disk  %5.2f TB/s of the 12 B/pixel" % (name, us, S * S * 12 / us / 1e6), flush=True)
