#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rates of the two entry points that accept host data: Evaluator.add_batch with numpy
arrays, and a training step whose batch starts in (pinned / pageable) host memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import numpy as np
import torch
from utils.metrics import Evaluator

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
gt = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.float32)
pr = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.int64)
ev = Evaluator(10)
for _ in range(2):
    ev.add_batch(gt, pr)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 10
for _ in range(R):
    ev.add_batch(gt, pr)
m = ev.confusion_matrix
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / R
print("Evaluator.add_batch, numpy in (pageable host): %.2f ms per 2748^2 disk = %.1f Mpix/s, %.2f GB/s of 12 B/pixel"
      % (t * 1e3, gt.size / t / 1e6, 12.0 * gt.size / t / 1e9), flush=True)
g_d, p_d = torch.from_numpy(gt).to(dev), torch.from_numpy(pr).to(dev)
for _ in range(2):
    ev.add_batch(g_d, p_d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    ev.add_batch(g_d, p_d)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / R
print("Evaluator.add_batch, device tensors in: %.3f ms per disk" % (t * 1e3), flush=True)

img = torch.rand(16, 14, 512, 512) * 255.0
lab = torch.randint(0, 10, (16, 512, 512)).float()
for name, (a, b) in (("pageable", (img, lab)), ("pinned", (img.pin_memory(), lab.pin_memory()))):
    for _ in range(2):
        a.to(dev, non_blocking=True), b.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        x, y = a.to(dev, non_blocking=True), b.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / R
    nbytes = a.numel() * 4 + b.numel() * 4
    print("training batch 16x14x512x512 f32 + labels, %s host -> HBM: %.2f ms (%.1f GB/s)" % (name, t * 1e3, nbytes / t / 1e9), flush=True)
