#!/usr/bin/env python3
"""How long the host needs to ENQUEUE one training step (no synchronisation) against the step's wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, _p)
import torch
from snws import harness
from modeling.UNet3p_SNws import UNet3p_SNws
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = UNet3p_SNws(14, 64, 10, 1, 1).to(dev)
model.compute_dtype = torch.bfloat16
tr = harness.Trainer(model, total_iters=1000)
img, lab = harness.synthetic_batch(16, 14, 512, 10, dev, dtype=torch.bfloat16)
for _ in range(3):
    tr.step(img, lab)
torch.cuda.synchronize()
enq, wall = [], []
for _ in range(6):
    t0 = time.perf_counter()
    tr.step(img, lab)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print("enqueue ms per step:", [round(v, 1) for v in enq])
print("wall ms per step   :", [round(v, 1) for v in wall])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    tr.step(img, lab)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
