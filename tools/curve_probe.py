#!/usr/bin/env python3
"""Prints per-step relative deviation of the bf16 200-step training curve from the reference's f32 curve."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)
import numpy as np, torch
import patterns
from test_gpu_kernels import _train_curve, G
from modeling.UNet3p_SNws import UNet3p_SNws
g = np.load(os.path.join(G, "g5b_loss_curve_nf16.npz"))
model = UNet3p_SNws(14, 16, 10, 1, 1)
patterns.det_init(model)
batches = [patterns.synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
got = _train_curve(model, batches, 200, torch.bfloat16)
ref = g["losses"]
rel = (got - ref) / ref
np.set_printoptions(precision=3, suppress=True, linewidth=200)
print("M16=%s" % os.environ.get("SNWS_CONV_M16"), "rel[:24] =", rel[:24])
print("   max |rel| by 20-step block:", np.abs(rel).reshape(10, 20).max(axis=1))
