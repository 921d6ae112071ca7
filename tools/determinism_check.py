#!/usr/bin/env python3
"""Bitwise repeatability of the conv forward/dgrad kernels and of a whole bf16 model forward (no atomics on those
paths, so any difference between repeats is a race).  Usage: python tools/determinism_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, _p)
import torch  # noqa: E402
from snws import ops  # noqa: E402
import modeling.weightstd as ws  # noqa: E402
from modeling.UNet3p_SNws import UNet3p_SNws  # noqa: E402

DEV = "cuda:0"
torch.manual_seed(0)
bad = 0
for (ci, co, k, hw, n) in [(64, 64, 3, 64, 4), (320, 320, 3, 64, 4), (16, 16, 3, 64, 4), (80, 80, 3, 32, 4), (80, 144, 1, 32, 4),
                           (256, 576, 1, 16, 4), (32, 48, 3, 37, 3)]:
    conv = ws.Conv2dws(ci, co, k, 1, k // 2).to(DEV)
    x = torch.randn(n, hw, hw, ci, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = torch.randn(n, hw, hw, co, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
    prep = conv.prepared(True)
    y0, st0, saved = ops.conv_forward(x, prep, conv.bias, want_stats=True)
    dx0 = ops.conv_backward(dy, prep, saved, conv.weight, True, False, False)[0]
    for rep in range(30):
        y1, st1, saved = ops.conv_forward(x, prep, conv.bias, want_stats=True)
        dx1 = ops.conv_backward(dy, prep, saved, conv.weight, True, False, False)[0]
        if not torch.equal(y0, y1) or not torch.equal(dx0, dx1) or not torch.equal(st0, st1):
            bad += 1
            print("MISMATCH conv", ci, co, k, hw, "rep", rep, float((y0.float() - y1.float()).abs().max()),
                  float((dx0.float() - dx1.float()).abs().max()))
            break
    print("conv %d->%d k%d @%d ok" % (ci, co, k, hw))
import patterns  # noqa: E402
model = UNet3p_SNws(14, 16, 10, 1, 1)
patterns.det_init(model)
model = model.to(DEV).train()
model.compute_dtype = torch.bfloat16
img, lab = patterns.synth_batch_learnable(4, 14, 64, 200)
img = img.to(DEV)
with torch.no_grad():
    ref = model(img).float().clone()
    for rep in range(20):
        out = model(img).float()
        if not torch.equal(ref, out):
            bad += 1
            print("MISMATCH model forward rep", rep, float((ref - out).abs().max()))
            break
# fused blocks: forward (up-sampled conv from LDS + statistics), and the backward chain whose dgrad epilogue feeds the
# SwitchNorm reduction; everything but the weight gradients (float atomics) must repeat bit for bit
from modeling.blocks import ConvSnRelu  # noqa: E402
torch.manual_seed(3)
b1, b2 = ConvSnRelu(64, 64).to(DEV).train(), ConvSnRelu(64, 64).to(DEV).train()
for k in (2, 4, 8, 16):
    x = torch.randn(2, 64, 64 // k * 2, 64 // k * 2, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    ref = None
    for rep in range(10):
        x.grad = None
        ops.FUSED_SN_BWD_HITS = 0
        y = b2(b1.on_upsampled(x, k))
        g = torch.ones_like(y) * 0.01
        y.backward(g)
        cur = (y.detach().float().clone(), x.grad.float().clone())
        if ref is None:
            ref, hits = cur, ops.FUSED_SN_BWD_HITS
        elif not (torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])):
            bad += 1
            print("MISMATCH fused chain k", k, "rep", rep)
            break
    print("upconv k%d -> conv chain repeatable (fused SwitchNorm reductions: %d)" % (k, hits))
print("all repeatable" if not bad else "FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
