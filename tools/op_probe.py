#!/usr/bin/env python3
"""Lists the stock torch ops (name, input shapes, count) that one training step of the bench model still launches.
    python tools/op_probe.py [filter substring] [unet3p|unet3p_res_ocr]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from modeling.UNet3p_SNws import UNet3p_SNws
from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws
from snws import harness

flt = sys.argv[1] if len(sys.argv) > 1 else "add"
dev = torch.device("cuda", 0)
which = sys.argv[2] if len(sys.argv) > 2 else "unet3p"
model = (UNet3p_res_ocr_SNws if which == "unet3p_res_ocr" else UNet3p_SNws)(14, 64, 10, 1, 1).to(dev).to(memory_format=torch.channels_last)
model.compute_dtype = torch.bfloat16
tr = harness.Trainer(model, total_iters=100)
img, lab = harness.synthetic_batch(4, 14, 256, 10, dev)
for _ in range(2):
    tr.step(img, lab)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    tr.step(img, lab)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if any(f in e.name for f in flt.split(",")):
        cnt[(e.name, str(e.input_shapes))] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
