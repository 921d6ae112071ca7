#!/usr/bin/env python3
"""Per-SHAPE breakdown of the convolution launches of one training step of the benchmark configuration
(UNet3p_SNws 14->10, n_filters 64, 512^2, batch 16, bf16): every snws_conv_mfma_* launch is bracketed by HIP events
and keyed by (kernel family, C_in -> C_out, kernel size, H).  Prints launches per step, ms per step and TFLOP/s,
sorted by time, so that tuning goes where the time is.
    python tools/layer_profile.py [--steps 3] [--batch 16] [--size 512]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, _p)
import torch  # noqa: E402
from snws import harness, ops  # noqa: E402
from modeling.UNet3p_SNws import UNet3p_SNws  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = UNet3p_SNws(14, 64, 10, 1, 1).to(dev)
model.compute_dtype = torch.bfloat16
tr = harness.Trainer(model, total_iters=1000)
img, lab = harness.synthetic_batch(args.batch, 14, args.size, 10, dev, seed=1, dtype=torch.bfloat16)
for _ in range(2):
    tr.step(img, lab)
torch.cuda.synchronize()

records = []
orig = ops._launch


def launch(tag, work, name, *a):
    key = tag
    if name.startswith("snws_conv_mfma"):
        d = a[0]._obj
        key = "%-26s %4d->%4d k%d @%3d" % (tag, d.C_in, d.C_out, d.kh, d.H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = ops.L.call(name, *a)
    e1.record()
    records.append((key, work, e0, e1))
    return rc


ops._launch = launch
for _ in range(args.steps):
    tr.step(img, lab)
torch.cuda.synchronize()
ops._launch = orig
agg = {}
for key, work, e0, e1 in records:
    v = agg.setdefault(key, [0, 0.0, 0.0])
    v[0] += 1
    v[1] += work
    v[2] += e0.elapsed_time(e1) * 1e-3
tot = 0.0
for key, (n, work, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    ms = t / args.steps * 1e3
    tot += ms
    rate = work / t
    unit = "TFLOP/s" if key.startswith("conv") else "GB/s   "
    print("%-50s %3d/step %8.3f ms/step %8.1f %s" % (key, n // args.steps, ms, rate / (1e12 if key.startswith("conv") else 1e9), unit))
print("sum of bracketed kernels: %.2f ms/step" % tot)
