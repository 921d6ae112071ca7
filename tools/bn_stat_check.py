#!/usr/bin/env python3
"""Full-size check of the wide 1x1 kernel's statistics epilogue (16 x 512 x 512 pixels, 512 -> 512 channels, plain and
accumulating) against float64 sums of the stored tensor, and of the res-OCR model's gradients with the BatchNorm
statistics taken from the convolutions' epilogues against the run that re-reads the tensors."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from snws import _lib as L
from snws import harness, ops

DEV = "cuda:0"
torch.manual_seed(0)
imgs, HW, N, K = 16, 512 * 512, 512, 512
M = imgs * HW
x = (torch.randn(M, K, device=DEV) + 0.3).to(torch.bfloat16)
w = (torch.randn(N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device=DEV)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())      # noqa: E731
T = L.lib().snws_gemm1x1_stat_tiles(HW)
for acc in (0, 1):
    y = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    stat = torch.full((imgs, T, N, 2), float("nan"), device=DEV)
    L.call("snws_gemm1x1_stats", P(x), K, P(w), None if acc else P(b), P(y), N, M, N, K, acc, HW, P(stat), st)
    torch.cuda.synchronize()
    worst = 0.0
    for i in range(imgs):
        yi = y[i * HW:(i + 1) * HW].double().view(T, 256, N)
        ref = torch.stack([yi.sum(1), (yi * yi).sum(1)], -1)
        worst = max(worst, float((stat[i].double() - ref).abs().max() / ref.abs().max()))
    print("accumulate %d: per-tile moments vs float64 of the stored tensor: max rel err %.3g, finite %s"
          % (acc, worst, bool(torch.isfinite(stat).all())), flush=True)
del x, w, y, stat

from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws
torch.manual_seed(1)
model = UNet3p_res_ocr_SNws(14, 64, 10, 1, 1).to(DEV).train()
model.compute_dtype = torch.bfloat16
img, _ = harness.synthetic_batch(16, 14, 512, 10, torch.device(DEV))
lab = img[:, :10].argmax(1)
res = {}
for tag, fuse, pair in (("fused", True, True), ("reread", False, True), ("reread2", False, True), ("fused_composed", True, False)):
    ops.BN_STAT_FUSION, ops.PAIR_CONV = fuse, pair
    model.zero_grad(set_to_none=True)
    for m in model.modules():
        if hasattr(m, "running_mean") and m.running_mean is not None and isinstance(m, torch.nn.BatchNorm2d):
            pass
    torch.manual_seed(5)
    h0 = ops.BN_STAT_HITS
    loss = torch.nn.functional.cross_entropy(model(img).float(), lab)
    loss.backward()
    res[tag] = (loss.item(), [p.grad.float().clone() for p in model.parameters() if p.grad is not None], ops.BN_STAT_HITS - h0)
    del loss


def l2(a, b):
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


for a, bb in (("fused", "reread"), ("reread2", "reread"), ("fused_composed", "fused")):
    errs = sorted(l2(p, q) for p, q in zip(res[a][1], res[bb][1]))
    print("%s vs %s: loss %.5f / %.5f, hits %d / %d, gradient errors median %.4f p90 %.4f max %.4f"
          % (a, bb, res[a][0], res[bb][0], res[a][2], res[bb][2], errs[len(errs) // 2], errs[int(len(errs) * 0.9)], errs[-1]), flush=True)
