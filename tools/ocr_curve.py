#!/usr/bin/env python3
"""Twelve full-size training steps of UNet3p_res_ocr_SNws on one learnable synthetic batch (the configuration of
tests/test_gpu_fullsize.py::test_training_step_full_size_loss_decreases) - prints the loss curve; run it under the
A/B switches SNWS_GRAD_SHARE / SNWS_PAIR_CONV / SNWS_DROPOUT_FOLD = 0 to see what a change does to the trajectory."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402
from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws  # noqa: E402
from snws import harness  # noqa: E402

DEV = "cuda:0"
torch.manual_seed(1)
model = UNet3p_res_ocr_SNws(14, 64, 10, 1, 1).to(DEV)
model.compute_dtype = torch.bfloat16
tr = harness.Trainer(model, total_iters=100)
img, _ = harness.synthetic_batch(16, 14, 512, 10, torch.device(DEV))
lab = img[:, :10].argmax(1).float()
losses = [float(tr.step(img, lab)) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12)]
print(" ".join("%s=%s" % (k, os.environ.get(k, "1")) for k in ("SNWS_GRAD_SHARE", "SNWS_PAIR_CONV", "SNWS_DROPOUT_FOLD")),
      "losses:", " ".join("%.5f" % l for l in losses), flush=True)
