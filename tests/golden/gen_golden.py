#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE on CPU.

Run only in the build container (the reference tree is mounted read-only at
/root/reference and never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

The reference modules are imported from where they lie; nothing of them is
copied.  Every tensor that enters the reference is produced by the closed-form
`fill()` pattern below (or, for labels, an integer pattern) so the fixtures do
not depend on torch's RNG stream.  Outputs are stored as small .npz files in
this directory and committed.

Reference entry points exercised (file:line in /root/reference):
  modeling/weightstd.py:8-18      Conv2dws
  modeling/snorm.py:5-81          SwitchNorm2d
  modeling/UNet_SNws.py:42-152    UNet_SNws
  modeling/UNet3p_SNws.py:42-160  UNet3p_SNws
  modeling/UNet3p_res_ocr_SNws.py:211-377  UNet3p_res_ocr_SNws (Dropout2d p set to 0: it is RNG-dependent)
  utils/metrics.py:4-52           Evaluator
  utils/loss.py:146-159           SegmentationLosses.CrossEntropyLoss
  utils/lr_scheduler.py:43-61     LR_Scheduler (poly)
  utils/gen_tiles_offs.py:3-27    gen_tiles_offs
"""
import io
import os
import sys
import contextlib

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import modeling.weightstd as ref_ws            # noqa: E402
import modeling.snorm as ref_sn                # noqa: E402
from modeling.UNet_SNws import UNet_SNws       # noqa: E402
from modeling.UNet3p_SNws import UNet3p_SNws   # noqa: E402
from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws   # noqa: E402
from utils.metrics import Evaluator            # noqa: E402
from utils.loss import SegmentationLosses      # noqa: E402
from utils.lr_scheduler import LR_Scheduler    # noqa: E402
from utils.gen_tiles_offs import gen_tiles_offs  # noqa: E402

torch.set_num_threads(8)


sys.path.insert(0, HERE)
from patterns import (fill, label_pattern, t32, det_init, det_init_ocr, synth_batch,  # noqa: E402
                      synth_batch_learnable)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


# --------------------------------------------------------------------------- G1
def gen_conv2dws():
    cases = [
        # (O, I, k, stride, pad, dil, bias)
        (8, 14, 3, 1, 1, 1, True),
        (8, 16, 1, 1, 0, 1, True),
        (4, 6, 3, 1, 2, 2, False),
        (6, 5, 3, 2, 1, 1, True),
    ]
    out = {}
    for ci, (O, I, k, s, p, d, has_b) in enumerate(cases):
        conv = ref_ws.Conv2dws(I, O, k, s, p, d, 1, has_b)
        w = t32(fill((O, I, k, k), 1 + ci, 0.7)) + t32(fill((O, I, k, k), 11 + ci, 0.2, 0.05)) ** 2
        conv.weight.data.copy_(w)
        if has_b:
            conv.bias.data.copy_(t32(fill((O,), 21 + ci, 0.3)))
        x = t32(fill((2, I, 12, 12), 31 + ci, 1.5, 0.25)).requires_grad_(True)
        y = conv(x)
        dy = t32(fill(tuple(y.shape), 41 + ci, 1.0))
        y.backward(dy)
        # standardized weight as the reference computes it (weightstd.py:13-17)
        with torch.no_grad():
            wt = conv.weight
            wm = wt.mean(dim=1, keepdim=True).mean(dim=2, keepdim=True).mean(dim=3, keepdim=True)
            wc = wt - wm
            std = wc.view(O, -1).std(dim=1).view(-1, 1, 1, 1) + 1e-5
            what = wc / std
        pre = "c%d_" % ci
        out[pre + "cfg"] = np.array([O, I, k, s, p, d, int(has_b)], dtype=np.int64)
        out[pre + "w"] = conv.weight.detach().numpy()
        if has_b:
            out[pre + "b"] = conv.bias.detach().numpy()
            out[pre + "db"] = conv.bias.grad.numpy()
        out[pre + "x"] = x.detach().numpy()
        out[pre + "dy"] = dy.numpy()
        out[pre + "what"] = what.numpy()
        out[pre + "y"] = y.detach().numpy()
        out[pre + "dx"] = x.grad.numpy()
        out[pre + "dw"] = conv.weight.grad.numpy()
    out["n_cases"] = np.array(len(cases))
    save("g1_conv2dws", **out)


# --------------------------------------------------------------------------- G2
def gen_switchnorm():
    shapes = [(3, 4, 5, 6), (2, 8, 16, 16), (1, 8, 8, 8), (4, 8, 16, 16)]
    variants = [
        # name, using_movavg, using_bn, last_gamma
        ("bn_mov", True, True, False),
        ("bn_acc", False, True, False),
        ("nobn", True, False, False),
        ("lastg", True, True, True),
    ]
    out = {}
    idx = 0
    for si, shp in enumerate(shapes):
        N, C, H, W = shp
        for (vn, mov, ubn, lg) in variants:
            if si >= 2 and vn != "bn_mov":
                continue
            m = ref_sn.SwitchNorm2d(C, using_movavg=mov, using_bn=ubn, last_gamma=lg)
            nw = 3 if ubn else 2
            if not lg:
                m.weight.data.copy_(t32(fill((1, C, 1, 1), 3 + idx, 0.3, 1.0)))
            m.bias.data.copy_(t32(fill((1, C, 1, 1), 5 + idx, 0.2)))
            m.mean_weight.data.copy_(t32(fill((nw,), 7 + idx, 0.8, 0.3)))
            m.var_weight.data.copy_(t32(fill((nw,), 9 + idx, 0.6, -0.1)))
            m.train()
            pre = "s%d_%s_" % (si, vn)
            x = t32(fill(shp, 13 + idx, 2.0, 0.7) + fill(shp, 17 + idx, 0.5) ** 3).requires_grad_(True)
            y = m(x)
            dy = t32(fill(shp, 19 + idx, 1.0, 0.1))
            y.backward(dy)
            out[pre + "cfg"] = np.array([N, C, H, W, int(mov), int(ubn), int(lg)], dtype=np.int64)
            out[pre + "x"] = x.detach().numpy()
            out[pre + "gamma"] = m.weight.detach().numpy().copy()
            out[pre + "beta"] = m.bias.detach().numpy().copy()
            out[pre + "mw"] = m.mean_weight.detach().numpy().copy()
            out[pre + "vw"] = m.var_weight.detach().numpy().copy()
            out[pre + "dy"] = dy.numpy()
            out[pre + "y"] = y.detach().numpy()
            out[pre + "dx"] = x.grad.numpy().copy()
            out[pre + "dgamma"] = m.weight.grad.numpy().copy()
            out[pre + "dbeta"] = m.bias.grad.numpy().copy()
            out[pre + "dmw"] = m.mean_weight.grad.numpy().copy()
            out[pre + "dvw"] = m.var_weight.grad.numpy().copy()
            if ubn:
                out[pre + "rm1"] = m.running_mean.numpy().copy()
                out[pre + "rv1"] = m.running_var.numpy().copy()
                # three more sequential training calls on scaled inputs (shared-module EMA order)
                with torch.no_grad():
                    for j in range(3):
                        m(x.detach() * (1.0 + 0.5 * (j + 1)) - 0.3 * j)
                out[pre + "rm4"] = m.running_mean.numpy().copy()
                out[pre + "rv4"] = m.running_var.numpy().copy()
            # eval-mode forward with the buffers as they stand now
            m.eval()
            with torch.no_grad():
                out[pre + "y_eval"] = m(x.detach()).numpy()
            idx += 1
    save("g2_switchnorm", **out)


# --------------------------------------------------------------------------- G4
def _run_model(cls, tag, dtype):
    crit = SegmentationLosses(weight=None, cuda=False).build_loss(mode="ce")
    model = cls(14, 4, 10, 1, 1)
    if tag == "unet3p_res_ocr":
        det_init_ocr(model)
        model.ocr_distri_head.conv_bn_dropout[3].p = 0.0
    else:
        det_init(model)
    model = model.to(dtype)
    model.train()
    img, lab = synth_batch(2, 14, 64, 101)
    img = img.to(dtype)
    out = model(img)
    loss = crit(out, lab)
    loss.backward()
    arrays = {"logits": out.detach().numpy(), "loss": np.array(loss.item(), dtype=np.float64)}
    names = []
    for name, p in model.named_parameters():
        names.append(name)
        arrays["grad:" + name] = (p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), out.detach().numpy().dtype))
    for name, b in model.named_buffers():
        arrays["buf:" + name] = b.numpy().copy()
    arrays["param_names"] = np.array(names)
    arrays["state_keys"] = np.array(list(model.state_dict().keys()))
    arrays["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in model.state_dict().values()])
    # eval-mode logits with the running buffers after that one training forward
    model.eval()
    with torch.no_grad():
        arrays["logits_eval"] = model(img).numpy()
    return arrays


def gen_models(only=None):
    for cls, tag in ((UNet_SNws, "unet"), (UNet3p_SNws, "unet3p"), (UNet3p_res_ocr_SNws, "unet3p_res_ocr")):
        if only and tag not in only:
            continue
        if tag != "unet3p_res_ocr":
            save("g4_" + tag, **_run_model(cls, tag, torch.float32))
            continue
        # The res-OCR model at this toy size (batch 2, 4x4 deepest map, BatchNorm over two images) amplifies float32
        # rounding to ~4e-3 of the logits, so the reference is run in float64 for the expected values and once more
        # in float32 to record how far the reference's own float32 arithmetic is from them ("gap:*" = max absolute
        # difference per tensor; some gradients are exactly 0 in exact arithmetic, e.g. a conv bias in front of BN).  Expected tensors are stored as float32 to keep the file small.
        exact = _run_model(cls, tag, torch.float64)
        single = _run_model(cls, tag, torch.float32)
        arrays = {}
        for k, v in exact.items():
            if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.ndim > 0:
                arrays[k] = v.astype(np.float32)
                if k.startswith(("grad:", "logits")):
                    arrays["gap:" + k] = np.array(np.abs(single[k].astype(np.float64) - v).max())
            else:
                arrays[k] = v
        arrays["gap:loss"] = np.array(abs(float(single["loss"]) - float(exact["loss"])))
        save("g4_" + tag, **arrays)


# --------------------------------------------------------------------------- G5
def gen_loss_curve(n_filters=4, name="g5_loss_curve"):
    steps = 200
    model = UNet3p_SNws(14, n_filters, 10, 1, 1)
    det_init(model)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=5e-4)
    with contextlib.redirect_stdout(io.StringIO()):
        sched = LR_Scheduler("poly", 2e-3, 1, steps)
    crit = SegmentationLosses(weight=None, cuda=False).build_loss(mode="ce")
    batches = [synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
    losses, lrs = [], []
    for i in range(steps):
        img, lab = batches[i % 4]
        with contextlib.redirect_stdout(io.StringIO()):
            sched(opt, i, 0, 0.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.zero_grad()
        loss = crit(model(img), lab)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if i % 20 == 0:
            print("step", i, "loss", losses[-1], flush=True)
    rm = torch.cat([b.flatten() for n, b in model.named_buffers() if n.endswith("running_mean")])
    rv = torch.cat([b.flatten() for n, b in model.named_buffers() if n.endswith("running_var")])
    save(name, losses=np.array(losses, np.float64), lrs=np.array(lrs, np.float64),
         running_mean_sum=np.array(rm.double().sum().item()), running_var_sum=np.array(rv.double().sum().item()))


# --------------------------------------------------------------------------- G6
def metrics_of(ev):
    with np.errstate(all="ignore"):
        return (np.array(ev.Pixel_Accuracy()), np.asarray(ev.Pixel_Accuracy_Class()),
                np.array(ev.Mean_Intersection_over_Union()),
                np.array(ev.Frequency_Weighted_Intersection_over_Union()))


def gen_evaluator():
    out = {}
    # case A: small, two accumulated batches
    rng = np.random.default_rng(0)
    gt = rng.integers(0, 10, size=(2, 64, 64)).astype(np.float32)
    pr = rng.integers(0, 10, size=(2, 64, 64)).astype(np.int64)
    ev = Evaluator(10)
    ev.add_batch(gt, pr)
    ev.add_batch(gt[::-1].copy(), pr)
    out["a_gt"], out["a_pred"], out["a_cm"] = gt, pr, ev.confusion_matrix.copy()
    out["a_pa"], out["a_pac"], out["a_miou"], out["a_fwiou"] = metrics_of(ev)
    # case B: edge labels (255, -1, fractional, NaN) and absent classes -> NaN paths
    gt = rng.integers(0, 7, size=(1, 48, 40)).astype(np.float32)
    gt[0, ::5, ::3] = 255.0
    gt[0, 1::7, 1::4] = -1.0
    gt[0, 2::9, ::2] = 9.0
    gt[0, 3::11, 1::5] += 0.75
    gt[0, 4, 4] = np.nan
    gt[0, 5, 5] = 10.0
    pr = rng.integers(0, 8, size=(1, 48, 40)).astype(np.int64)
    ev = Evaluator(10)
    with np.errstate(all="ignore"):
        ev.add_batch(gt, pr)
    out["b_gt"], out["b_pred"], out["b_cm"] = gt, pr, ev.confusion_matrix.copy()
    out["b_pa"], out["b_pac"], out["b_miou"], out["b_fwiou"] = metrics_of(ev)
    # case C: everything ignored
    gt = np.full((1, 8, 8), 255.0, np.float32)
    pr = np.zeros((1, 8, 8), np.int64)
    ev = Evaluator(10)
    ev.add_batch(gt, pr)
    out["c_cm"] = ev.confusion_matrix.copy()
    out["c_pa"], out["c_pac"], out["c_miou"], out["c_fwiou"] = metrics_of(ev)
    # case D: one full FY-4A disk, stored as seed + expected results only
    rng = np.random.default_rng(0)
    gt = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.float32)
    pr = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.int64)
    ev = Evaluator(10)
    ev.add_batch(gt, pr)
    out["d_seed"] = np.array(0)
    out["d_cm"] = ev.confusion_matrix.copy()
    out["d_pa"], out["d_pac"], out["d_miou"], out["d_fwiou"] = metrics_of(ev)
    # case E: 3 classes, different C
    gt = rng.integers(0, 3, size=(3, 17, 19)).astype(np.float32)
    pr = rng.integers(0, 3, size=(3, 17, 19)).astype(np.int64)
    ev = Evaluator(3)
    ev.add_batch(gt, pr)
    out["e_gt"], out["e_pred"], out["e_cm"] = gt, pr, ev.confusion_matrix.copy()
    out["e_pa"], out["e_pac"], out["e_miou"], out["e_fwiou"] = metrics_of(ev)
    save("g6_evaluator", **out)


# --------------------------------------------------------------------------- G7 / G8
def gen_misc():
    out = {}
    for i, args in enumerate([(2748, 2748, 512, 0), (2748, 2748, 513, 0), (2748, 2748, 256, 0),
                              (2748, 2748, 512, 64), (770, 1024, 256, 150), (512, 512, 512, 0),
                              (1024, 600, 512, 0)]):
        out["tiles%d_args" % i] = np.array(args, np.int64)
        out["tiles%d" % i] = np.array(gen_tiles_offs(*args), np.int64)
    out["n_tiles_cases"] = np.array(7)
    # poly LR
    with contextlib.redirect_stdout(io.StringIO()):
        sched = LR_Scheduler("poly", 0.002, 200, 746)

    class _Opt:
        param_groups = [{"lr": 0.0}]
    lrs = []
    pts = [(0, 0), (1, 0), (745, 0), (0, 1), (373, 100), (745, 199)]
    for (i, ep) in pts:
        with contextlib.redirect_stdout(io.StringIO()):
            sched(_Opt, i, ep, 0.0)
        lrs.append(_Opt.param_groups[0]["lr"])
    out["lr_pts"] = np.array(pts, np.int64)
    out["lr_vals"] = np.array(lrs, np.float64)
    # CE / N with ignore_index 255
    crit = SegmentationLosses(weight=None, cuda=False).build_loss(mode="ce")
    logit = t32(fill((3, 10, 9, 7), 71, 2.0)).requires_grad_(True)
    tgt = t32(label_pattern((3, 9, 7), 10, ignore_every=5))
    loss = crit(logit, tgt)
    loss.backward()
    out["ce_logit"], out["ce_target"] = logit.detach().numpy(), tgt.numpy()
    out["ce_loss"], out["ce_dlogit"] = np.array(loss.item(), np.float64), logit.grad.numpy()
    save("g78_misc", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g4", "g5", "g6", "g78"]
    if "g1" in which:
        gen_conv2dws()
    if "g2" in which:
        gen_switchnorm()
    if "g4" in which:
        gen_models()
    if "g4ocr" in which:    # only the res-OCR model (the other G4 fixtures stay as committed)
        gen_models(only=("unet3p_res_ocr",))
    if "g5" in which:
        gen_loss_curve()
    if "g5b" in which:      # MFMA-aligned widths (16..256 channels) for the bf16 tier's 200-step curve
        gen_loss_curve(16, "g5b_loss_curve_nf16")
    if "g6" in which:
        gen_evaluator()
    if "g78" in which:
        gen_misc()
