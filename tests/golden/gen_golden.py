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
G9 runs modeling/UNet3p_SNws.py:42-160 a second time with bf16 storage roundings applied from outside (hooks).
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


def gen_unet_odd():
    """UNet_SNws on an input whose level sizes go odd (75 -> 38 -> 19 -> 10 -> 5): CropConcat (UNet_SNws.py:35-39)
    zero-pads the skips upwards (19 vs 20, 38 vs 40, 75 vs 80), the output is 80 x 80.  No label of that size exists,
    so the differentiated scalar is sum(logits * dy) with a pattern dy."""
    def run(dtype):
        model = UNet_SNws(14, 4, 10, 1, 1)
        det_init(model)
        model = model.to(dtype).train()
        img, _ = synth_batch(2, 14, 75, 131)
        out = model(img.to(dtype))
        dy = t32(fill(tuple(out.shape), 133, 1.0, 0.1)).to(dtype)
        (out * dy).sum().backward()
        res = {"logits": out.detach().numpy()}
        for name, p in model.named_parameters():
            res["grad:" + name] = p.grad.numpy()
        for name, b in model.named_buffers():
            res["buf:" + name] = b.numpy().copy()
        return res, [n for n, _ in model.named_parameters()]
    # gradients of the first layers are sums over 80 x 80 x 2 positions of strongly cancelling terms: the reference's
    # own float32 run is a few 1e-3 away from its float64 run there.  Expected values come from the float64 run
    # (stored as float32); "gap:*" records the float32 run's distance per tensor (max abs / max abs).
    exact, names = run(torch.float64)
    single, _ = run(torch.float32)
    arrays = {"in_size": np.array(75), "param_names": np.array(names)}
    for k, v in exact.items():
        arrays[k] = v.astype(np.float32)
        arrays["gap:" + k] = np.array(np.abs(single[k].astype(np.float64) - v).max() / max(np.abs(v).max(), 1e-30))
    print("worst float32-vs-float64 gap of the reference:", max(float(v) for k, v in arrays.items() if k.startswith("gap:")))
    save("g4_unet_odd", **arrays)


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


def gen_weighted_ce():
    """g8w: the reference's criterion built with class weights (train.py:55-67 -> SegmentationLosses(weight=...),
    utils/loss.py:146-159): weighted mean over the non-ignored pixels, / batch size; loss and d(loss)/d(logits)."""
    w = t32(fill((10,), 73, 0.8, 1.3))
    crit = SegmentationLosses(weight=w, cuda=False).build_loss(mode="ce")
    logit = t32(fill((3, 10, 9, 7), 71, 2.0)).requires_grad_(True)
    tgt = t32(label_pattern((3, 9, 7), 10, ignore_every=5))
    loss = crit(logit, tgt)
    loss.backward()
    save("g8w_weighted_ce", weight=w.numpy(), logit=logit.detach().numpy(), target=tgt.numpy(),
         loss=np.array(loss.item(), np.float64), dlogit=logit.grad.numpy())


# --------------------------------------------------------------------------- G9
def _ste_round(t, dtype=torch.bfloat16):
    """Value rounded to `dtype`, gradient passed straight through."""
    return t + (t.detach().to(dtype).to(t.dtype) - t.detach())


class _Bf16Storage:
    """Runs the REFERENCE model as if every tensor the MI355X path stores in bf16 were stored in bf16: the
    standardized filter, every Conv2dws output, every ConvSnRelu block output, every up-sampled map, the head's
    filter, bias and output (straight-through rounding, arithmetic stays float32).  Implemented from outside the
    reference: forward hooks on its modules and a wrapper around torch.nn.functional.conv2d while the model runs."""

    def __init__(self, model, n_class):
        self.model, self.n_class, self.handles, self.orig = model, n_class, [], None

    def __enter__(self):
        import torch.nn as nn
        import torch.nn.functional as F
        for m in self.model.modules():
            if isinstance(m, (ref_ws.Conv2dws, nn.Sequential, nn.Upsample)) or \
                    (isinstance(m, nn.Conv2d) and m.out_channels == self.n_class):
                self.handles.append(m.register_forward_hook(lambda mod, inp, out: _ste_round(out)))
        self.orig = F.conv2d
        n_class = self.n_class

        def conv2d(input, weight, bias=None, *args, **kw):
            head = weight.shape[0] == n_class and weight.shape[2] == 1
            return self.orig(input, _ste_round(weight), _ste_round(bias) if (head and bias is not None) else bias,
                             *args, **kw)
        F.conv2d = conv2d
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.conv2d = self.orig
        for h in self.handles:
            h.remove()


def _pack_scaled_f16(t):
    a = t.detach().numpy().astype(np.float64)
    scale = float(np.abs(a).max()) or 1.0
    return (a / scale).astype(np.float16), np.array(scale, np.float64)


def gen_bf16_storage(warm_steps=20, n_filters=16):
    """Reference-generated fixture for the bf16 tier: UNet3p_SNws(14, 16, 10) after `warm_steps` float32 AdamW steps
    of the reference on learnable synthetic batches (a trained, well-conditioned state instead of a pattern-filled
    one), parameters then rounded to bf16 values; on one more batch: the plain float32 reference step, the reference
    step with bf16 storage roundings (see _Bf16Storage), and that step once more with every parameter moved by a
    relative 1e-6 - how far the reference's OWN bf16-storage result moves under a change far below float32
    resolution of the weights' effect ("self noise": two correct implementations cannot agree better)."""
    crit = SegmentationLosses(weight=None, cuda=False).build_loss(mode="ce")
    model = UNet3p_SNws(14, n_filters, 10, 1, 1)
    det_init(model)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=5e-4)
    batches = [synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
    for i in range(warm_steps):
        img, lab = batches[i % 4]
        opt.zero_grad()
        loss = crit(model(img), lab)
        loss.backward()
        opt.step()
    print("warm-up done, loss", loss.item(), flush=True)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img, lab = synth_batch_learnable(4, 14, 64, 211)
    img = img.to(torch.bfloat16).float()

    def run(storage, perturb=0.0):
        model.load_state_dict(state0)
        if perturb:
            with torch.no_grad():
                for k, p in enumerate(model.parameters()):
                    p.mul_(1.0 + perturb * t32(fill(tuple(p.shape), 300 + k, 1.0)))
        model.zero_grad()
        model.train()
        if storage:
            with _Bf16Storage(model, 10):
                out = model(img)
                loss = crit(out, lab)
                loss.backward()
        else:
            out = model(img)
            loss = crit(out, lab)
            loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        bufs = {n: b.detach().clone() for n, b in model.named_buffers()}
        return out.detach().clone(), float(loss.item()), grads, bufs

    def l2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())

    def cat(g):
        return torch.cat([v.flatten() for v in g.values()])

    lo32, loss32, g32, _ = run(False)
    lob, lossb, gb, bufb = run(True)
    lop, lossp, gp, _ = run(True, perturb=1e-6)
    arrays = {"warm_steps": np.array(warm_steps), "n_filters": np.array(n_filters),
              "logits_f32": lo32.numpy(), "loss_f32": np.array(loss32), "logits_bf16": lob.numpy(),
              "loss_bf16": np.array(lossb),
              "noise_logits": np.array(l2(lop, lob)), "noise_grads_all": np.array(l2(cat(gp), cat(gb))),
              "storage_logits": np.array(l2(lob, lo32)), "storage_grads_all": np.array(l2(cat(gb), cat(g32)))}
    names = []
    for n, p in model.named_parameters():
        names.append(n)
        arrays["param:" + n] = state0[n].to(torch.bfloat16).view(torch.int16).numpy()     # exact: values are bf16
        arrays["grad16:" + n], arrays["gscale:" + n] = _pack_scaled_f16(gb[n])
        arrays["noise:" + n] = np.array(l2(gp[n], gb[n]))
        arrays["storage:" + n] = np.array(l2(gb[n], g32[n]))
    for n, b in model.named_buffers():
        arrays["buf0:" + n] = state0[n].numpy()
        arrays["buf1:" + n] = bufb[n].numpy()
    arrays["param_names"] = np.array(names)
    print("self noise: logits %.4f grads(all) %.4f | storage vs f32: logits %.4f grads(all) %.4f" %
          (arrays["noise_logits"], arrays["noise_grads_all"], arrays["storage_logits"], arrays["storage_grads_all"]))
    save("g9_bf16_storage", **arrays)


# --------------------------------------------------------------------------- G9b / G5c
class _Bf16StorageAll:
    """_Bf16Storage for models that hold more than ConvSnRelu blocks (UNet3p_res_ocr_SNws.py:211-377): every tensor
    the MI355X path stores in bf16 is rounded straight-through in the REFERENCE, from outside: every filter that
    enters F.conv2d (the standardized one for Conv2dws), the output of every nn.Conv2d (Conv2dws included), of every
    nn.ReLU (block outputs, the residual blocks' two uses of their ReLU, the BatchNorm + ReLU pairs of the OCR head),
    of every nn.Upsample, both attention soft-max results (F.softmax on 3-D inputs), the object-region features (output of
    SpatialGather_Module) and the attention context (input of f_up)."""

    def __init__(self, model):
        self.model, self.handles, self.orig = model, [], None

    def __enter__(self):
        import torch.nn as nn
        import torch.nn.functional as F
        rnd_out = lambda mod, inp, out: _ste_round(out)                      # noqa: E731
        for name, m in self.model.named_modules():
            if isinstance(m, (nn.Conv2d, nn.ReLU, nn.Upsample)) or type(m).__name__ == "SpatialGather_Module":
                self.handles.append(m.register_forward_hook(rnd_out))
            if name.endswith("object_context_block.f_up"):
                self.handles.append(m.register_forward_pre_hook(lambda mod, inp: (_ste_round(inp[0]),)))
        self.orig = (F.conv2d, F.softmax)
        o_conv, o_soft = self.orig
        F.conv2d = lambda input, weight, *a, **kw: o_conv(input, _ste_round(weight), *a, **kw)
        # (nn.Softmax(0) over SwitchNorm's three mixing weights, snorm.py:68-70, also lands in F.softmax: those stay
        # float32 on the MI355X path - only the two attention soft-maxes over (B, K, HW) / (B, HW, K) are stored)
        F.softmax = lambda x, *a, **kw: _ste_round(o_soft(x, *a, **kw)) if x.dim() >= 3 else o_soft(x, *a, **kw)
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.conv2d, F.softmax = self.orig
        for h in self.handles:
            h.remove()


def gen_bf16_storage_ocr(warm_steps=20, n_filters=16, write=True):
    """g9b: gen_bf16_storage's protocol for the reference UNet3p_res_ocr_SNws(14, 16, 10) (Dropout2d p = 0): a trained
    state (20 float32 AdamW steps of the reference), parameters rounded to bf16 values, then on one more batch
      * the plain float32 step,
      * the step with bf16 storage roundings (_Bf16StorageAll) and FLOAT64 arithmetic between the roundings - the
        expected values: what the bf16-storage network computes when nothing but the storage format rounds,
      * the same with the reference's float32 arithmetic ("f32gap_*": its distance from the float64-arithmetic run;
        on this data SwitchNorm's var_ln = mean(var_in + mean_in^2) - mean_ln^2 (snorm.py:48-50,54-55) cancels in
        float32, which moves first-layer outputs by 1e-3 and flips a tenth of the bf16 roundings that follow),
      * the float64-arithmetic step with every parameter moved by a relative 1e-6 ("noise_*": the bf16-storage
        network's sensitivity to rounding flips alone)."""
    crit = SegmentationLosses(weight=None, cuda=False).build_loss(mode="ce")
    model = UNet3p_res_ocr_SNws(14, n_filters, 10, 1, 1)
    det_init_ocr(model)
    model.ocr_distri_head.conv_bn_dropout[3].p = 0.0
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=5e-4)
    batches = [synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
    for i in range(warm_steps):
        img, lab = batches[i % 4]
        opt.zero_grad()
        loss = crit(model(img), lab)
        loss.backward()
        opt.step()
        print("warm-up step", i, "loss", loss.item(), flush=True)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img, lab = synth_batch_learnable(4, 14, 64, 211)
    img = img.to(torch.bfloat16).float()

    def run(storage, dtype, perturb=0.0):
        model.to(dtype)
        model.load_state_dict(state0)
        if perturb:
            with torch.no_grad():
                for k, p in enumerate(model.parameters()):
                    p.mul_(1.0 + perturb * t32(fill(tuple(p.shape), 300 + k, 1.0)).to(dtype))
        model.zero_grad()
        model.train()
        with (_Bf16StorageAll(model) if storage else contextlib.nullcontext()):
            out = model(img.to(dtype))
            loss = crit(out, lab)
            loss.backward()
        grads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
                 for n, p in model.named_parameters()}
        bufs = {n: b.detach().clone() for n, b in model.named_buffers()}
        return out.detach().clone(), float(loss.item()), grads, bufs

    def l2(a, b):
        return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))

    def cat(g):
        return torch.cat([v.flatten().double() for v in g.values()])

    lo32, loss32, g32, _ = run(False, torch.float32)
    lob, lossb, gb, bufb = run(True, torch.float64)
    lof, lossf, gf, _ = run(True, torch.float32)
    lop, lossp, gp, _ = run(True, torch.float64, perturb=1e-6)
    arrays = {"warm_steps": np.array(warm_steps), "n_filters": np.array(n_filters),
              "logits_f32": lo32.numpy(), "loss_f32": np.array(loss32), "logits_bf16": lob.float().numpy(),
              "loss_bf16": np.array(lossb),
              "noise_logits": np.array(l2(lop, lob)), "noise_grads_all": np.array(l2(cat(gp), cat(gb))),
              "f32gap_logits": np.array(l2(lof, lob)), "f32gap_grads_all": np.array(l2(cat(gf), cat(gb))),
              "f32gap_loss": np.array(abs(lossf - lossb) / abs(lossb)),
              "storage_logits": np.array(l2(lob, lo32)), "storage_grads_all": np.array(l2(cat(gb), cat(g32)))}
    names = []
    for n, p in model.named_parameters():
        names.append(n)
        arrays["param:" + n] = state0[n].to(torch.bfloat16).view(torch.int16).numpy()
        arrays["grad16:" + n], arrays["gscale:" + n] = _pack_scaled_f16(gb[n])
        arrays["noise:" + n] = np.array(l2(gp[n], gb[n]))
        arrays["f32gap:" + n] = np.array(l2(gf[n], gb[n]))
        arrays["storage:" + n] = np.array(l2(gb[n], g32[n]))
    for n, b in model.named_buffers():
        arrays["buf0:" + n] = state0[n].numpy()
        arrays["buf1:" + n] = bufb[n].float().numpy() if bufb[n].is_floating_point() else bufb[n].numpy()
    arrays["param_names"] = np.array(names)
    print("res-OCR bf16 storage: rounding-flip noise logits %.4f grads(all) %.4f | float32 arithmetic vs float64: logits %.4f "
          "grads(all) %.4f | storage vs f32: logits %.4f grads(all) %.4f" %
          (arrays["noise_logits"], arrays["noise_grads_all"], arrays["f32gap_logits"], arrays["f32gap_grads_all"],
           arrays["storage_logits"], arrays["storage_grads_all"]))
    model.float()
    if write:
        save("g9b_bf16_storage_res_ocr", **arrays)
    return arrays


def gen_loss_curve_bf16_storage(n_filters=16, runs=3):
    """g5c: gen_loss_curve's 200 steps of the reference UNet3p_SNws(14, 16, 10) with bf16 storage roundings
    (_Bf16Storage; master parameters and optimizer stay float32, as on the MI355X bf16 tier).  Run `runs` times, run r
    with the initial parameters moved by a relative r x 1e-6: how far apart the reference's own curves end up in that
    mode is the yardstick for any other implementation of it (stored as losses_perturbed)."""
    steps = 200
    crit = SegmentationLosses(weight=None, cuda=False).build_loss(mode="ce")
    batches = [synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
    curves = []
    for r in range(runs):
        model = UNet3p_SNws(14, n_filters, 10, 1, 1)
        det_init(model)
        if r:
            with torch.no_grad():
                for k, p in enumerate(model.parameters()):
                    p.mul_(1.0 + r * 1e-6 * t32(fill(tuple(p.shape), 300 + k, 1.0)))
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=5e-4)
        with contextlib.redirect_stdout(io.StringIO()):
            sched = LR_Scheduler("poly", 2e-3, 1, steps)
        losses = []
        for i in range(steps):
            img, lab = batches[i % 4]
            with contextlib.redirect_stdout(io.StringIO()):
                sched(opt, i, 0, 0.0)
            opt.zero_grad()
            with _Bf16Storage(model, 10):
                loss = crit(model(img.to(torch.bfloat16).float()), lab)
                loss.backward()
            opt.step()
            losses.append(loss.item())
            if i % 20 == 0:
                print("run", r, "step", i, "loss", losses[-1], flush=True)
        curves.append(losses)
    c = np.array(curves, np.float64)
    print("bf16-storage reference curves: max relative spread between runs %.4g, of the 10-step moving average %.4g"
          % (float((np.abs(c[1:] - c[0]) / c[0]).max()),
             float(max((np.abs(np.convolve(x, np.ones(10) / 10, "valid") - np.convolve(c[0], np.ones(10) / 10, "valid"))
                        / np.convolve(c[0], np.ones(10) / 10, "valid")).max() for x in c[1:]))))
    save("g5c_loss_curve_nf16_bf16_storage", losses=c[0], losses_perturbed=c[1:])


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
    if "g4odd" in which:    # UNet_SNws on an odd-sized input (CropConcat's zero-padding branch)
        gen_unet_odd()
    if "g5" in which:
        gen_loss_curve()
    if "g5b" in which:      # MFMA-aligned widths (16..256 channels) for the bf16 tier's 200-step curve
        gen_loss_curve(16, "g5b_loss_curve_nf16")
    if "g6" in which:
        gen_evaluator()
    if "g78" in which:
        gen_misc()
    if "g8w" in which:      # class-weighted criterion
        gen_weighted_ce()
    if "g9" in which:       # bf16-storage run of the reference (not part of the default list: takes a few minutes)
        gen_bf16_storage()
    if "g9b" in which:      # the same protocol for UNet3p_res_ocr_SNws
        gen_bf16_storage_ocr()
    if "g5c" in which:      # the reference's 200-step curve WITH bf16 storage, three times (self-noise of the curve)
        gen_loss_curve_bf16_storage()
