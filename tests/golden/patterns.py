"""Closed-form input patterns shared by the golden generator and the tests.

Every tensor fed to the reference when the fixtures were generated came from
these functions, so a test can rebuild the same inputs without storing them.
"""
import numpy as np
import torch


def fill(shape, k, scale=1.0, offset=0.0):
    """Deterministic pattern: offset + scale*cos(0.37*(i+1) + k) over the flat index i."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    return (offset + scale * np.cos(0.37 * (i + 1.0) + float(k))).reshape(shape)


def label_pattern(shape, n_class=10, ignore_every=0):
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.int64)
    lab = ((i * 7 + i // 13 + (i // 101) * 3) % n_class).astype(np.float32)
    if ignore_every:
        lab[::ignore_every] = 255.0
    return lab.reshape(shape)


def t32(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32).copy())


def det_init(model):
    """Overwrite every parameter with the closed-form pattern (buffers stay zero)."""
    with torch.no_grad():
        for k, (name, p) in enumerate(model.named_parameters()):
            if name.endswith("mean_weight") or name.endswith("var_weight"):
                p.copy_(t32(fill(tuple(p.shape), k, 0.7, 0.2)))
            elif p.dim() == 4 and p.shape[0] == 1:          # SwitchNorm gamma / beta (1,C,1,1)
                if name.endswith(".weight"):
                    p.copy_(t32(fill(tuple(p.shape), k, 0.25, 1.0)))
                else:
                    p.copy_(t32(fill(tuple(p.shape), k, 0.1)))
            elif p.dim() == 4:                               # conv weight
                p.copy_(t32(fill(tuple(p.shape), k, 0.5)))
            else:                                            # conv bias
                p.copy_(t32(fill(tuple(p.shape), k, 0.1)))


def det_init_ocr(model):
    """det_init, then better-conditioned values for what UNet3p_res_ocr_SNws adds to the SNws blocks: BatchNorm2d
    gammas around 1 and the plain (not standardized) nn.Conv2d filters scaled by 1.5/sqrt(fan_in)."""
    det_init(model)
    plain = ("aux_head.", "conv3x3_ocr.", "ocr_distri_head.", "cls_head.", "output_seg_map.")
    with torch.no_grad():
        for k, (name, p) in enumerate(model.named_parameters()):
            if not name.startswith(plain):
                continue
            if p.dim() == 4:
                p.copy_(t32(fill(tuple(p.shape), k, 1.5 / np.sqrt(p.shape[1] * p.shape[2] * p.shape[3]))))
            elif name.endswith(".weight"):                   # BatchNorm2d gamma (C,)
                p.copy_(t32(fill(tuple(p.shape), k, 0.25, 1.0)))


def synth_batch(B, C, S, k):
    img = fill((B, C, S, S), k, 127.5, 127.5) * (0.6 + 0.4 * np.cos(0.011 * np.arange(S))[None, None, None, :])
    lab = label_pattern((B, S, S), 10, ignore_every=97)
    return t32(img), t32(lab)


def synth_batch_learnable(B, C, S, k):
    """Images as synth_batch; labels are a 10-level quantisation of a smooth mix of two bands, so the
    200-step curve has something to learn."""
    img, lab = synth_batch(B, C, S, k)
    a = img.numpy().astype(np.float64)
    sm = a[:, 0] * 0.5 + a[:, 3] * 0.5
    # 5x5 box blur keeps the target low-frequency
    pad = np.pad(sm, ((0, 0), (2, 2), (2, 2)), mode="edge")
    acc = np.zeros_like(sm)
    for dy in range(5):
        for dx in range(5):
            acc += pad[:, dy:dy + S, dx:dx + S]
    q = np.clip(np.floor(acc / 25.0 / 25.6), 0, 9).astype(np.float32)
    q[:, ::31, ::29] = 255.0
    return img, t32(q)


def load_bf16_storage_fixture(g):
    """g9_bf16_storage.npz -> (state_dict of float32 tensors whose parameter values are exact bf16 numbers, image,
    label, expected gradients as float64 tensors).  Parameters are stored as bf16 bit patterns, the expected
    gradients as per-tensor scale x float16."""
    state, grads = {}, {}
    for n in [str(k) for k in g["param_names"]]:
        state[n] = torch.from_numpy(g["param:" + n].copy()).view(torch.bfloat16).float()
        grads[n] = torch.from_numpy(g["grad16:" + n].astype(np.float64)) * float(g["gscale:" + n])
    for k in g.files:
        if k.startswith("buf0:"):
            state[k[5:]] = torch.from_numpy(g[k].copy())
    img, lab = synth_batch_learnable(4, 14, 64, 211)
    return state, img.to(torch.bfloat16).float(), lab, grads
