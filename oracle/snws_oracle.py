"""CPU oracle for the normalized-conv hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch-CPU / numpy arithmetic, the algorithms of the
reference's hot path so that the HIP kernels can be checked against them on any
machine.  It is imported only by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py - never by the product modules, which fail loudly
when the HIP library is missing.

Pinning: every function here is checked by tests/test_oracle_golden.py against
fixtures in tests/golden/*.npz that were produced by running the reference's
own modules (tests/golden/gen_golden.py).  Exception, "parity unpinned": the
SynchronizedBatchNorm statistics (`sync_bn_*`) follow
modeling/sync_batchnorm/batchnorm.py:45-65,103-115 as text only, because that
package cannot be imported (SyntaxError in comm.py:64) and the reference holds
no test or fixture for it.

Reference lines each function follows are cited in its docstring
(paths relative to the reference repository root).
"""

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# Weight standardisation + convolution
# ----------------------------------------------------------------------------
def ws_standardize(w):
    """modeling/weightstd.py:13-17.  w (O,I,kh,kw) -> (w-mean_o)/(std_o+1e-5), std unbiased over I*kh*kw."""
    O = w.shape[0]
    flat = w.reshape(O, -1)
    n = flat.shape[1]
    mu = flat.sum(1, keepdim=True) / n
    c = flat - mu
    s = torch.sqrt((c * c).sum(1, keepdim=True) / (n - 1))
    return (c / (s + 1e-5)).reshape(w.shape)


def ws_standardize_bwd(w, g):
    """Closed-form d(loss)/dw given g = d(loss)/d(w_hat); autograd of weightstd.py:13-17."""
    O = w.shape[0]
    flat = w.reshape(O, -1)
    gf = g.reshape(O, -1)
    n = flat.shape[1]
    mu = flat.sum(1, keepdim=True) / n
    c = flat - mu
    s = torch.sqrt((c * c).sum(1, keepdim=True) / (n - 1))
    d = s + 1e-5
    gc = (gf * c).sum(1, keepdim=True)
    t = gf / d - gc * c / (d * d * (n - 1) * s)
    return (t - t.sum(1, keepdim=True) / n).reshape(w.shape)


def conv2dws_fwd(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    """modeling/weightstd.py:12-18."""
    return F.conv2d(x, ws_standardize(w), b, stride, padding, dilation, groups)


def conv2dws_bwd(x, w, dy, has_bias=True, stride=1, padding=0, dilation=1, groups=1):
    """Gradients of conv2dws_fwd: dgrad, wgrad through the standardisation, bias grad."""
    what = ws_standardize(w)
    dx = torch.nn.grad.conv2d_input(x.shape, what, dy, stride, padding, dilation, groups)
    dwhat = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride, padding, dilation, groups)
    dw = ws_standardize_bwd(w, dwhat)
    db = dy.sum((0, 2, 3)) if has_bias else None
    return dx, dw, db


# ----------------------------------------------------------------------------
# SwitchNorm
# ----------------------------------------------------------------------------
class SNState:
    """Parameters + buffers of one SwitchNorm2d (modeling/snorm.py:6-35)."""

    def __init__(self, C, eps=1e-5, momentum=0.997, using_movavg=True, using_bn=True, last_gamma=False,
                 dtype=torch.float32):
        self.C, self.eps, self.momentum = C, eps, momentum
        self.using_movavg, self.using_bn = bool(using_movavg), bool(using_bn)
        self.gamma = torch.zeros(C, dtype=dtype) if last_gamma else torch.ones(C, dtype=dtype)
        self.beta = torch.zeros(C, dtype=dtype)
        k = 3 if using_bn else 2
        self.mean_weight = torch.ones(k, dtype=dtype)
        self.var_weight = torch.ones(k, dtype=dtype)
        self.running_mean = torch.zeros(C, dtype=dtype)
        self.running_var = torch.zeros(C, dtype=dtype)


def _softmax(v):
    e = torch.exp(v - v.max())
    return e / e.sum()


def switchnorm_fwd(x, st, training=True, allreduce=None, relu=False):
    """modeling/snorm.py:41-81 (+ the ReLU of UNet3p_SNws.py:32-33 when relu=True).

    allreduce: optional callable(tensor)->tensor summing a packed [2C+1] vector
    [sum_n mean_in, sum_n (var_in+mean_in^2), N_local] over data-parallel shards; with it the
    BN branch sees the global batch.  Mutates st.running_* like the module does.
    Returns (y, ctx) with ctx holding what switchnorm_bwd needs.
    """
    if x.dim() != 4:
        raise ValueError('expected 4D input (got {}D input)'.format(x.dim()))
    N, C, H, W = x.shape
    HW = H * W
    xf = x.reshape(N, C, HW)
    mean_in = xf.sum(-1) / HW                                    # (N,C)   snorm.py:45
    var_in = ((xf - mean_in[..., None]) ** 2).sum(-1) / (HW - 1)  # unbiased, snorm.py:46
    temp = var_in + mean_in ** 2                                 # snorm.py:49
    mean_ln = mean_in.sum(1, keepdim=True) / C                   # (N,1)   snorm.py:48
    var_ln = temp.sum(1, keepdim=True) / C - mean_ln ** 2        # snorm.py:50
    a = _softmax(st.mean_weight)
    b = _softmax(st.var_weight)
    Ng = N
    if st.using_bn:
        if training:
            pack = torch.cat([mean_in.sum(0), temp.sum(0), torch.tensor([float(N)], dtype=x.dtype)])
            if allreduce is not None:
                pack = allreduce(pack)
            Ng = float(pack[-1].detach())
            mean_bn = pack[:C] / Ng                              # snorm.py:54
            var_bn = pack[C:2 * C] / Ng - mean_bn ** 2           # snorm.py:55
            with torch.no_grad():
                if st.using_movavg:                              # snorm.py:57-60
                    st.running_mean.mul_(st.momentum).add_((1 - st.momentum) * mean_bn.detach())
                    st.running_var.mul_(st.momentum).add_((1 - st.momentum) * var_bn.detach())
                else:                                            # snorm.py:62-63
                    st.running_mean.add_(mean_bn.detach())
                    st.running_var.add_(mean_bn.detach() ** 2 + var_bn.detach())
        else:                                                    # snorm.py:65-66
            mean_bn = st.running_mean.to(x.dtype)
            var_bn = st.running_var.to(x.dtype)
        mean = a[0] * mean_in + a[1] * mean_ln + a[2] * mean_bn[None, :]   # snorm.py:73
        var = b[0] * var_in + b[1] * var_ln + b[2] * var_bn[None, :]       # snorm.py:74
    else:
        mean_bn = var_bn = None
        mean = a[0] * mean_in + a[1] * mean_ln                   # snorm.py:76
        var = b[0] * var_in + b[1] * var_ln                      # snorm.py:77
    r = 1.0 / torch.sqrt(var + st.eps)                           # snorm.py:79
    xhat = (xf - mean[..., None]) * r[..., None]
    z = xhat * st.gamma[None, :, None] + st.beta[None, :, None]  # snorm.py:81
    y = torch.relu(z) if relu else z
    ctx = dict(x=xf, xhat=xhat, r=r, mean_in=mean_in, var_in=var_in, mean_ln=mean_ln, var_ln=var_ln,
               mean_bn=mean_bn, var_bn=var_bn, a=a, b=b, z=z, relu=relu, training=training, Ng=Ng,
               shape=(N, C, H, W))
    return y.reshape(N, C, H, W), ctx


def switchnorm_bwd(dy, st, ctx, allreduce=None):
    """Closed-form backward of switchnorm_fwd.  Returns dx, dgamma, dbeta, dmean_weight, dvar_weight.

    allreduce: optional callable summing the packed [sum_n dm (C), sum_n dv (C)] over shards
    (data-parallel BN branch).  The parameter gradients returned are the LOCAL shard's share.
    """
    N, C, H, W = ctx['shape']
    HW = H * W
    dyf = dy.reshape(N, C, HW)
    if ctx['relu']:
        dyf = dyf * (ctx['z'] > 0).to(dyf.dtype)
    x, xhat, r = ctx['x'], ctx['xhat'], ctx['r']
    a, b = ctx['a'], ctx['b']
    P = dyf.sum(-1)
    Q = (dyf * xhat).sum(-1)
    dgamma = Q.sum(0)
    dbeta = P.sum(0)
    S1 = st.gamma[None, :] * P
    S2 = st.gamma[None, :] * Q
    dm = -r * S1
    dv = -0.5 * r * r * S2
    mean_in, var_in, mean_ln, var_ln = ctx['mean_in'], ctx['var_in'], ctx['mean_ln'], ctx['var_ln']
    use_bn_batch = st.using_bn and ctx['training']
    da = [(dm * mean_in).sum(), (dm * mean_ln).sum()]
    db = [(dv * var_in).sum(), (dv * var_ln).sum()]
    if st.using_bn:
        da.append((dm * ctx['mean_bn'][None, :]).sum())
        db.append((dv * ctx['var_bn'][None, :]).sum())
    da = torch.stack(da)
    db = torch.stack(db)
    dmw = a * (da - (a * da).sum())
    dvw = b * (db - (b * db).sum())
    dvar_ln = b[1] * dv.sum(1, keepdim=True)
    dmean_ln = a[1] * dm.sum(1, keepdim=True) - 2 * mean_ln * dvar_ln
    dtemp = dvar_ln / C
    dmean_in = a[0] * dm + dmean_ln / C
    if use_bn_batch:
        pack = torch.cat([dm.sum(0), dv.sum(0)])
        if allreduce is not None:
            pack = allreduce(pack)
        Ng = ctx['Ng']
        dvar_bn = b[2] * pack[C:]
        dmean_bn = a[2] * pack[:C] - 2 * ctx['mean_bn'] * dvar_bn
        dtemp = dtemp + dvar_bn[None, :] / Ng
        dmean_in = dmean_in + dmean_bn[None, :] / Ng
    dvar_in = b[0] * dv + dtemp
    dmean_in = dmean_in + 2 * mean_in * dtemp
    dx = (r * st.gamma[None, :])[..., None] * dyf + (dmean_in / HW)[..., None] \
        + (2 * dvar_in / (HW - 1))[..., None] * (x - mean_in[..., None])
    return dx.reshape(N, C, H, W), dgamma, dbeta, dmw, dvw


# ----------------------------------------------------------------------------
# Synchronized BatchNorm statistics (parity unpinned - see module docstring)
# ----------------------------------------------------------------------------
def sync_bn_mean_invstd(sum_, ssum, size, eps):
    """modeling/sync_batchnorm/batchnorm.py:103-115: mean, inv_std (biased var, clamp(eps)), unbiased var."""
    assert size > 1, 'BatchNorm computes unbiased standard-deviation, which requires size > 1.'
    mean = sum_ / size
    sumvar = ssum - sum_ * mean
    unbias_var = sumvar / (size - 1)
    bias_var = sumvar / size
    return mean, bias_var.clamp(min=eps) ** -0.5, unbias_var


def sync_bn_fwd(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, allreduce=None):
    """batchnorm.py:45-65 training path: per-channel sum/ssum over (N,HW), reduced over shards."""
    N, C = x.shape[0], x.shape[1]
    xf = x.reshape(N, C, -1)
    size = float(N * xf.shape[2])
    pack = torch.cat([xf.sum((0, 2)), (xf * xf).sum((0, 2)), torch.tensor([size], dtype=x.dtype)])
    if allreduce is not None:
        pack = allreduce(pack)
    mean, inv_std, unbias = sync_bn_mean_invstd(pack[:C], pack[C:2 * C], float(pack[-1]), eps)
    running_mean.mul_(1 - momentum).add_(momentum * mean)
    running_var.mul_(1 - momentum).add_(momentum * unbias)
    y = (xf - mean[None, :, None]) * (inv_std * gamma)[None, :, None] + beta[None, :, None]
    return y.reshape(x.shape)


# ----------------------------------------------------------------------------
# Resampling used by the full-scale skips
# ----------------------------------------------------------------------------
def maxpool_ceil(x, k):
    """nn.MaxPool2d(k, k, ceil_mode=True) as used at UNet3p_SNws.py:49-51: out = ceil(H/k), border windows
    clipped to the image; backward routes the gradient to the FIRST maximum of a window (row-major scan), which
    is what the stock op does - so the stock CPU op is used as the primitive here."""
    return F.max_pool2d(x, k, k, ceil_mode=True)


def _bilinear_axis(n_in, scale, dtype):
    o = torch.arange(n_in * scale, dtype=torch.float64)
    src = (o + 0.5) / scale - 0.5
    src = torch.clamp(src, min=0.0)
    i0 = torch.floor(src).to(torch.int64)
    i0 = torch.clamp(i0, max=n_in - 1)
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    lam = (src - i0.to(torch.float64)).to(dtype)
    return i0, i1, lam


def upsample_bilinear(x, s):
    """nn.Upsample(scale_factor=s, mode='bilinear', align_corners=False), UNet3p_SNws.py:52-55."""
    N, C, H, W = x.shape
    y0, y1, ly = _bilinear_axis(H, s, x.dtype)
    x0, x1, lx = _bilinear_axis(W, s, x.dtype)
    top = x[:, :, y0, :]
    bot = x[:, :, y1, :]
    rows = top * (1 - ly)[None, None, :, None] + bot * ly[None, None, :, None]
    left = rows[:, :, :, x0]
    right = rows[:, :, :, x1]
    return left * (1 - lx)[None, None, None, :] + right * lx[None, None, None, :]


def conv_of_upsampled_commuted(x, what, b, k, rnd=None):
    """conv2d(upsample_bilinear(x, k), what, b, padding=same) evaluated in the commuted order the GPU path uses:
    Z_t = W_t x at the source resolution, then y(p) = b + sum_t [p+t inside] up_k(Z_t)(p+t).  Equal to the direct
    order in exact arithmetic (tests/test_oracle_golden.py checks it); rnd optionally rounds Z and y where the GPU
    path stores them."""
    O_, I, kh, kw = what.shape
    pad = (kh - 1) // 2
    rnd = rnd or (lambda t: t)
    wre = what.permute(2, 3, 0, 1).reshape(kh * kw * O_, I, 1, 1)
    z = rnd(F.conv2d(x, wre))
    H, W = x.shape[2] * k, x.shape[3] * k
    acc = 0
    for t in range(kh * kw):
        u = F.pad(upsample_bilinear(z[:, t * O_:(t + 1) * O_], k), (pad, pad, pad, pad))
        dy, dx = t // kw, t % kw
        acc = acc + u[:, :, dy:dy + H, dx:dx + W]
    if b is not None:
        acc = acc + b.view(1, -1, 1, 1)
    return rnd(acc)


def crop_concat(a, b):
    """UNet_SNws.py:35-39: F.pad(a, (-(dX//2), -(dX-dX//2), -(dY//2), -(dY-dY//2))) then cat((a,b),1) - a negative
    pad crops (a larger than b: centre crop, diff//2 from the top/left), a positive one zero-pads (a smaller than b,
    which ceil-mode pooling followed by x2 up-sampling produces on odd sizes)."""
    dY = a.shape[2] - b.shape[2]
    dX = a.shape[3] - b.shape[3]
    a = F.pad(a, (-(dX // 2), -(dX - dX // 2), -(dY // 2), -(dY - dY // 2)))
    return torch.cat((a, b), dim=1)


# ----------------------------------------------------------------------------
# Loss, LR, tiling
# ----------------------------------------------------------------------------
def ce_loss_batch_avg(logit, target, ignore_index=255, weight=None):
    """utils/loss.py:146-159: mean CE over non-ignored pixels, then divided by the batch size.  With class weights
    (utils/loss.py:148, nn.CrossEntropyLoss(weight=w, reduction='mean')) the mean is the weighted one:
    sum w[t] * ce / sum w[t] over the non-ignored pixels."""
    n, c, h, w = logit.shape
    t = target.long()
    lse = torch.logsumexp(logit, dim=1)
    valid = t != ignore_index
    tc = torch.where(valid, t, torch.zeros_like(t))
    picked = torch.gather(logit, 1, tc[:, None]).squeeze(1)
    wt = valid.to(logit.dtype) if weight is None else weight.to(logit.dtype)[tc] * valid.to(logit.dtype)
    per = (lse - picked) * wt
    return per.sum() / wt.sum() / n


def poly_lr(base_lr, T, N):
    """utils/lr_scheduler.py:44,48."""
    return base_lr * pow((1 - 1.0 * T / N), 0.9)


def gen_tiles_offs(xsize, ysize, block, overlap):
    """utils/gen_tiles_offs.py:3-27: x-major list of (xoff, yoff); last tile clamped to size-block."""
    def axis(size):
        step = block - overlap
        n = int((size - block) / step) + 1
        offs = [step * j for j in range(n)] + [size - block]
        if len(offs) >= 2 and offs[-1] == offs[-2]:
            offs.pop()
        return offs
    return [(xo, yo) for xo in axis(xsize) for yo in axis(ysize)]


# ----------------------------------------------------------------------------
# Evaluator (confusion matrix + metrics)
# ----------------------------------------------------------------------------
def confusion_matrix(gt, pred, num_class):
    """utils/metrics.py:40-45.  Rows = ground truth, cols = prediction.  Pixels whose label is outside
    [0, num_class) (or NaN) are dropped; the label is truncated toward zero like astype('int').
    A prediction that sends num_class*gt+pred outside [0, num_class^2) makes the reference raise."""
    gt = np.asarray(gt).reshape(-1)
    pred = np.asarray(pred).reshape(-1)
    cm = np.zeros((num_class, num_class), dtype=np.int64)
    with np.errstate(invalid='ignore'):
        keep = (gt >= 0) & (gt < num_class)
    g = gt[keep].astype(np.int64)
    p = pred[keep].astype(np.int64)
    idx = num_class * g + p
    if idx.size and (idx.min() < 0 or idx.max() >= num_class * num_class):
        raise ValueError('prediction outside the confusion matrix')
    np.add.at(cm.reshape(-1), idx, 1)
    return cm


def evaluator_metrics(cm):
    """utils/metrics.py:9-38 on a (C,C) matrix: PA, per-class PA, mIoU (nanmean), FWIoU."""
    cm = np.asarray(cm, dtype=np.float64)
    with np.errstate(all='ignore'):
        diag = np.diag(cm)
        pa = diag.sum() / cm.sum()
        pac = diag / cm.sum(axis=1)
        iu = diag / (cm.sum(axis=1) + cm.sum(axis=0) - diag)
        miou = np.nanmean(iu)
        freq = cm.sum(axis=1) / cm.sum()
        fwiou = (freq[freq > 0] * iu[freq > 0]).sum()
    return pa, pac, miou, fwiou


# ----------------------------------------------------------------------------
# Input-side tile preparation (dataloaders/datasets/cloud_dataset.py:127-134)
# ----------------------------------------------------------------------------
def normalize_tile(img_tif):
    """cloud_dataset.py:127-131 on one (C, H, W) float32 numpy tile: NaN -> 0, min-max over the whole tile, x 255."""
    img_tif = np.array(img_tif, dtype=np.float32, copy=True)
    img_tif[np.isnan(img_tif)] = 0
    imax = img_tif.max()
    imin = img_tif.min()
    with np.errstate(all="ignore"):
        img_tif = (img_tif - imin) / (imax - imin)
        img_tif *= 255
    return img_tif


def remap_label(mask):
    """cloud_dataset.py:132-137: NaN -> 255, 10 -> 0, 255 -> 0, as float32."""
    mask = np.array(mask, copy=True)
    if mask.dtype.kind == "f":
        mask[np.isnan(mask)] = 255
    mask[mask == 10] = 0
    mask[mask == 255] = 0
    return torch.from_numpy(mask).float().numpy()


# ----------------------------------------------------------------------------
# Whole models as functions of a reference-compatible state_dict
# ----------------------------------------------------------------------------
class OracleNet:
    """Runs UNet_SNws / UNet3p_SNws forward from a state_dict with reference key names
    ('convd1_1.0.weight', 'convd1_1.1.running_mean', ...).  Parameters are torch tensors with
    requires_grad so autograd gives the gradients; buffers are updated in call order like the module."""

    def __init__(self, state, using_movavg=True, using_bn=True, allreduce=None, act_round=None):
        self.p = state
        self.using_movavg, self.using_bn = using_movavg, using_bn
        self.training = True
        self.allreduce = allreduce
        # act_round: optional storage dtype (torch.bfloat16) to mimic where the GPU path rounds: the standardized
        # filter, every conv output, every SwitchNorm output and every resampled map (straight-through in backward)
        self.act_round = act_round

    def rnd(self, t):
        if self.act_round is None:
            return t
        return t + (t.detach().to(self.act_round).to(t.dtype) - t.detach())

    def csr(self, name, x):
        """ConvSnRelu block (UNet3p_SNws.py:29-33): Sequential(0=Conv2dws, 1=SwitchNorm2d, 2=ReLU)."""
        p = self.p
        w = p[name + '.0.weight']
        pad = (w.shape[2] - 1) // 2
        y = self.rnd(F.conv2d(x, self.rnd(ws_standardize(w)), p[name + '.0.bias'], 1, pad))
        st = SNState.__new__(SNState)
        st.C, st.eps, st.momentum = w.shape[0], 1e-5, 0.997
        st.using_movavg, st.using_bn = self.using_movavg, self.using_bn
        st.gamma = p[name + '.1.weight'].reshape(-1)
        st.beta = p[name + '.1.bias'].reshape(-1)
        st.mean_weight, st.var_weight = p[name + '.1.mean_weight'], p[name + '.1.var_weight']
        st.running_mean = p[name + '.1.running_mean'].view(-1)
        st.running_var = p[name + '.1.running_var'].view(-1)
        out, _ = switchnorm_fwd(y, st, self.training, self.allreduce, relu=True)
        return self.rnd(out)

    def head(self, x):
        return self.rnd(F.conv2d(x, self.rnd(self.p['output_seg_map.weight']), self.rnd(self.p['output_seg_map.bias'])))

    def unet3p(self, x):
        """UNet3p_SNws.py:96-160."""
        c, mp = self.csr, maxpool_ceil
        up = lambda t, k: self.rnd(upsample_bilinear(t, k))  # noqa: E731
        d1 = c('convd1_2', c('convd1_1', x))
        d2 = c('convd2_2', c('convd2_1', mp(d1, 2)))
        d3 = c('convd3_2', c('convd3_1', mp(d2, 2)))
        d4 = c('convd4_2', c('convd4_1', mp(d3, 2)))
        c5 = c('conv5_2', c('conv5_1', mp(d4, 2)))
        return self.head(self.full_scale_decoder(d1, d2, d3, d4, c5))

    def full_scale_decoder(self, d1, d2, d3, d4, c5):
        """UNet3p_SNws.py:127-156 (the same lines as UNet3p_res_ocr_SNws.py:327-361)."""
        c, mp = self.csr, maxpool_ceil
        up = lambda t, k: self.rnd(upsample_bilinear(t, k))  # noqa: E731
        u4 = c('conv_fusion', torch.cat((c('convd1_cat', mp(d1, 8)), c('convd2_cat', mp(d2, 4)),
                                         c('convd3_cat', mp(d3, 2)), c('convd4_cat', d4),
                                         c('conv5_cat', up(c5, 2))), 1))
        u3 = c('conv_fusion', torch.cat((c('convd1_cat', mp(d1, 4)), c('convd2_cat', mp(d2, 2)),
                                         c('convd3_cat', d3), c('convu_cat', up(u4, 2)),
                                         c('conv5_cat', up(c5, 4))), 1))
        u2 = c('conv_fusion', torch.cat((c('convd1_cat', mp(d1, 2)), c('convd2_cat', d2),
                                         c('convu_cat', up(u3, 2)), c('convu_cat', up(u4, 4)),
                                         c('conv5_cat', up(c5, 8))), 1))
        u1 = c('conv_fusion', torch.cat((c('convd1_cat', d1), c('convu_cat', up(u2, 2)),
                                         c('convu_cat', up(u3, 4)), c('convu_cat', up(u4, 8)),
                                         c('conv5_cat', up(c5, 16))), 1))
        return u1

    # ---- UNet3p_res_ocr_SNws (UNet3p_res_ocr_SNws.py:211-377)
    def _sn(self, y, prefix, relu):
        p = self.p
        st = SNState.__new__(SNState)
        st.C, st.eps, st.momentum = y.shape[1], 1e-5, 0.997
        st.using_movavg, st.using_bn = self.using_movavg, self.using_bn
        st.gamma, st.beta = p[prefix + '.weight'].reshape(-1), p[prefix + '.bias'].reshape(-1)
        st.mean_weight, st.var_weight = p[prefix + '.mean_weight'], p[prefix + '.var_weight']
        st.running_mean, st.running_var = p[prefix + '.running_mean'].view(-1), p[prefix + '.running_var'].view(-1)
        out, _ = switchnorm_fwd(y, st, self.training, self.allreduce, relu=relu)
        return out

    def basic(self, name, x):
        """Resnet_BasicBlock with downsample=None (UNet3p_res_ocr_SNws.py:134-161): conv1-bn1-relu-conv2-bn2,
        += the block input, relu."""
        p = self.p
        y = self.rnd(F.conv2d(x, self.rnd(ws_standardize(p[name + '.conv1.weight'])), p[name + '.conv1.bias'], 1, 1))
        y = self.rnd(self._sn(y, name + '.bn1', True))
        y = self.rnd(F.conv2d(y, self.rnd(ws_standardize(p[name + '.conv2.weight'])), p[name + '.conv2.bias'], 1, 1))
        return self.rnd(torch.relu(self._sn(y, name + '.bn2', False) + x))

    def conv_plain(self, name, x):
        w = self.p[name + '.weight']
        b = self.p.get(name + '.bias')
        return self.rnd(F.conv2d(x, self.rnd(w), b, 1, (w.shape[2] - 1) // 2))

    def bn_relu(self, name, x):
        """nn.BatchNorm2d (batch statistics of THIS process, momentum 0.1) + ReLU."""
        p = self.p
        y = F.batch_norm(x, p[name + '.running_mean'], p[name + '.running_var'], p[name + '.weight'], p[name + '.bias'],
                         self.training, 0.1, 1e-5)
        if self.training:
            p[name + '.num_batches_tracked'] += 1
        return self.rnd(torch.relu(y))

    def cbr(self, seq, x, i=0):
        """Sequential(..., i: Conv2d, i+1: BatchNorm2d, i+2: ReLU, ...)."""
        return self.bn_relu('%s.%d' % (seq, i + 1), self.conv_plain('%s.%d' % (seq, i), x))

    def ocr_head(self, u1):
        """UNet3p_res_ocr_SNws.py:363-371 with SpatialGather_Module :21-28, _ObjectAttentionBlock :82-104 (scale 1)
        and SpatialOCR_Module :129-132 (Dropout2d taken as identity: p = 0 or eval mode)."""
        B = u1.shape[0]
        out_aux = self.conv_plain('aux_head.3', self.cbr('aux_head', u1))
        feats = self.cbr('conv3x3_ocr', u1)
        C, H, W = feats.shape[1:]
        probs = self.rnd(torch.softmax(out_aux.reshape(B, out_aux.shape[1], -1), dim=2))       # B x K x HW
        proxy = self.rnd(torch.matmul(probs, feats.reshape(B, C, -1).permute(0, 2, 1)))        # B x K x C
        proxy = proxy.permute(0, 2, 1).unsqueeze(3)                                            # B x C x K x 1
        blk = 'ocr_distri_head.object_context_block.'
        kc = self.p[blk + 'f_pixel.0.weight'].shape[0]
        query = self.cbr(blk + 'f_pixel', self.cbr(blk + 'f_pixel', feats), 3).reshape(B, kc, -1).permute(0, 2, 1)
        key = self.cbr(blk + 'f_object', self.cbr(blk + 'f_object', proxy), 3).reshape(B, kc, -1)
        value = self.cbr(blk + 'f_down', proxy).reshape(B, kc, -1).permute(0, 2, 1)
        sim = self.rnd(torch.softmax(torch.matmul(query, key) * kc ** -0.5, dim=-1))           # B x HW x K
        ctx = self.rnd(torch.matmul(sim, value)).permute(0, 2, 1).reshape(B, kc, H, W)
        ctx = self.cbr(blk + 'f_up', ctx)
        feats = self.cbr('ocr_distri_head.conv_bn_dropout', torch.cat([ctx, feats], 1))
        return self.conv_plain('cls_head', feats)

    def unet3p_res_ocr(self, x):
        c, mp, rb = self.csr, maxpool_ceil, self.basic
        up = lambda t, k: self.rnd(upsample_bilinear(t, k))  # noqa: E731
        d1 = rb('convd1_2', c('convd1_1', x))
        d2 = rb('convd2_2', c('convd2_1', mp(d1, 2)))
        d3 = rb('convd3_2', c('convd3_1', mp(d2, 2)))
        d4 = rb('convd4_2', c('convd4_1', mp(d3, 2)))
        c5 = rb('conv5_2', c('conv5_1', mp(d4, 2)))
        return self.ocr_head(self.full_scale_decoder(d1, d2, d3, d4, c5))

    def unet(self, x):
        """UNet_SNws.py:95-152."""
        c, mp = self.csr, maxpool_ceil
        up = lambda t, k: self.rnd(upsample_bilinear(t, k))  # noqa: E731
        c1 = c('conv1_2', c('conv1_1', x))
        c2 = c('conv2_2', c('conv2_1', mp(c1, 2)))
        c3 = c('conv3_2', c('conv3_1', mp(c2, 2)))
        c4 = c('conv4_2', c('conv4_1', mp(c3, 2)))
        x = c('conv_d2', c('conv_d1', mp(c4, 2)))
        x = c('conv_u3', c('conv_u2', crop_concat(c4, c('conv_u1', up(x, 2)))))
        x = c('conv5_3', c('conv5_2', crop_concat(c3, c('conv5_1', up(x, 2)))))
        x = c('conv6_3', c('conv6_2', crop_concat(c2, c('conv6_1', up(x, 2)))))
        x = c('conv7_3', c('conv7_2', crop_concat(c1, c('conv7_1', up(x, 2)))))
        return self.head(x)


UNET3P_BLOCKS = [  # (name, cin multiple of n_filters or 'in', cout multiple) - UNet3p_SNws.py:58-91
    ('convd1_1', 'in', 1), ('convd1_2', 1, 1), ('convd2_1', 1, 2), ('convd2_2', 2, 2),
    ('convd3_1', 2, 4), ('convd3_2', 4, 4), ('convd4_1', 4, 8), ('convd4_2', 8, 8),
    ('conv5_1', 8, 16), ('conv5_2', 16, 16),
    ('convd1_cat', 1, 1), ('convd2_cat', 2, 1), ('convd3_cat', 4, 1), ('convd4_cat', 8, 1),
    ('conv5_cat', 16, 1), ('convu_cat', 5, 1), ('conv_fusion', 5, 5)]

UNET_BLOCKS = [  # (name, cin, cout, k) - UNet_SNws.py:50-90
    ('conv1_1', 'in', 1, 3), ('conv1_2', 1, 1, 3), ('conv2_1', 1, 2, 3), ('conv2_2', 2, 2, 3),
    ('conv3_1', 2, 4, 3), ('conv3_2', 4, 4, 3), ('conv4_1', 4, 8, 3), ('conv4_2', 8, 8, 3),
    ('conv_d1', 8, 16, 3), ('conv_d2', 16, 16, 3),
    ('conv_u1', 16, 8, 1), ('conv_u2', 16, 8, 3), ('conv_u3', 8, 8, 3),
    ('conv5_1', 8, 4, 1), ('conv5_2', 8, 4, 3), ('conv5_3', 4, 4, 3),
    ('conv6_1', 4, 2, 1), ('conv6_2', 4, 2, 3), ('conv6_3', 2, 2, 3),
    ('conv7_1', 2, 1, 1), ('conv7_2', 2, 1, 3), ('conv7_3', 1, 1, 3)]


def make_state(arch, n_channels, n_filters, n_class, dtype=torch.float32):
    """Zero-initialised state_dict with the reference's key order and shapes (SwitchNorm buffers (1,C,1),
    gamma/beta (1,C,1,1)); fill it with tests/golden/patterns.det_init-style values or random ones."""
    st = {}
    if arch == 'unet3p':
        blocks = [(n, ci, co, 3) for (n, ci, co) in UNET3P_BLOCKS]
        head_in = 5 * n_filters
    elif arch == 'unet':
        blocks = UNET_BLOCKS
        head_in = n_filters
    elif arch == 'unet3p_res_ocr':
        return _make_state_res_ocr(n_channels, n_filters, n_class, dtype)
    else:
        raise ValueError(arch)
    for name, ci, co, k in blocks:
        cin = n_channels if ci == 'in' else ci * n_filters
        cout = co * n_filters
        st[name + '.0.weight'] = torch.zeros(cout, cin, k, k, dtype=dtype)
        st[name + '.0.bias'] = torch.zeros(cout, dtype=dtype)
        st[name + '.1.weight'] = torch.ones(1, cout, 1, 1, dtype=dtype)
        st[name + '.1.bias'] = torch.zeros(1, cout, 1, 1, dtype=dtype)
        st[name + '.1.mean_weight'] = torch.ones(3, dtype=dtype)
        st[name + '.1.var_weight'] = torch.ones(3, dtype=dtype)
        st[name + '.1.running_mean'] = torch.zeros(1, cout, 1, dtype=dtype)
        st[name + '.1.running_var'] = torch.zeros(1, cout, 1, dtype=dtype)
    st['output_seg_map.weight'] = torch.zeros(n_class, head_in, 1, 1, dtype=dtype)
    st['output_seg_map.bias'] = torch.zeros(n_class, dtype=dtype)
    return st


def _sn_entries(st, prefix, c, dtype):
    st[prefix + '.weight'] = torch.ones(1, c, 1, 1, dtype=dtype)
    st[prefix + '.bias'] = torch.zeros(1, c, 1, 1, dtype=dtype)
    st[prefix + '.mean_weight'] = torch.ones(3, dtype=dtype)
    st[prefix + '.var_weight'] = torch.ones(3, dtype=dtype)
    st[prefix + '.running_mean'] = torch.zeros(1, c, 1, dtype=dtype)
    st[prefix + '.running_var'] = torch.zeros(1, c, 1, dtype=dtype)


def _bn_entries(st, prefix, c, dtype):
    st[prefix + '.weight'] = torch.ones(c, dtype=dtype)
    st[prefix + '.bias'] = torch.zeros(c, dtype=dtype)
    st[prefix + '.running_mean'] = torch.zeros(c, dtype=dtype)
    st[prefix + '.running_var'] = torch.ones(c, dtype=dtype)
    st[prefix + '.num_batches_tracked'] = torch.zeros((), dtype=torch.int64)


def _make_state_res_ocr(n_channels, f, n_class, dtype):
    """state_dict of UNet3p_res_ocr_SNws in the reference's registration order (UNet3p_res_ocr_SNws.py:226-294)."""
    st = {}

    def csr(name, cin, cout):
        st[name + '.0.weight'] = torch.zeros(cout, cin, 3, 3, dtype=dtype)
        st[name + '.0.bias'] = torch.zeros(cout, dtype=dtype)
        _sn_entries(st, name + '.1', cout, dtype)

    def basic(name, c):
        for i in ('1', '2'):
            st[name + '.conv' + i + '.weight'] = torch.zeros(c, c, 3, 3, dtype=dtype)
            st[name + '.conv' + i + '.bias'] = torch.zeros(c, dtype=dtype)
            _sn_entries(st, name + '.bn' + i, c, dtype)

    def conv(name, cin, cout, k, bias):
        st[name + '.weight'] = torch.zeros(cout, cin, k, k, dtype=dtype)
        if bias:
            st[name + '.bias'] = torch.zeros(cout, dtype=dtype)

    widths = [n_channels, f, 2 * f, 4 * f, 8 * f, 16 * f]
    for i, name in enumerate(['convd1', 'convd2', 'convd3', 'convd4', 'conv5']):
        csr(name + '_1', widths[i], widths[i + 1])
        basic(name + '_2', widths[i + 1])
    mid, key = 512, 256
    conv('aux_head.0', 5 * f, 5 * f, 1, True); _bn_entries(st, 'aux_head.1', 5 * f, dtype)
    conv('aux_head.3', 5 * f, n_class, 1, True)
    conv('conv3x3_ocr.0', 5 * f, mid, 3, True); _bn_entries(st, 'conv3x3_ocr.1', mid, dtype)
    blk = 'ocr_distri_head.object_context_block.'
    for branch in ('f_pixel', 'f_object'):
        conv(blk + branch + '.0', mid, key, 1, False); _bn_entries(st, blk + branch + '.1', key, dtype)
        conv(blk + branch + '.3', key, key, 1, False); _bn_entries(st, blk + branch + '.4', key, dtype)
    conv(blk + 'f_down.0', mid, key, 1, False); _bn_entries(st, blk + 'f_down.1', key, dtype)
    conv(blk + 'f_up.0', key, mid, 1, False); _bn_entries(st, blk + 'f_up.1', mid, dtype)
    conv('ocr_distri_head.conv_bn_dropout.0', 2 * mid, mid, 1, False)
    _bn_entries(st, 'ocr_distri_head.conv_bn_dropout.1', mid, dtype)
    conv('cls_head', mid, n_class, 1, True)
    for name, cin in (('convd1_cat', f), ('convd2_cat', 2 * f), ('convd3_cat', 4 * f), ('convd4_cat', 8 * f),
                      ('conv5_cat', 16 * f), ('convu_cat', 5 * f)):
        csr(name, cin, f)
    csr('conv_fusion', 5 * f, 5 * f)
    conv('output_seg_map', f, n_class, 1, True)     # registered but unused by forward (:294)
    return st


def is_buffer(key):
    return key.endswith('running_mean') or key.endswith('running_var') or key.endswith('num_batches_tracked')


def model_step(arch, state, img, target, training=True, allreduce=None, loss_div=None, act_round=None, dlogits=None):
    """One forward (+ backward when training) of the oracle model.  Returns logits, loss, grads dict.  With dlogits
    the scalar that is differentiated is sum(logits * dlogits) instead of the CE loss (inputs whose output size
    differs from the label size: the reference's CropConcat zero-pads odd sizes upwards)."""
    params = {k: (v if is_buffer(k) else v.detach().clone().requires_grad_(training)) for k, v in state.items()}
    net = OracleNet(params, allreduce=allreduce, act_round=act_round)
    net.training = training
    logits = {'unet3p': net.unet3p, 'unet': net.unet, 'unet3p_res_ocr': net.unet3p_res_ocr}[arch](img)
    loss = ce_loss_batch_avg(logits, target) if dlogits is None else (logits * dlogits).sum()
    if loss_div is not None:
        loss = loss * (logits.shape[0] / loss_div)
    grads = {}
    if training:
        loss.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in params.items() if not is_buffer(k)}
    return logits.detach(), float(loss.detach()), grads
