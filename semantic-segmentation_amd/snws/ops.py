"""Host-side plumbing over the libsnws C ABI: tensor layout handling, kernel-path selection and the
torch.autograd.Functions behind Conv2dws / SwitchNorm2d / the fused ConvSnRelu block.

Tensors keep the reference's logical NCHW shape; physically they are NHWC ("channels last"), which is what
the kernels read.  Everything here requires a GPU tensor: there is no CPU path.
"""
import ctypes
import os

import torch

from . import _lib as L

_DT = {torch.float32: L.SNWS_F32, torch.bfloat16: L.SNWS_BF16}


# When PROFILE is a list, every hot-path kernel launch is bracketed by two HIP events recorded on the launch
# stream and appended as (tag, algorithmic work, start, end); bench.py reads the durations after the timed region.
PROFILE = None


# Standardized-filter cache control.  By default a Conv2dws standardizes its filter on every call, exactly like the
# reference.  A training driver that knows when weights change (snws.harness.Trainer) switches the cache on for the
# Conv2dws modules of ITS model (module.cache_filters = True) and every weight update is announced with
# weights_changed() - the Trainer after its optimizer step, FlatAdamW.step / load_state_dict and
# checkpoint.load_checkpoint by themselves; then a module shared by several call sites standardizes once per step.
# (Tensor version counters are not a safe key on their own: fused optimizers update in place without bumping them.)
# Modules outside such a driver never cache, whatever other drivers exist in the process.
WEIGHT_EPOCH = 0


def weights_changed():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def filter_cache_key(w, want_mfma, enabled):
    return (WEIGHT_EPOCH, w._version, w.data_ptr(), str(w.device), bool(want_mfma)) if enabled else None


def set_filter_cache(model, enabled):
    """Switch the standardized-filter cache of every Conv2dws under `model` on or off."""
    for m in model.modules():
        if hasattr(m, "cache_filters"):
            m.cache_filters = bool(enabled)
            if not enabled:
                m._prep = m._prep_key = m._taps = m._taps_key = None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Gradient sinks.  An optimizer that keeps every gradient in one flat buffer (snws.optim.FlatAdamW) registers, per
# parameter, where that parameter's gradient lives.  A backward kernel then writes its result straight into the
# flat buffer: grad_out() hands out a fresh view of the slot (zero-filled by the optimizer's zero_grad) once per
# step, autograd adopts the returned view as .grad without launching a copy or an add, and a parameter used a second
# time in the same step gets an ordinary tensor that autograd accumulates as usual.
# parameter data_ptr -> (flat gradient, offset, numel, [handed out this step], weakref to the owning optimizer)
GRAD_SINKS = {}


def grad_key(param):
    return None if param is None else param.data_ptr()


def grad_out(key, shape, device, zero=False, acc_ok=False):
    """Where a backward kernel should write d(loss)/d(parameter `key` = grad_key(param)): the parameter's slot in
    the flat gradient buffer when one is registered and still unused in this step, else a new float32 tensor of
    `shape` (zero-filled on request; a slot always is)."""
    sink = GRAD_SINKS.get(key) if key is not None else None
    if sink is not None and sink[4]() is None:      # its optimizer is gone: the address may belong to anyone now
        del GRAD_SINKS[key]
        sink = None
    if sink is not None and not sink[3][0] and sink[2] == _numel(shape):
        sink[3][0] = True
        flat, off, n = sink[:3]
        return flat[off:off + n].view(shape)
    if sink is not None and sink[3][0] and sink[2] == _numel(shape):
        # a further use of the parameter in this step
        if acc_ok:          # the kernel can add: hand it the slot itself; grad_fold then returns None to autograd
            flat, off, n = sink[:3]
            t = flat[off:off + n].view(shape)
            _FOLD[id(t)] = (t, None)
            return t
        t = (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=device)
        _FOLD[id(t)] = (t, sink)        # grad_fold adds it into the slot
        return t
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=device)


def grad_accumulates(t):
    """True when t (from grad_out(..., acc_ok=True)) is a slot that already holds this step's earlier gradient: the
    kernel must add to it."""
    e = _FOLD.get(id(t))
    return e is not None and e[0] is t and e[1] is None


_FOLD = {}


def grad_fold(t):
    """Call on a tensor obtained from grad_out once its kernel has been enqueued, and return the result to autograd.
    The first gradient of a parameter in a step lives in its flat slot and is returned as is; a further one (a
    module used at several places) is added into the slot here and None is returned, so autograd neither sums nor
    copies anything for that parameter."""
    if t is None:
        return None
    entry = _FOLD.pop(id(t), None)
    if entry is None or entry[0] is not t:
        return t
    if entry[1] is not None:
        flat, off, n = entry[1][:3]
        flat[off:off + n].view(t.shape).add_(t)
    return None


def _view(t, shape):
    return None if t is None else t.view(shape)


def _numel(shape):
    n = 1
    for v in shape:
        n *= int(v)
    return n


PROFILE_ONLY = None     # optional tuple of tag prefixes: only those launches are bracketed (events cost host time)


def _launch(tag, work, name, *args):
    if PROFILE is None or (PROFILE_ONLY is not None and not tag.startswith(PROFILE_ONLY)):
        return L.call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = L.call(name, *args)
    e1.record()
    PROFILE.append((tag, work, e0, e1))
    return rc


def _timed(tag, work, fn):
    """fn() bracketed by events like _launch, for work that is not a libsnws call (the library GEMMs)."""
    if PROFILE is None or (PROFILE_ONLY is not None and not tag.startswith(PROFILE_ONLY)):
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.append((tag, work, e0, e1))
    return r


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s runs on the MI355X HIP kernels only; got a %s tensor (there is no CPU fallback)"
                           % (what, t.device))


def dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("snws kernels take float32 or bfloat16 activations, got %s" % t.dtype)


def as_nhwc(x):
    """(N,C,H,W) logical tensor -> (tensor whose storage is NHWC starting at data_ptr, pixel stride ld).
    A channels-last tensor or a channel slice of one is used in place; anything else is copied."""
    N, C, H, W = x.shape
    s = x.stride()
    ld = s[3] if W > 1 else (s[2] if H > 1 else (s[0] if N > 1 else C))
    ok = (C == 1 or s[1] == 1) and (W == 1 or s[3] == ld) and (H == 1 or s[2] == W * ld) and \
         (N == 1 or s[0] == H * W * ld) and ld >= C
    if ok:
        return x, ld
    xc = x.contiguous(memory_format=torch.channels_last)
    if xc.stride()[1] != 1 and C > 1:  # degenerate shapes where torch keeps NCHW strides
        xc = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return xc, C


def new_nhwc(N, C, H, W, dtype, device):
    """Fresh logical (N,C,H,W) tensor with NHWC storage."""
    return torch.empty((N, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------ conv
class ConvCfg:
    def __init__(self, weight_shape, stride, padding, dilation, groups):
        def one(v):
            if isinstance(v, (tuple, list)):
                if len(set(v)) != 1:
                    raise NotImplementedError("snws convolutions take equal stride/padding/dilation on both axes")
                return int(v[0])
            return int(v)
        self.O, self.Ig, self.kh, self.kw = [int(v) for v in weight_shape]
        self.stride, self.pad, self.dil, self.groups = one(stride), one(padding), one(dilation), int(groups)
        if isinstance(padding, str):
            raise NotImplementedError("string padding modes are not supported")
        self.I = self.Ig * self.groups

    def mfma_shape_ok(self):
        k3 = self.kh == 3 and self.kw == 3 and self.pad == 1
        k1 = self.kh == 1 and self.kw == 1 and self.pad == 0
        return (k3 or k1) and self.stride == 1 and self.dil == 1 and self.groups == 1 and self.O % 8 == 0

    def desc(self, N, H, W, c_in, ldx, ldy):
        return L.ConvDesc(N, H, W, c_in, self.O, self.kh, self.kw, self.stride, self.pad, self.dil, self.groups,
                          ldx, ldy)

    def out_hw(self, H, W):
        lib = L.lib()
        return (lib.snws_conv_out_size(H, self.kh, self.stride, self.pad, self.dil),
                lib.snws_conv_out_size(W, self.kw, self.stride, self.pad, self.dil))


class PreparedWeight:
    """Standardized filter in the layouts the kernels read, valid for one version of the master weight."""

    def __init__(self, weight, cfg, want_mfma, standardize=True, reuse=None):
        """reuse: the PreparedWeight of the previous weight version, whose buffers are overwritten in place (only a
        driver that owns the step structure, i.e. one that switched the module's cache_filters on, may pass it: no graph may still need the old images)."""
        require_gpu(weight, "Conv2dws")
        self.standardize = standardize
        w = weight.detach()
        if w.dtype != torch.float32:
            w = w.float()
        w = w.contiguous()
        dev = w.device
        self.cfg = cfg
        self.pack_fwd = self.pack_dgrad = None
        if reuse is not None and reuse.ohwi.device == dev and tuple(reuse.ohwi.shape) == (cfg.O, cfg.kh, cfg.kw, cfg.Ig) and \
                (reuse.pack_fwd is not None) == bool(want_mfma):
            self.ohwi, self.pack_fwd, self.pack_dgrad = reuse.ohwi, reuse.pack_fwd, reuse.pack_dgrad   # padding stays zero
        else:
            self.ohwi = torch.empty((cfg.O, cfg.kh, cfg.kw, cfg.Ig), dtype=torch.float32, device=dev)
            reuse = None
        if want_mfma and reuse is None:
            nbytes = L.lib().snws_conv_pack_bytes(cfg.I, cfg.O, cfg.kh, cfg.kw)
            self.pack_fwd = torch.zeros(nbytes // 2, dtype=torch.bfloat16, device=dev)
            nbytes = L.lib().snws_conv_pack_bytes(cfg.O, cfg.I, cfg.kh, cfg.kw)
            self.pack_dgrad = torch.zeros(nbytes // 2, dtype=torch.bfloat16, device=dev)
        self._flipped = None
        if standardize:
            L.call("snws_ws_fwd", _ptr(w), cfg.O, cfg.Ig, cfg.kh, cfg.kw, None, _ptr(self.ohwi), _ptr(self.pack_fwd),
                   _ptr(self.pack_dgrad), _stream())
        else:
            L.call("snws_conv_pack", _ptr(w), cfg.O, cfg.Ig, cfg.kh, cfg.kw, _ptr(self.ohwi), _ptr(self.pack_fwd),
                   _ptr(self.pack_dgrad), _stream())

    def flipped(self):
        """The float32 filter of the dgrad convolution, (I, kh, kw, O) with both spatial axes reversed (built on first
        use for this weight version; the float32 MFMA tier reads it)."""
        if self._flipped is None:
            cfg = self.cfg
            self._flipped = torch.empty((cfg.Ig, cfg.kh, cfg.kw, cfg.O), dtype=torch.float32, device=self.ohwi.device)
            L.call("snws_conv_f32_flip", _ptr(self.ohwi), _ptr(self._flipped), cfg.O, cfg.Ig, cfg.kh, cfg.kw, _stream())
        return self._flipped


def _pad_channels(x, ld, c_to):
    """NHWC tensor with its channel count zero-padded to c_to (first-layer inputs: 14 -> 16)."""
    N, C, H, W = x.shape
    out = torch.zeros((N, H, W, c_to), dtype=x.dtype, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out.permute(0, 3, 1, 2), c_to


def planar_to_padded_nhwc(x, dtype):
    """Contiguous NCHW float32 / bfloat16 batch -> logical (N,C,H,W) tensor of `dtype` stored pixel-major with its channels
    zero-padded to a multiple of 16 (snws_nchw_to_nhwc_pad: one pass); the padded tensor rides along as `_snws_padded`."""
    require_gpu(x, "input layout")
    N, C, H, W = x.shape
    c_pad = (C + 15) // 16 * 16
    buf = torch.empty((N, H, W, c_pad), dtype=dtype, device=x.device)
    _launch("input_layout", float(x.element_size() * C + buf.element_size() * c_pad) * N * H * W, "snws_nchw_to_nhwc_pad",
            _ptr(x), dtype_code(x), _ptr(buf), dtype_code(buf), N, C, H, W, c_pad, _stream())
    full = buf.permute(0, 3, 1, 2)
    y = full[:, :C]
    y._snws_padded = full
    return y


def conv_forward(x, prep, bias, want_stats=False, force_direct=False):
    """y = conv(x, standardized weight) + bias.  Returns (y, stat_part or None, saved) where `saved` is what
    conv_backward needs."""
    cfg = prep.cfg
    require_gpu(x, "Conv2dws")
    N, C, H, W = x.shape
    if C != cfg.I:
        raise RuntimeError("Conv2dws: expected %d input channels, got %d" % (cfg.I, C))
    dt = dtype_code(x)
    xs, ldx = as_nhwc(x)
    OH, OW = cfg.out_hw(H, W)
    use_mfma = (not force_direct) and prep.pack_fwd is not None and x.dtype == torch.bfloat16 and cfg.mfma_shape_ok()
    b32 = None if bias is None else (bias.detach() if bias.dtype == torch.float32 else bias.detach().float())
    stat = None
    if use_mfma:
        c_run = (C + 15) // 16 * 16
        if c_run != C or ldx % 8 or xs.data_ptr() % 16:
            padded = getattr(x, "_snws_padded", None)
            if padded is not None and padded.shape[1] == c_run and ldx == c_run and padded.dtype == x.dtype and \
                    padded.data_ptr() == xs.data_ptr() and xs.data_ptr() % 16 == 0:
                xs = padded       # the input layout kernel already left the padding channels zero
            else:
                xs, ldx = _pad_channels(xs, ldx, c_run)
        y = new_nhwc(N, cfg.O, OH, OW, x.dtype, x.device)
        if want_stats:
            T = L.lib().snws_conv_mfma_tiles(H, W)
            stat = torch.empty((N, T, cfg.O, 2), dtype=torch.float32, device=x.device)
        d = cfg.desc(N, H, W, c_run, ldx, cfg.O)
        try:
            flops = 2.0 * N * OH * OW * cfg.O * cfg.I * cfg.kh * cfg.kw
            _launch("conv_mfma_fwd_k%d_kc%d" % (cfg.kh, 32 if c_run % 32 == 0 else 16), flops, "snws_conv_mfma_fwd",
                    ctypes.byref(d), _ptr(xs), _ptr(prep.pack_fwd), _ptr(b32), _ptr(y), _ptr(stat), _stream())
        except L.SnwsUnsupported:
            use_mfma, stat = False, None
    if not use_mfma and F32_MFMA and not force_direct and x.dtype == torch.float32 and cfg.mfma_shape_ok():
        # float32 tier on the f32 matrix instruction (exact f32 arithmetic, any channel counts)
        y = new_nhwc(N, cfg.O, OH, OW, x.dtype, x.device)
        if want_stats:
            stat = torch.empty((N, L.lib().snws_conv_f32_tiles(H, W), cfg.O, 2), dtype=torch.float32, device=x.device)
        d = cfg.desc(N, H, W, C, ldx, cfg.O)
        flops = 2.0 * N * OH * OW * cfg.O * cfg.I * cfg.kh * cfg.kw
        _launch("conv_f32_fwd_k%d" % cfg.kh, flops, "snws_conv_f32_fwd", ctypes.byref(d), _ptr(xs), _ptr(prep.ohwi), _ptr(b32),
                _ptr(y), _ptr(stat), _stream())
        use_mfma = "f32"
    if not use_mfma:
        y = new_nhwc(N, cfg.O, OH, OW, x.dtype, x.device)
        d = cfg.desc(N, H, W, C, ldx, cfg.O)
        L.call("snws_conv_direct_fwd", ctypes.byref(d), dt, _ptr(xs), _ptr(prep.ohwi), _ptr(b32), _ptr(y), _stream())
    saved = (xs, ldx, (N, C, H, W), use_mfma)
    return y, stat, saved


# Weight gradients through per-workgroup partial tiles and a fixed-order reduction (snws_conv_mfma_wgrad_det) instead
# of float atomics: two runs of the same step give bit-identical gradients.  SNWS_WGRAD_ATOMIC=1 selects the atomic
# flush (A/B measurements).
WGRAD_DETERMINISTIC = os.environ.get("SNWS_WGRAD_ATOMIC", "0") != "1"


# float32 activations run the f32 matrix-instruction kernels (snws_conv_f32_*) for stride-1 3x3 / 1x1 convolutions;
# SNWS_F32_MFMA=0 sends them to the direct (scalar FMA) kernels instead (A/B measurements).
F32_MFMA = os.environ.get("SNWS_F32_MFMA", "1") != "0"


# SwitchNorm-backward fusion.  A fused block tags its output with the SwitchNorm state it saved for backward
# (tag_sn_output / assemble); a convolution that consumes a tagged tensor remembers the tags, and its dgrad launch
# then also produces pass 1 of those layers' backward (the per-tile sums of snws_sn_bwd_reduce) while it stores the
# gradient.  The layer uses them only if the gradient it finally receives is that very buffer, unmodified - i.e. the
# convolution was its only consumer; otherwise it runs its own reduce pass.
FUSE_SN_BWD = True
FUSED_SN_BWD_HITS = 0      # layers whose backward reused a consumer's partial sums (diagnostics / tests)


def tag_sn_output(y):
    """Mark y (the output of a fused conv+SwitchNorm block) as 'channels [0, C) come from the SwitchNorm whose
    backward state is y.grad_fn.sn'."""
    fn = y.grad_fn
    s = getattr(fn, "sn", None) if fn is not None else None
    if isinstance(s, dict) and s.get("y") is None:
        y._snws_seg = [(0, y.shape[1], s)]
    return y


def input_segments(x):
    """The SwitchNorm tags of a convolution input, when they qualify for the fused dgrad epilogue: every tagged layer
    starts at a multiple of 64 channels, spans a multiple of 8, and stores bf16."""
    segs = getattr(x, "_snws_seg", None) if FUSE_SN_BWD else None
    if not segs or x.dtype != torch.bfloat16 or x.shape[1] > 1024:
        return None
    for c0, c, s in segs:
        if c0 % 64 or c % 8 or s["xs"].dtype != torch.bfloat16 or s["ldx"] % 8 or s["xs"].data_ptr() % 16:
            return None
    return segs


def _launch_dgrad(cfg, d, dys, prep, dxp, segs, N, H, W, flops):
    kc = 32 if cfg.O % 32 == 0 else 16
    tag = "conv_mfma_dgrad_k%d_kc%d" % (cfg.kh, kc)
    if not segs:
        _launch(tag, flops, "snws_conv_mfma_dgrad", ctypes.byref(d), _ptr(dys), _ptr(prep.pack_dgrad), _ptr(dxp), _stream())
        return
    T = L.lib().snws_conv_mfma_tiles(H, W)
    arr = (L.SnBwdSeg * len(segs))()
    parts = []
    for i, (c0, c, s) in enumerate(segs):
        part = torch.empty((N, T, c, 2), dtype=torch.float32, device=dxp.device)
        parts.append(part)
        arr[i] = L.SnBwdSeg(c0, c, s["xs"].data_ptr(), s["ldx"], s["ctx"].data_ptr(), s["beta"].data_ptr(),
                            part.data_ptr(), int(s["relu"]))
    _launch("conv_mfma_dgradsn_k%d_kc%d" % (cfg.kh, kc), flops, "snws_conv_mfma_dgrad_sn", ctypes.byref(d), _ptr(dys), _ptr(prep.pack_dgrad), _ptr(dxp), arr,
            len(segs), _stream())
    for (c0, c, s), part in zip(segs, parts):
        # (partials, tiles, the gradient buffer - kept alive so its address cannot be recycled -, its version, channel)
        s["fused_bwd"] = (part, T, dxp, dxp._version, c0)


def fused_bwd_partials(s, dys, lddy):
    """The partial sums a consumer's dgrad epilogue left for this SwitchNorm layer, if `dys` is exactly the gradient
    that launch wrote (same memory, never modified since); else None."""
    fb = s.pop("fused_bwd", None)
    if fb is None:
        return None
    part, T, dxp, version, c0 = fb
    if dys.dtype != dxp.dtype or dys.data_ptr() != dxp.data_ptr() + c0 * dxp.element_size() or \
            lddy != dxp.shape[1] or dys._version != version or dxp._version != version or \
            tuple(dys.shape[2:]) != tuple(dxp.shape[2:]) or dys.shape[0] != dxp.shape[0] or dys.shape[1] != part.shape[2]:
        return None
    global FUSED_SN_BWD_HITS
    FUSED_SN_BWD_HITS += 1
    return part, T


# Gradient sharing for tensors with several consumers (the encoder maps, their pooled copies and the tap maps of
# the UNet3+ full-scale skips, modeling/UNet3p_SNws.py:127-156).  Autograd would sum the consumers' gradients with
# one out-of-place add per extra consumer.  share_grad(x) instead hands the consumers a tensor that carries a
# GradSlot: the first consumer to run backward returns its gradient tensor as usual and parks it in the slot; every
# later consumer adds its share INTO that tensor - in its own kernel's epilogue where the kernel can
# (snws_conv_mfma_dgrad_acc, snws_resample_bwd with accumulate), else with snws_add_inplace - and returns None.  The
# engine runs the producer's backward only after all consumers have reported, so it sees the finished sum.  Every
# consumer of a shared tensor must be one of the slot-aware ops below: the identity node share_grad inserts checks that
# the gradient arriving at the producer IS the parked tensor and raises otherwise.  A shared tensor carries no
# SwitchNorm tags: a consumer's fused dgrad epilogue would reduce one share of the gradient, not the sum.
GRAD_SHARE = os.environ.get("SNWS_GRAD_SHARE", "1") != "0"
GRAD_SHARE_HITS = 0     # gradients added into a parked tensor so far (tests read it)


class GradSlot:
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class _GradJoinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        buf, ctx.slot.buf = ctx.slot.buf, None
        if buf is not None and (g.data_ptr() != buf.data_ptr() or g.shape != buf.shape or g.dtype != buf.dtype):
            raise RuntimeError("share_grad: a consumer that does not know about shared gradients contributed to this tensor "
                               "(the gradient that arrived is not the one the slot-aware consumers accumulated into)")
        return g, None


def share_grad(x):
    """Mark x - about to be consumed by several libsnws ops - for in-place gradient accumulation (see above)."""
    if not (GRAD_SHARE and torch.is_grad_enabled() and x.requires_grad):
        return x
    if grad_slot(x) is not None:        # already shared: its consumers keep accumulating into the one slot
        return x
    slot = GradSlot()
    y = _GradJoinFn.apply(x, slot)
    y._snws_gslot = slot
    return y


def grad_slot(x):
    return getattr(x, "_snws_gslot", None)


def add_inplace(dst, src):
    """dst += src for logical (N, C, H, W) tensors with NHWC storage."""
    d, ldd = as_nhwc(dst)
    if d.data_ptr() != dst.data_ptr():      # not NHWC storage (as_nhwc made a copy): the stock in-place add
        dst.add_(src.to(dst.dtype))
        return
    if src.dtype != dst.dtype:
        src = src.to(dst.dtype)
    s_, lds = as_nhwc(src)
    N, C, H, W = dst.shape
    _launch("grad_add", 3.0 * dst.element_size() * N * C * H * W, "snws_add_inplace", _ptr(d), ldd, _ptr(s_), lds, dtype_code(d),
            N * H * W, C, _stream())


def deliver_grad(slot, dx):
    """What a slot-aware consumer returns for its input gradient dx."""
    if slot is None or dx is None:
        return dx
    if slot.buf is None:
        slot.buf = dx
        return dx
    global GRAD_SHARE_HITS
    GRAD_SHARE_HITS += 1
    add_inplace(slot.buf, dx)
    return None


def conv_backward(dy, prep, saved, weight, need_dx, need_dw, need_db, bias_key=None, in_segs=None, gslot=None):
    """Returns (dx or None, dw (O,I/g,kh,kw) f32 or None, dbias f32 or None).  bias_key: grad_key(bias), which routes
    the bias gradient to its slot in a flat gradient buffer (the weight's is found through `weight`).  in_segs: the
    SwitchNorm tags of the convolution's input (input_segments)."""
    cfg = prep.cfg
    xs, ldx, (N, C, H, W), used_mfma = saved
    dys, lddy = as_nhwc(dy)
    dt = dtype_code(dys)
    dev = dys.device
    c_run = xs.shape[1]
    dx = dw = db = None
    if need_dx and used_mfma == "f32" and dys.dtype == torch.float32:
        dx = new_nhwc(N, C, H, W, dys.dtype, dev)
        d = cfg.desc(N, H, W, C, C, lddy)
        _launch("conv_f32_dgrad_k%d" % cfg.kh, 2.0 * N * H * W * cfg.O * cfg.I * cfg.kh * cfg.kw, "snws_conv_f32_dgrad",
                ctypes.byref(d), _ptr(dys), _ptr(prep.flipped()), _ptr(dx), _stream())
    elif need_dx:
        done = False
        if used_mfma is True and cfg.O % 16 == 0 and lddy % 8 == 0 and dys.data_ptr() % 16 == 0:
            flops = 2.0 * N * H * W * cfg.O * cfg.I * cfg.kh * cfg.kw
            acc = gslot.buf if gslot is not None else None
            if acc is not None and c_run == C and acc.dtype == dys.dtype and tuple(acc.shape) == (N, C, H, W):
                # a later contributor to a shared gradient: the dgrad epilogue adds into the parked tensor
                a_, lda = as_nhwc(acc)
                if a_.data_ptr() == acc.data_ptr() and lda % 8 == 0 and a_.data_ptr() % 16 == 0:
                    d = cfg.desc(N, H, W, c_run, lda, lddy)
                    try:
                        _launch("conv_mfma_dgrad_k%d_kc%d" % (cfg.kh, 32 if cfg.O % 32 == 0 else 16), flops,
                                "snws_conv_mfma_dgrad_acc", ctypes.byref(d), _ptr(dys), _ptr(prep.pack_dgrad), _ptr(a_), _stream())
                        done, gslot = True, None          # delivered: nothing to return
                        global GRAD_SHARE_HITS
                        GRAD_SHARE_HITS += 1
                    except L.SnwsUnsupported:
                        pass
            if not done:
                dxp = new_nhwc(N, c_run, H, W, dys.dtype, dev)
                d = cfg.desc(N, H, W, c_run, c_run, lddy)
                try:
                    _launch_dgrad(cfg, d, dys, prep, dxp, in_segs, N, H, W, flops)
                    dx = dxp[:, :C] if c_run != C else dxp
                    done = True
                except L.SnwsUnsupported:
                    pass
        if not done:
            dx = new_nhwc(N, C, H, W, dys.dtype, dev)
            d = cfg.desc(N, H, W, C, C, lddy)
            L.call("snws_conv_direct_dgrad", ctypes.byref(d), dt, _ptr(dys), _ptr(prep.ohwi), _ptr(dx), _stream())
    dx = deliver_grad(gslot, dx)
    if need_dw or need_db:
        dw, db = _conv_backward_weights(dys, lddy, prep, saved, weight, need_dw, need_db, bias_key)
    return dx, dw, db


def _conv_backward_weights(dys, lddy, prep, saved, weight, need_dw, need_db, bias_key):
    cfg = prep.cfg
    xs, ldx, (N, C, H, W), used_mfma = saved
    dt, dev, c_run = dtype_code(dys), dys.device, xs.shape[1]
    dw = db = None
    done = False
    if need_db:
        db = grad_out(bias_key, (cfg.O,), dev, zero=True, acc_ok=True)    # the wgrad kernels add to it
    if used_mfma == "f32" and dys.dtype == torch.float32:
        d = cfg.desc(N, H, W, C, ldx, lddy)
        ws_bytes = L.lib().snws_conv_f32_wgrad_workspace_bytes(ctypes.byref(d))
        g = torch.empty((cfg.O, cfg.kh, cfg.kw, cfg.Ig), dtype=torch.float32, device=dev)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
        _launch("conv_f32_wgrad_k%d" % cfg.kh, 2.0 * N * H * W * cfg.O * cfg.I * cfg.kh * cfg.kw, "snws_conv_f32_wgrad",
                ctypes.byref(d), _ptr(xs), _ptr(dys), _ptr(g), _ptr(db), _ptr(ws), ws_bytes, _stream())
        ldg, done = cfg.Ig, True
    elif used_mfma is True and lddy % 8 == 0 and dys.data_ptr() % 16 == 0:
        d = cfg.desc(N, H, W, c_run, ldx, lddy)
        flops = 2.0 * N * H * W * cfg.O * cfg.I * cfg.kh * cfg.kw
        ws_bytes = L.lib().snws_conv_mfma_wgrad_workspace_bytes(ctypes.byref(d)) if WGRAD_DETERMINISTIC else -1
        try:
            if ws_bytes > 0:
                # per-workgroup partial tiles + a fixed-order reduction: bit-reproducible, and g needs no zero-fill
                g = torch.empty((cfg.O, cfg.kh, cfg.kw, c_run), dtype=torch.float32, device=dev)
                ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
                _launch("conv_mfma_wgrad_k%d" % cfg.kh, flops, "snws_conv_mfma_wgrad_det", ctypes.byref(d), _ptr(xs),
                        _ptr(dys), _ptr(g), _ptr(db), _ptr(ws), ws_bytes, _stream())
            else:
                g = torch.zeros((cfg.O, cfg.kh, cfg.kw, c_run), dtype=torch.float32, device=dev)
                _launch("conv_mfma_wgrad_k%d" % cfg.kh, flops, "snws_conv_mfma_wgrad", ctypes.byref(d), _ptr(xs),
                        _ptr(dys), _ptr(g), _ptr(db), _stream())
            ldg, done = c_run, True
        except L.SnwsUnsupported:
            pass
    if not done:
        g = torch.empty((cfg.O, cfg.kh, cfg.kw, cfg.Ig), dtype=torch.float32, device=dev)
        d = cfg.desc(N, H, W, C, ldx, lddy)
        if db is not None and grad_accumulates(db):     # the direct kernel overwrites its bias gradient
            db_once = torch.empty_like(db)
            L.call("snws_conv_direct_wgrad", ctypes.byref(d), dt, _ptr(xs), _ptr(dys), _ptr(g), _ptr(db_once), _stream())
            db.add_(db_once)
        else:
            L.call("snws_conv_direct_wgrad", ctypes.byref(d), dt, _ptr(xs), _ptr(dys), _ptr(g), _ptr(db), _stream())
        ldg = cfg.Ig
    if need_dw and not prep.standardize:
        dw = g[..., :cfg.Ig].permute(0, 3, 1, 2).contiguous()
    elif need_dw:
        w32 = weight.detach()
        w32 = (w32 if w32.dtype == torch.float32 else w32.float()).contiguous()
        dw = grad_out(grad_key(weight), (cfg.O, cfg.Ig, cfg.kh, cfg.kw), dev, acc_ok=True)
        L.call("snws_ws_bwd", _ptr(w32), _ptr(g), ldg, _ptr(dw), cfg.O, cfg.Ig, cfg.kh, cfg.kw,
               int(grad_accumulates(dw)), _stream())
    return grad_fold(dw), grad_fold(db)


# ------------------------------------------------------------------------------------------------ SwitchNorm
_REDUCE_OPS = {"sum": "SUM", "max": "MAX"}


def all_reduce(t, op="sum", group=None, async_op=False):
    """In-place all-reduce of t over `group`.  RCCL takes the device tensor directly; gloo - only the CPU / shared-GPU
    test vehicle - gets device tensors staged through the host explicitly (its own device-tensor path stalls when
    ranks share one GPU) and is then always synchronous.  Returns the work handle of an asynchronous call, else None."""
    import torch.distributed as dist
    rop = getattr(dist.ReduceOp, _REDUCE_OPS[op])
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=rop, group=group)
        t.copy_(h)
        return None
    w = dist.all_reduce(t, op=rop, group=group, async_op=async_op)
    return w if async_op else None


def broadcast(t, src=0, group=None):
    """In-place broadcast with the same gloo staging rule as all_reduce."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src, group=group)


# The SwitchNorm / BatchNorm statistic all-reduces (a few KB each, one per normalisation call and direction) get a
# process group of their own: under RCCL that is a separate communicator with its own stream, so a statistic
# all-reduce issued in backward never queues behind a 25 MB gradient bucket that is in flight on the default group.
# SNWS_STAT_GROUP=0 keeps them on the default group.
_STAT_GROUP = {}


def stat_group(group=None):
    """The process group the statistic all-reduces of `group` (None = the default group) run on.  Created on first
    use; every rank reaches its first normalisation call of a step at the same point, so the collective creation
    call is matched on all ranks."""
    import torch.distributed as dist
    if group is not None or os.environ.get("SNWS_STAT_GROUP", "1") == "0":
        return group
    key = id(dist.group.WORLD)
    g = _STAT_GROUP.get(key)
    if g is None:
        g = _STAT_GROUP[key] = dist.new_group(backend=dist.get_backend())
    return g


def _allreduce_sum(t, group):
    trace_collective("sn", t.numel())
    all_reduce(t, "sum", stat_group(group))


def trace_collective(kind, numel):
    """SNWS_TRACE_COLLECTIVES=<dir>: append every collective this rank issues to <dir>/coll_rank<r>.log (debug aid
    for order mismatches between ranks)."""
    import os
    d = os.environ.get("SNWS_TRACE_COLLECTIVES")
    if d:
        with open(os.path.join(d, "coll_rank%s.log" % os.environ.get("RANK", "0")), "a") as f:
            f.write("%s %d\n" % (kind, numel))


def dp_collectives(group=None):
    """True when the data-parallel collectives have to be issued: a process group of more than one rank - or of one
    rank when SNWS_DP_SELFTEST=1, which lets a single-GPU box drive every RCCL call of the multi-GPU path."""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("SNWS_DP_SELFTEST") == "1"


def sync_world(group=None):
    """Number of data-parallel ranks whose BN-branch statistics are summed (1 when not distributed)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


# Independent SwitchNorm calls whose inputs are all available before any of their outputs is needed - the five
# branch blocks of a UNet3+ decoder scale (modeling/UNet3p_SNws.py:127-156) - are collected inside
# `with stat_batch(group):` and run together when the block ends: their per-(image, channel) passes and their
# normalise passes go out as ONE batched launch per pass (the small passes are latency-bound: five cost what one
# does), and in data-parallel runs their [2C+1] payloads travel in ONE statistics all-reduce.  Outputs of calls made
# inside the block must not be read before it ends.  A module used several times inside one block (the shared
# convu_cat block on up to three sources) keeps its call order: its k-th use runs in the k-th round of launches, so
# running statistics are updated in the order the reference updates them.  SNWS_SN_BATCH=0: single launches.
STAT_BATCH = None
SN_BATCHING = os.environ.get("SNWS_SN_BATCH", "1") != "0"


def _rounds(calls, key):
    """Splits `calls` into rounds such that no round holds two calls of the same key; the k-th call of a key lands in
    round k, so calls of one key keep their order across rounds."""
    seen, rounds = {}, []
    for c in calls:
        r = seen.get(key(c), 0)
        seen[key(c)] = r + 1
        while len(rounds) <= r:
            rounds.append([])
        rounds[r].append(c)
    return rounds


def _groups(calls, sig):
    """The calls of one round grouped by launch signature, at most L.SN_BATCH_MAX per group, order kept."""
    out = {}
    for c in calls:
        g = out.setdefault(sig(c), [[]])
        if len(g[-1]) == L.SN_BATCH_MAX:
            g.append([])
        g[-1].append(c)
    return [g for gs in out.values() for g in gs]


def _addr(t):
    return None if t is None else t.data_ptr()


class _SnCall:
    """One deferred SwitchNorm forward of a StatBatch."""
    __slots__ = ("part", "T", "shift", "per_image", "ctx", "pack", "gamma", "beta", "mw", "vw", "rm", "rv", "xs", "ldx",
                 "y", "ldy", "N", "HW", "C", "dt", "relu", "flags", "esize")

    @property
    def module_key(self):
        return _addr(self.rm) if self.rm is not None else _addr(self.gamma)

    @property
    def sig(self):
        return (self.N, self.HW, self.C, self.dt, self.relu, self.flags)

    def item(self, it):
        it.part, it.shift, it.ctx, it.pack = _addr(self.part), _addr(self.shift), _addr(self.ctx), _addr(self.pack)
        it.gamma, it.beta, it.mean_weight, it.var_weight = _addr(self.gamma), _addr(self.beta), _addr(self.mw), _addr(self.vw)
        it.running_mean, it.running_var = _addr(self.rm), _addr(self.rv)
        it.x, it.y, it.ldx, it.ldy = _addr(self.xs), _addr(self.y), self.ldx, self.ldy
        it.T, it.shift_per_image = self.T, self.per_image


def _sn_fwd_a(grp):
    c0, st = grp[0], _stream()
    if len(grp) > 1 and SN_BATCHING:
        items = (L.SnFwdItem * len(grp))()
        for c, it in zip(grp, items):
            c.item(it)
        L.call("snws_sn_fwd_a_batched", items, len(grp), c0.N, c0.HW, c0.C, st)
        return
    for c in grp:
        L.call("snws_sn_finalize", _ptr(c.part), c.T, _ptr(c.shift), c.per_image, c.N, c.HW, c.C, _ptr(c.ctx), st)
        L.call("snws_sn_mix_a", _ptr(c.ctx), _ptr(c.pack), c.N, c.C, st)


def _sn_fwd_b(grp):
    c0, st = grp[0], _stream()
    eps, momentum, training, movavg, using_bn = c0.flags
    work = sum(2.0 * c.esize * c.N * c.HW * c.C for c in grp)
    if len(grp) > 1 and SN_BATCHING:
        items = (L.SnFwdItem * len(grp))()
        for c, it in zip(grp, items):
            c.item(it)
        L.call("snws_sn_mix_b_batched", items, len(grp), c0.N, c0.C, eps, momentum, training, movavg, using_bn, st)
        try:
            _launch("sn_apply", work, "snws_sn_apply_batched", items, len(grp), c0.dt, c0.N, c0.HW, c0.C, int(c0.relu), st)
            return
        except L.SnwsUnsupported:
            pass
    else:
        for c in grp:
            L.call("snws_sn_mix_b", _ptr(c.ctx), _ptr(c.pack), _ptr(c.gamma), _ptr(c.mw), _ptr(c.vw), _ptr(c.rm), _ptr(c.rv),
                   c.N, c.C, eps, momentum, training, movavg, using_bn, st)
    for c in grp:
        _launch("sn_apply", 2.0 * c.esize * c.N * c.HW * c.C, "snws_sn_apply", _ptr(c.xs), c.ldx, _ptr(c.y), c.ldy, None, 0,
                c.dt, _ptr(c.ctx), _ptr(c.beta), c.N, c.HW, c.C, int(c.relu), st)


class StatBatch:
    def __init__(self, group=None, capacity=16384):
        self.group, self.capacity = group, capacity
        self.buf, self.used, self.pending = None, 0, []

    def slot(self, n, device):
        if self.buf is None:
            self.buf = torch.empty(self.capacity, dtype=torch.float32, device=device)
        if self.used + n > self.capacity:
            raise RuntimeError("StatBatch: capacity %d exceeded" % self.capacity)
        t = self.buf[self.used:self.used + n]
        self.used += n
        return t

    def flush(self):
        calls, self.pending = self.pending, []
        launches = [g for rnd in _rounds(calls, lambda c: c.module_key) for g in _groups(rnd, lambda c: c.sig)]
        for g in launches:
            _sn_fwd_a(g)
        if self.used:
            _allreduce_sum(self.buf[:self.used], self.group)
        for g in launches:      # round by round: a module's later use finds the running statistics of its earlier one
            _sn_fwd_b(g)
        self.buf, self.used = None, 0

    def __enter__(self):
        global STAT_BATCH
        if STAT_BATCH is not None:
            raise RuntimeError("stat_batch blocks do not nest")
        STAT_BATCH = self
        return self

    def __exit__(self, exc_type, exc, tb):
        global STAT_BATCH
        STAT_BATCH = None
        if exc_type is None:
            self.flush()
        else:
            self.pending = []
        return False


class _NoBatch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def stat_batch(group=None):
    """Context manager: a StatBatch when there is something to share - the statistics all-reduce of a data-parallel
    run, or the batched launches - else a no-op."""
    return StatBatch(group) if (SN_BATCHING or dp_collectives(group)) else _NoBatch()


class SNParams:
    """Plain bundle of one SwitchNorm2d's tensors and flags (module attributes, modeling/snorm.py:6-25)."""

    def __init__(self, mod):
        self.gamma, self.beta = mod.weight, mod.bias
        self.mean_weight, self.var_weight = mod.mean_weight, mod.var_weight
        self.using_bn, self.using_movavg = bool(mod.using_bn), bool(mod.using_movavg)
        self.running_mean = mod.running_mean if self.using_bn else None
        self.running_var = mod.running_var if self.using_bn else None
        self.eps, self.momentum, self.training = float(mod.eps), float(mod.momentum), bool(mod.training)
        self.sync = bool(getattr(mod, "sync_bn_stats", True))
        self.group = getattr(mod, "process_group", None)


def _f32c(t):
    t = t.detach()
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


def sn_forward(x, p, relu, stat_part=None, stat_shift=None, residual=None, out=None, apply=True):
    """SwitchNorm (+ReLU) of the logical (N,C,H,W) tensor x.  stat_part/stat_shift: partial moments from the
    conv epilogue (then x is not re-read for statistics).  Returns (y, saved).  apply=False stops after the
    coefficients (saved["ctx"]): the caller runs its own normalise pass (the fused classification head); y is None."""
    require_gpu(x, "SwitchNorm2d")
    N, C, H, W = x.shape
    HW = H * W
    dt = dtype_code(x)
    dev = x.device
    lib = L.lib()
    xs, ldx = as_nhwc(x)
    st = _stream()
    ctx = torch.empty(lib.snws_sn_ctx_floats(N, C), dtype=torch.float32, device=dev)
    if stat_part is None:
        T = lib.snws_sn_stats_tiles(N, HW, C)
        stat_part = torch.empty((N, T, C, 2), dtype=torch.float32, device=dev)
        stat_shift = torch.empty((N, C), dtype=torch.float32, device=dev)
        _launch("sn_stats", float(x.element_size()) * N * HW * C, "snws_sn_stats", _ptr(xs), dt, N, HW, C, ldx,
                _ptr(stat_part), _ptr(stat_shift), st)
        per_image = 1
    else:
        T = stat_part.shape[1]
        per_image = 0
    synced = p.sync and p.training and p.using_bn and dp_collectives(p.group)
    world = sync_world(p.group) if synced else 1
    batch = STAT_BATCH if (apply and residual is None and STAT_BATCH is not None and STAT_BATCH.group is p.group) else None
    pack = batch.slot(2 * C + 1, dev) if (batch is not None and synced) else \
        torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
    if batch is None:
        L.call("snws_sn_finalize", _ptr(stat_part), T, _ptr(stat_shift), per_image, N, HW, C, _ptr(ctx), st)
        L.call("snws_sn_mix_a", _ptr(ctx), _ptr(pack), N, C, st)
        if synced:
            _allreduce_sum(pack, p.group)
    gamma, beta = _f32c(p.gamma).view(-1), _f32c(p.beta).view(-1)
    mw, vw = _f32c(p.mean_weight), _f32c(p.var_weight)
    rm = p.running_mean.view(-1) if p.using_bn else None
    rv = p.running_var.view(-1) if p.using_bn else None
    if p.using_bn and (rm.dtype != torch.float32 or not rm.is_contiguous() or rv.dtype != torch.float32):
        raise TypeError("SwitchNorm2d running buffers must be contiguous float32")
    if not apply:
        y, ldy = None, 0
    elif out is None:
        y, ldy = new_nhwc(N, C, H, W, x.dtype, dev), C
    else:       # caller-provided destination, e.g. a channel slice of a concat buffer
        if tuple(out.shape) != (N, C, H, W) or out.dtype != x.dtype:
            raise ValueError("SwitchNorm2d: out has shape %s / dtype %s, expected %s / %s"
                             % (tuple(out.shape), out.dtype, (N, C, H, W), x.dtype))
        y, ldy = as_nhwc(out)
        if y.data_ptr() != out.data_ptr():
            raise ValueError("SwitchNorm2d: out must be channels-last (or a channel slice of a channels-last tensor)")
    rs, ldr = (None, 0)
    if residual is not None:
        rs, ldr = as_nhwc(residual.to(x.dtype))
    momentum, eps, training, movavg, using_bn = p.momentum, p.eps, int(p.training), int(p.using_movavg), int(p.using_bn)

    def finish():       # everything after the (possibly shared) statistics all-reduce
        L.call("snws_sn_mix_b", _ptr(ctx), _ptr(pack), _ptr(gamma), _ptr(mw), _ptr(vw), _ptr(rm), _ptr(rv), N, C,
               eps, momentum, training, movavg, using_bn, st)
        if apply:
            _launch("sn_apply", 2.0 * x.element_size() * N * HW * C, "snws_sn_apply", _ptr(xs), ldx, _ptr(y), ldy, _ptr(rs),
                    ldr, dt, _ptr(ctx), _ptr(beta), N, HW, C, int(relu), st)
    if batch is not None:       # every pass after the statistics runs when the batch is flushed; y is filled then
        c = _SnCall()
        c.part, c.T, c.shift, c.per_image, c.ctx, c.pack = stat_part, T, stat_shift, per_image, ctx, pack
        c.gamma, c.beta, c.mw, c.vw, c.rm, c.rv = gamma, beta, mw, vw, rm, rv
        c.xs, c.ldx, c.y, c.ldy, c.N, c.HW, c.C, c.dt, c.relu = xs, ldx, y, ldy, N, HW, C, dt, bool(relu)
        c.flags, c.esize = (eps, momentum, training, movavg, using_bn), x.element_size()
        batch.pending.append(c)
    else:
        finish()
    n_global = float(N * world) if world > 1 else float(N)
    saved = dict(xs=xs, ldx=ldx, ctx=ctx, gamma=gamma, beta=beta, mw=mw, vw=vw, relu=bool(relu), shape=(N, C, H, W),
                 training=p.training, using_bn=p.using_bn, world=world, synced=synced, group=p.group, n_global=n_global,
                 y=(y if residual is not None and relu else None))
    return y, saved


def sn_backward_pass1(dy, s, packb=None, partials=None, launch=True):
    """First half of the SwitchNorm backward, up to the payload of the backward all-reduce: the two reductions per
    (n, c) (or the partial sums a consumer's dgrad epilogue left), d gamma / d beta, and packb = [sum_n dm, sum_n dv]
    (written into `packb` when given - a slot of a buffer shared by several layers).  Returns the state
    sn_backward_pass2 continues from; the all-reduce of packb happens between the two (sn_backward, or
    AssembleFn.backward for the layers of a concatenation).  launch=False: everything but the snws_sn_bwd_mix_a call,
    which the caller issues for several layers at once (_sn_bwd_a)."""
    N, C, H, W = s["shape"]
    HW = H * W
    lib = L.lib()
    st = _stream()
    if partials is not None:
        # (part, T) computed by the caller (the fused classification head rebuilds dy on the fly): dy may be None
        part, T = partials
        dys, lddy, yms, ldym, dev = None, 0, None, 0, part.device
    else:
        dys, lddy = as_nhwc(dy)
        if dys.dtype != s["xs"].dtype:
            dys, lddy = as_nhwc(dy.to(s["xs"].dtype))
        dt = dtype_code(dys)
        dev = dys.device
        ym = s["y"]
        yms, ldym = (None, 0) if ym is None else as_nhwc(ym)
    fused = fused_bwd_partials(s, dys, lddy) if (partials is None and ym is None) else None
    if partials is not None:
        pass
    elif fused is not None:
        part, T = fused
    else:
        T = lib.snws_sn_stats_tiles(N, HW, C)
        part = torch.empty((N, T, C, 2), dtype=torch.float32, device=dev)
        _launch("sn_bwd_reduce", 2.0 * dys.element_size() * N * HW * C, "snws_sn_bwd_reduce", _ptr(dys), lddy,
                _ptr(s["xs"]), s["ldx"], _ptr(yms), ldym, dt, _ptr(s["ctx"]), _ptr(s["gamma"]), _ptr(s["beta"]), N, HW, C,
                int(s["relu"]), _ptr(part), st)
    ctxb = torch.empty(lib.snws_sn_ctxb_floats(N, C), dtype=torch.float32, device=dev)
    if packb is None:
        packb = torch.empty(2 * C, dtype=torch.float32, device=dev)
    dgamma = grad_out(grad_key(s["gamma"]), (C,), dev, acc_ok=True)
    dbeta = grad_out(grad_key(s["beta"]), (C,), dev, acc_ok=True)
    acc_a = grad_accumulates(dgamma) and grad_accumulates(dbeta)
    if not acc_a and (grad_accumulates(dgamma) or grad_accumulates(dbeta)):
        raise RuntimeError("SwitchNorm weight and bias must be used together")
    if launch:
        L.call("snws_sn_bwd_mix_a", _ptr(part), T, _ptr(s["ctx"]), _ptr(s["gamma"]), _ptr(ctxb), _ptr(packb), _ptr(dgamma),
               _ptr(dbeta), N, C, int(acc_a), st)
    return dict(dys=dys, lddy=lddy, yms=yms, ldym=ldym, ctxb=ctxb, packb=packb, dgamma=dgamma, dbeta=dbeta, dev=dev,
                part=part, T=T, acc_a=int(acc_a),
                key=None if dy is None else (dy.data_ptr(), tuple(dy.shape), tuple(dy.stride()), dy._version))


def _sn_bwd_item(it, s, p1, q=None):
    """Fills one snws_sn_bwd_item from a layer's saved forward state, its pass-1 state and (second half) q."""
    it.part, it.ctx, it.gamma, it.beta = _addr(p1["part"]), _addr(s["ctx"]), _addr(s["gamma"]), _addr(s["beta"])
    it.mean_weight, it.var_weight = _addr(s["mw"]), _addr(s["vw"])
    it.ctxb, it.packb, it.dgamma, it.dbeta = _addr(p1["ctxb"]), _addr(p1["packb"]), _addr(p1["dgamma"]), _addr(p1["dbeta"])
    it.T, it.accumulate_a = p1["T"], p1["acc_a"]
    if q is not None:
        it.dmean_weight, it.dvar_weight, it.dx_sum, it.accumulate_b = _addr(q["dmw"]), _addr(q["dvw"]), _addr(q["dxsum"]), q["acc_b"]
        it.dy, it.lddy, it.x, it.ldx, it.dx, it.lddx = _addr(p1["dys"]), p1["lddy"], _addr(s["xs"]), s["ldx"], _addr(q["dx"]), s["shape"][1]


def _sn_bwd_sig(sp):
    s, p1 = sp
    return (s["shape"], dtype_code(p1["dys"]), s["relu"], s["n_global"], s["training"], s["using_bn"])


def _sn_bwd_a(grp):
    """snws_sn_bwd_mix_a for the (saved state, pass-1 state) pairs of one launch group."""
    st = _stream()
    N, C = grp[0][0]["shape"][:2]
    if len(grp) > 1:
        items = (L.SnBwdItem * len(grp))()
        for (s, p1), it in zip(grp, items):
            _sn_bwd_item(it, s, p1)
        L.call("snws_sn_bwd_a_batched", items, len(grp), N, C, st)
        return
    for s, p1 in grp:
        L.call("snws_sn_bwd_mix_a", _ptr(p1["part"]), p1["T"], _ptr(s["ctx"]), _ptr(s["gamma"]), _ptr(p1["ctxb"]),
               _ptr(p1["packb"]), _ptr(p1["dgamma"]), _ptr(p1["dbeta"]), N, C, p1["acc_a"], st)


def _sn_bwd_b(grp):
    """Second half of the backward (coefficients, d mean_weight / d var_weight, the apply pass) for one launch group;
    leaves every layer's result in its saved state as "pass2" for sn_backward to hand out."""
    st = _stream()
    s0, p0 = grp[0]
    N, C, H, W = s0["shape"]
    HW = H * W
    qs = []
    for s, p1 in grp:
        want, key = s.get("dx_sum_req", (False, None))
        q = _sn_bwd2_buffers(s, p1, want, key)
        q["dx"] = new_nhwc(N, C, H, W, p1["dys"].dtype, p1["dev"])
        qs.append(q)
    dt = dtype_code(p0["dys"])
    items = (L.SnBwdItem * len(grp))()
    for (s, p1), q, it in zip(grp, qs, items):
        _sn_bwd_item(it, s, p1, q)
    L.call("snws_sn_bwd_mix_b_batched", items, len(grp), N, HW, C, s0["n_global"], int(s0["training"]),
           int(s0["using_bn"]), st)
    work = 3.0 * p0["dys"].element_size() * N * HW * C
    try:
        _launch("sn_bwd_apply", work * len(grp), "snws_sn_bwd_apply_batched", items, len(grp), dt, N, HW, C,
                int(s0["relu"]), st)
    except L.SnwsUnsupported:
        for (s, p1), q in zip(grp, qs):
            _launch("sn_bwd_apply", work, "snws_sn_bwd_apply", _ptr(p1["dys"]), p1["lddy"], _ptr(s["xs"]), s["ldx"], None, 0,
                    _ptr(q["dx"]), C, dt, _ptr(s["ctx"]), _ptr(p1["ctxb"]), _ptr(s["beta"]), N, HW, C, int(s["relu"]), st)
    for (s, p1), q in zip(grp, qs):
        s["pass2"] = (p1["key"], _sn_bwd2_result(p1, q))


def _sn_bwd2_buffers(s, p1, want_dx_sum, dx_sum_key):
    """Where the second half of a layer's backward writes its parameter gradients (see grad_out)."""
    dev = p1["dev"]
    C = s["shape"][1]
    K = s["mw"].numel()
    dmw = grad_out(grad_key(s["mw"]), (K,), dev, acc_ok=True)
    dvw = grad_out(grad_key(s["vw"]), (K,), dev, acc_ok=True)
    dxsum = grad_out(dx_sum_key, (C,), dev, acc_ok=True) if want_dx_sum else None
    if grad_accumulates(dmw) != grad_accumulates(dvw):
        raise RuntimeError("SwitchNorm mean_weight and var_weight must be used together")
    acc_b = int(grad_accumulates(dmw)) | (2 if dxsum is not None and grad_accumulates(dxsum) else 0)
    return dict(dmw=dmw, dvw=dvw, dxsum=dxsum, acc_b=acc_b, want=want_dx_sum)


def _sn_bwd2_result(p1, q):
    res = (q["dx"], grad_fold(p1["dgamma"]), grad_fold(p1["dbeta"]), grad_fold(q["dmw"]), grad_fold(q["dvw"]))
    return res + (grad_fold(q["dxsum"]),) if q["want"] else res


def sn_backward_pass2(s, p1, want_dx_sum=False, dx_sum_key=None, dres_dtype=None, apply_fn=None):
    """Second half: coefficients from the (all-reduced) packb, d mean_weight / d var_weight, the apply pass (or
    apply_fn(ctxb) -> dx in its place)."""
    N, C, H, W = s["shape"]
    HW = H * W
    st = _stream()
    dys, lddy, yms, ldym, ctxb, packb = p1["dys"], p1["lddy"], p1["yms"], p1["ldym"], p1["ctxb"], p1["packb"]
    dev = p1["dev"]
    q = _sn_bwd2_buffers(s, p1, want_dx_sum, dx_sum_key)
    dmw, dvw, dxsum, acc_b = q["dmw"], q["dvw"], q["dxsum"], q["acc_b"]
    L.call("snws_sn_bwd_mix_b", _ptr(s["ctx"]), _ptr(ctxb), _ptr(packb), _ptr(s["mw"]), _ptr(s["vw"]), None, None,
           _ptr(dmw), _ptr(dvw), _ptr(dxsum), N, HW, C, s["n_global"], int(s["training"]), int(s["using_bn"]), acc_b, st)
    if apply_fn is not None:
        q["dx"] = apply_fn(ctxb)
        return _sn_bwd2_result(p1, q)
    dt = dtype_code(dys)
    dx = new_nhwc(N, C, H, W, dys.dtype, dev)
    if dres_dtype is not None and dres_dtype == dys.dtype and yms is not None and s["relu"]:
        dres = s["dres"] = new_nhwc(N, C, H, W, dys.dtype, dev)
        _launch("sn_bwd_apply", 4.0 * dys.element_size() * N * HW * C, "snws_sn_bwd_apply_res", _ptr(dys), lddy,
                _ptr(s["xs"]), s["ldx"], _ptr(yms), ldym, _ptr(dx), C, _ptr(dres), C, dt, _ptr(s["ctx"]), _ptr(ctxb),
                _ptr(s["beta"]), N, HW, C, st)
    else:
        _launch("sn_bwd_apply", 3.0 * dys.element_size() * N * HW * C, "snws_sn_bwd_apply", _ptr(dys), lddy,
                _ptr(s["xs"]), s["ldx"], _ptr(yms), ldym, _ptr(dx), C, dt, _ptr(s["ctx"]), _ptr(ctxb), _ptr(s["beta"]),
                N, HW, C, int(s["relu"]), st)
    q["dx"] = dx
    return _sn_bwd2_result(p1, q)


def sn_backward(dy, s, want_dx_sum=False, dx_sum_key=None, dres_dtype=None):
    """Returns (dx, dgamma (C), dbeta (C), dmean_weight, dvar_weight[, dx_sum (C)]).  dx_sum is the closed-form
    per-channel sum of dx, i.e. the gradient of a bias added just before the normalisation.  dres_dtype (residual
    variant only): also produce the gradient of the residual input, [y > 0] * dy, in the apply pass when it has the
    activation dtype; it is left in s["dres"]."""
    p2 = s.pop("pass2", None)
    if p2 is not None:
        # the whole SwitchNorm backward of this layer already ran together with its siblings (AssembleFn.backward)
        key, res = p2
        if key != (dy.data_ptr(), tuple(dy.shape), tuple(dy.stride()), dy._version) or \
                s.get("dx_sum_req", (False, None)) != (bool(want_dx_sum), dx_sum_key if want_dx_sum else None):
            raise RuntimeError("SwitchNorm backward: the gradient that arrived is not the one the grouped pre-pass "
                               "worked on (a block output inside a concatenation has a second consumer?)")
        return res
    p1 = s.pop("pass1", None)
    if p1 is not None:
        # pass 1 and the all-reduce already ran for this layer together with its siblings (AssembleFn.backward)
        if p1["key"] != (dy.data_ptr(), tuple(dy.shape), tuple(dy.stride()), dy._version):
            raise RuntimeError("SwitchNorm backward: the gradient that arrived is not the one the grouped statistics "
                               "pre-pass reduced (a block output inside a concatenation has a second consumer?)")
    else:
        p1 = sn_backward_pass1(dy, s)
        if s["synced"]:
            _allreduce_sum(p1["packb"], s["group"])
    return sn_backward_pass2(s, p1, want_dx_sum, dx_sum_key, dres_dtype)


# ------------------------------------------------------------------------------------------------ BatchNorm
def bn_forward(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, sync=False, group=None,
               clamp_eps=False, post=None):
    """nn.BatchNorm2d (optionally + ReLU) of the logical (N,C,H,W) tensor x on the SwitchNorm streaming kernels.
    sync: all-reduce the packed moments over the process group (SynchronizedBatchNorm2d semantics).  post: optional
    (N, C) float32 factors >= 0 applied to the output per (image, channel) - the feature-dropout mask of a
    Dropout2d that follows - in the same pass.  Returns (y, saved)."""
    require_gpu(x, "BatchNorm2d")
    N, C, H, W = x.shape
    HW = H * W
    dt, dev, lib, st = dtype_code(x), x.device, L.lib(), _stream()
    xs, ldx = as_nhwc(x)
    ctx = torch.empty(lib.snws_sn_ctx_floats(N, C), dtype=torch.float32, device=dev)
    use_batch = training or running_mean is None
    if use_batch and HW < 2:
        # the moment kernels work per image with the unbiased (HW - 1) divisor in between
        raise ValueError("BatchNorm2d: batch statistics need at least 2 positions per image (got %dx%d); "
                         "present single-position samples as one image of N positions" % (H, W))
    synced = bool(sync and use_batch and dp_collectives(group))
    world = sync_world(group) if synced else 1
    pack = None
    if use_batch:
        fused = getattr(x, "_snws_stat", None)
        if fused is not None and (fused["ptr"] != xs.data_ptr() or fused["version"] != x._version or
                                  fused["shape"] != (N, C, H, W) or tuple(fused["part"].shape) != (N, fused["T"], C, 2)):
            fused = None
        if fused is not None:       # the producing convolution's epilogue already summed the moments
            global BN_STAT_HITS
            BN_STAT_HITS += 1
            L.call("snws_sn_finalize", _ptr(fused["part"]), fused["T"], _ptr(fused["shift"]), 0, N, HW, C, _ptr(ctx), st)
        else:
            T = lib.snws_sn_stats_tiles(N, HW, C)
            part = torch.empty((N, T, C, 2), dtype=torch.float32, device=dev)
            shift = torch.empty((N, C), dtype=torch.float32, device=dev)
            _launch("sn_stats", float(x.element_size()) * N * HW * C, "snws_sn_stats", _ptr(xs), dt, N, HW, C, ldx,
                    _ptr(part), _ptr(shift), st)
            L.call("snws_sn_finalize", _ptr(part), T, _ptr(shift), 1, N, HW, C, _ptr(ctx), st)
        pack = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
        L.call("snws_bn_pack", _ptr(ctx), N, HW, C, _ptr(pack), st)
        if synced:
            _allreduce_sum(pack, group)
    g32 = None if gamma is None else _f32c(gamma).view(-1)
    b32 = torch.zeros(C, dtype=torch.float32, device=dev) if beta is None else _f32c(beta).view(-1)
    for buf in (running_mean, running_var):
        if buf is not None and (buf.dtype != torch.float32 or not buf.is_contiguous()):
            raise TypeError("BatchNorm2d running buffers must be contiguous float32")
    L.call("snws_bn_coeff", _ptr(ctx), _ptr(pack), _ptr(g32), _ptr(running_mean), _ptr(running_var), N, HW, C, float(eps),
           float(momentum), int(use_batch), int(bool(clamp_eps)), st)
    y = new_nhwc(N, C, H, W, x.dtype, dev)
    if post is not None:
        post = post.detach().to(torch.float32).reshape(N, C).contiguous()
        _launch("sn_apply", 2.0 * x.element_size() * N * HW * C, "snws_sn_apply_post", _ptr(xs), ldx, _ptr(y), C, dt,
                _ptr(ctx), _ptr(b32), _ptr(post), N, HW, C, int(relu), st)
    else:
        _launch("sn_apply", 2.0 * x.element_size() * N * HW * C, "snws_sn_apply", _ptr(xs), ldx, _ptr(y), C, None, 0, dt,
                _ptr(ctx), _ptr(b32), N, HW, C, int(relu), st)
    saved = dict(xs=xs, ldx=ldx, ctx=ctx, beta=b32, gamma=g32 if g32 is not None else torch.ones(C, dtype=torch.float32, device=dev),
                 relu=bool(relu), shape=(N, C, H, W), batch=use_batch, synced=synced, group=group,
                 count=float(N * world * HW), post=post)
    return y, saved


def bn_backward(dy, s):
    """Returns (dx, dgamma (C), dbeta (C)) of bn_forward."""
    N, C, H, W = s["shape"]
    HW = H * W
    lib, st = L.lib(), _stream()
    dys, lddy = as_nhwc(dy)
    if dys.dtype != s["xs"].dtype:
        dys, lddy = as_nhwc(dy.to(s["xs"].dtype))
    dt, dev = dtype_code(dys), dys.device
    T = lib.snws_sn_stats_tiles(N, HW, C)
    part = torch.empty((N, T, C, 2), dtype=torch.float32, device=dev)
    _launch("sn_bwd_reduce", 2.0 * dys.element_size() * N * HW * C, "snws_sn_bwd_reduce", _ptr(dys), lddy, _ptr(s["xs"]),
            s["ldx"], None, 0, dt, _ptr(s["ctx"]), _ptr(s["gamma"]), _ptr(s["beta"]), N, HW, C, int(s["relu"]), _ptr(part), st)
    packb = torch.empty(2 * C, dtype=torch.float32, device=dev)
    dgamma = grad_out(grad_key(s["gamma"]), (C,), dev)
    dbeta = grad_out(grad_key(s["beta"]), (C,), dev)
    post = s.get("post")
    if post is not None:
        L.call("snws_bn_bwd_pack_post", _ptr(part), T, N, C, _ptr(post), _ptr(packb), _ptr(dgamma), _ptr(dbeta), st)
    else:
        L.call("snws_bn_bwd_pack", _ptr(part), T, N, C, _ptr(packb), _ptr(dgamma), _ptr(dbeta), st)
    if s["synced"]:
        _allreduce_sum(packb, s["group"])
    ctxb = torch.empty(lib.snws_sn_ctxb_floats(N, C), dtype=torch.float32, device=dev)
    L.call("snws_bn_bwd_coeff", _ptr(s["ctx"]), _ptr(ctxb), _ptr(packb), N, C, float(s["count"]), int(s["batch"]), st)
    dx = new_nhwc(N, C, H, W, dys.dtype, dev)
    if post is not None:
        _launch("sn_bwd_apply", 3.0 * dys.element_size() * N * HW * C, "snws_sn_bwd_apply_post", _ptr(dys), lddy, _ptr(s["xs"]),
                s["ldx"], _ptr(dx), C, dt, _ptr(s["ctx"]), _ptr(ctxb), _ptr(s["beta"]), _ptr(post), N, HW, C, int(s["relu"]), st)
    else:
        _launch("sn_bwd_apply", 3.0 * dys.element_size() * N * HW * C, "snws_sn_bwd_apply", _ptr(dys), lddy, _ptr(s["xs"]),
                s["ldx"], None, 0, _ptr(dx), C, dt, _ptr(s["ctx"]), _ptr(ctxb), _ptr(s["beta"]), N, HW, C, int(s["relu"]), st)
    return dx, grad_fold(dgamma), grad_fold(dbeta)


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d.forward (+ optional ReLU) as one differentiable op on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, sync, group, clamp_eps,
                post=None):
        y, saved = bn_forward(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, sync, group,
                              clamp_eps, post)
        ctx.s = saved
        ctx.has = (gamma is not None, beta is not None)
        ctx.shapes = (None if gamma is None else gamma.shape, None if beta is None else beta.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dg, db = bn_backward(dy, ctx.s)
        ctx.s = None
        return (dx, _view(dg, ctx.shapes[0]) if ctx.has[0] else None, _view(db, ctx.shapes[1]) if ctx.has[1] else None,
                None, None, None, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------------ autograd
class Conv2dwsFn(torch.autograd.Function):
    """Conv2dws.forward of modeling/weightstd.py:12-18 as one differentiable op."""

    @staticmethod
    def forward(ctx, x, weight, bias, prep, force_direct):
        y, _, saved = conv_forward(x, prep, bias, want_stats=False, force_direct=force_direct)
        ctx.in_segs, ctx.gslot = input_segments(x), grad_slot(x)
        ctx.prep, ctx.saved, ctx.has_bias, ctx.bias_key = prep, saved, bias is not None, grad_key(bias)
        ctx.save_for_backward(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        nx, nw, nb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx, dw, db = conv_backward(dy, ctx.prep, ctx.saved, weight, nx, nw, nb, ctx.bias_key, ctx.in_segs, ctx.gslot)
        ctx.in_segs = ctx.gslot = None
        if dw is not None:
            dw = dw.to(weight.dtype)
        return dx, dw, db, None, None


GEMM_1X1 = os.environ.get("SNWS_CONV1X1_GEMM", "1") != "0"


def gemm1x1_ok(n_out, n_in):
    """Does the wide 1x1 kernel (snws_gemm1x1) take a product with n_out output and n_in contracted channels?"""
    return GEMM_1X1 and bool(L.lib().snws_gemm1x1_supported(int(n_out), int(n_in)))


# Per-tile moments a convolution's store epilogue left for the normalisation layer behind it (conv -> BatchNorm pairs of
# the OCR head): the producer attaches them to its output tensor, bn_forward uses them if the tensor it receives is
# that very tensor, unmodified, and otherwise runs its own statistics pass.
BN_STAT_FUSION = os.environ.get("SNWS_BN_STAT_FUSION", "1") != "0"
BN_STAT_HITS = 0        # BatchNorm calls that took their moments from the producer (tests read it)
_LAST_STAT = None


def attach_stat(y):
    """Moves the moments the last producer left (see _LAST_STAT) onto its output tensor y."""
    global _LAST_STAT
    st, _LAST_STAT = _LAST_STAT, None
    if st is not None and st["ptr"] == y.data_ptr() and st["shape"] == tuple(y.shape):
        st["version"] = y._version
        y._snws_stat = st
    return y


def gemm1x1(x, w2d, bias=None, tag="gemm1x1_fwd", out=None, accumulate=False, want_stats=False):
    """y[n, :, h, w] = w2d @ x[n, :, h, w] (+ bias) for a logical (N, K, H, W) bf16 tensor x with channels-last storage
    (a channel slice of a wider buffer is fine) and a bf16 (C_out, K) matrix; returns (y, x's NHWC view, its pixel
    stride).  out: an existing (N, C_out, H, W) NHWC bf16 tensor to write; with accumulate the product is added to it."""
    N, K, H, W = x.shape
    xs, ldx = as_nhwc(x)
    if ldx % 8 or xs.data_ptr() % 16:
        xs, ldx = _pad_channels(xs, ldx, K)
    n_out = w2d.shape[0]
    y = new_nhwc(N, n_out, H, W, torch.bfloat16, x.device) if out is None else out
    ys, ldy = as_nhwc(y)
    if ys.data_ptr() != y.data_ptr() or y.dtype != torch.bfloat16 or tuple(y.shape) != (N, n_out, H, W):
        raise RuntimeError("gemm1x1: out must be an NHWC bf16 tensor of the result's shape")
    if accumulate and bias is not None:
        raise RuntimeError("gemm1x1: no bias in accumulate mode")
    T = L.lib().snws_gemm1x1_stat_tiles(H * W) if (want_stats and BN_STAT_FUSION) else -1
    if T > 0:
        global _LAST_STAT
        b32 = None if bias is None else (bias.detach() if bias.dtype == torch.float32 else bias.detach().float())
        stat = torch.empty((N, T, n_out, 2), dtype=torch.float32, device=x.device)
        _launch(tag, 2.0 * N * H * W * n_out * K, "snws_gemm1x1_stats", _ptr(xs), ldx, _ptr(w2d), _ptr(b32), _ptr(ys), ldy,
                N * H * W, n_out, K, int(bool(accumulate)), H * W, _ptr(stat), _stream())
        _LAST_STAT = dict(ptr=y.data_ptr(), shape=tuple(y.shape), part=stat, T=T,
                          shift=torch.zeros(n_out, dtype=torch.float32, device=x.device))
    elif accumulate:
        _launch(tag, 2.0 * N * H * W * n_out * K, "snws_gemm1x1_acc", _ptr(xs), ldx, _ptr(w2d), _ptr(ys), ldy, N * H * W, n_out,
                K, _stream())
    else:
        b32 = None if bias is None else (bias.detach() if bias.dtype == torch.float32 else bias.detach().float())
        _launch(tag, 2.0 * N * H * W * n_out * K, "snws_gemm1x1", _ptr(xs), ldx, _ptr(w2d), _ptr(b32), _ptr(ys), ldy,
                N * H * W, n_out, K, _stream())
    return y, xs, ldx


def slot_buffer(slot, shape, dtype):
    """The tensor parked in a shared-gradient slot, if a kernel can add a (shape, dtype) gradient straight into it
    (NHWC storage, 16-byte aligned pixel rows); else None."""
    buf = slot.buf if slot is not None else None
    if buf is None or buf.dtype != dtype or tuple(buf.shape) != tuple(shape):
        return None
    b_, ld = as_nhwc(buf)
    if b_.data_ptr() != buf.data_ptr() or ld % 8 or b_.data_ptr() % 16:
        return None
    global GRAD_SHARE_HITS
    GRAD_SHARE_HITS += 1
    return buf


def gemm1x1_dgrad(dyb, w_t, shape, slot):
    """Input gradient of a wide 1x1 convolution (dyb bf16, w_t the transposed bf16 filter (C_in, C_out)), delivered to
    the input's shared-gradient slot when it has one: a later contributor adds in the kernel's epilogue.  Returns what
    the autograd function hands back for that input."""
    tgt = slot_buffer(slot, shape, torch.bfloat16)
    if tgt is not None:
        gemm1x1(dyb, w_t, None, tag="gemm1x1_dgrad", out=tgt, accumulate=True)
        return None
    dx, _, _ = gemm1x1(dyb, w_t, None, tag="gemm1x1_dgrad")
    return deliver_grad(slot, dx)


def _gemm_1x1_ok(x, cfg):
    """Plain 1x1 convolutions wide enough (>= 128 channels both sides) that a 64-channel output tile would re-read
    the pixels more than once: both the product and its input gradient run on snws_gemm1x1."""
    return cfg.kh == 1 and cfg.kw == 1 and x.dtype == torch.bfloat16 and x.is_cuda and cfg.I >= 128 and cfg.O >= 128 and \
        gemm1x1_ok(cfg.O, cfg.I) and gemm1x1_ok(cfg.I, cfg.O)


class _PlainCfg:
    """What _conv_backward_weights reads of a PreparedWeight, for a convolution that never built filter images."""
    standardize = False

    def __init__(self, cfg):
        self.cfg = cfg


class PlainConvFn(torch.autograd.Function):
    """A plain (not standardized) convolution through the same kernels - the nn.Conv2d classification head."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False):
        # want_stats: a normalisation layer reads the output next - leave it the output's per-tile moments (attach_stat)
        global _LAST_STAT
        _LAST_STAT = None
        cfg = ConvCfg(weight.shape, 1, (weight.shape[2] - 1) // 2, 1, 1)
        ctx.in_segs, ctx.gslot = input_segments(x), grad_slot(x)
        ctx.gemm_w = None
        if ctx.in_segs is None and _gemm_1x1_ok(x, cfg):
            # a wide plain 1x1 convolution of an NHWC tensor IS a GEMM [pixels x C_in] x [C_in x C_out]: the wide-tile
            # kernel stages the pixels once per 128-256 output channels instead of once per 64; the weight gradient
            # stays on snws_conv_mfma_wgrad (f32 result)
            N, C, H, W = x.shape
            wb = weight.detach().reshape(cfg.O, C).to(torch.bfloat16)
            y, xs, ldx = gemm1x1(x, wb, bias, want_stats=want_stats)
            ctx.gemm_w = wb
            ctx.prep = _PlainCfg(cfg)
            ctx.saved, ctx.has_bias, ctx.bias_key = (xs, ldx, (N, C, H, W), True), bias is not None, grad_key(bias)
            ctx.save_for_backward(weight)
            return y
        prep = PreparedWeight(weight, cfg, x.dtype == torch.bfloat16, standardize=False)
        y, stat, saved = conv_forward(x, prep, bias, want_stats=bool(want_stats and BN_STAT_FUSION))
        if stat is not None:      # (N, T, C, 2) moments of (y - bias) from the conv epilogue
            shift = _f32c(bias).view(-1) if bias is not None else torch.zeros(cfg.O, dtype=torch.float32, device=x.device)
            _LAST_STAT = dict(ptr=y.data_ptr(), shape=tuple(y.shape), part=stat, T=stat.shape[1], shift=shift)
        ctx.prep, ctx.saved, ctx.has_bias, ctx.bias_key = prep, saved, bias is not None, grad_key(bias)
        ctx.save_for_backward(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        nx, nw, nb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.gemm_w is not None:
            wb, (xs, ldx, (N, C, H, W), _) = ctx.gemm_w, ctx.saved
            dyb = dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16)
            dys, lddy = as_nhwc(dyb)
            dx = dw = db = None
            if nx:
                dx = gemm1x1_dgrad(dyb, wb.t().contiguous(), (N, C, H, W), ctx.gslot)
            if nw or nb:
                if lddy % 8 or dys.data_ptr() % 16:
                    dys = dys.contiguous(memory_format=torch.channels_last)
                    lddy = dys.shape[1]
                dw, db = _conv_backward_weights(dys, lddy, ctx.prep, ctx.saved, weight, nw, nb, ctx.bias_key)
            ctx.saved = ctx.gemm_w = ctx.gslot = None
            return dx, None if dw is None else dw.to(weight.dtype), db, None
        dx, dw, db = conv_backward(dy, ctx.prep, ctx.saved, weight, nx, nw, nb, ctx.bias_key, ctx.in_segs, ctx.gslot)
        ctx.in_segs = ctx.gslot = None
        if dw is not None:
            dw = dw.to(weight.dtype)
        return dx, dw, db, None


PAIR_CONV = os.environ.get("SNWS_PAIR_CONV", "1") != "0"            # A/B switch of Conv1x1PairFn
DROPOUT_FOLD = os.environ.get("SNWS_DROPOUT_FOLD", "1") != "0"      # A/B switch of HipBatchNorm2d.with_dropout's one-pass form


def conv1x1_pair_ok(a, b, conv):
    """Can conv(torch.cat([a, b], 1)) for the bias-free 1x1 nn.Conv2d `conv` run as two products into one output?"""
    if not (PAIR_CONV and a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 4 and a.shape[2:] == b.shape[2:]):
        return False
    w = conv.weight
    Ca, Cb, O = a.shape[1], b.shape[1], w.shape[0]
    return conv.bias is None and tuple(w.shape[1:]) == (Ca + Cb, 1, 1) and conv.stride == (1, 1) and conv.groups == 1 and \
        all(gemm1x1_ok(n, k) for n, k in ((O, Ca), (O, Cb), (Ca, O), (Cb, O)))


class Conv1x1PairFn(torch.autograd.Function):
    """F.conv2d(torch.cat([a, b], 1), weight) for a 1x1 filter (modeling/UNet3p_res_ocr_SNws.py:130) as
    weight[:, :Ca] a + weight[:, Ca:] b: the second product adds into the first one's output (snws_gemm1x1_acc), so the
    concatenated tensor is never built, and in backward each input receives its own product (into its shared-gradient
    slot when it has one)."""

    @staticmethod
    def forward(ctx, a, b, weight, want_stats=False):
        global _LAST_STAT
        _LAST_STAT = None
        N, Ca, H, W = a.shape
        Cb, O = b.shape[1], weight.shape[0]
        w2 = weight.detach().reshape(O, Ca + Cb).to(torch.bfloat16)
        wa, wb = w2[:, :Ca].contiguous(), w2[:, Ca:].contiguous()
        y, as_, lda = gemm1x1(a, wa, None, tag="gemm1x1_fwd")
        _, bs, ldb = gemm1x1(b, wb, None, tag="gemm1x1_fwd", out=y, accumulate=True, want_stats=want_stats)
        ctx.saved = (as_, lda, bs, ldb, wa, wb, (N, Ca, Cb, O, H, W))
        ctx.slots = (grad_slot(a), grad_slot(b))
        ctx.save_for_backward(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        as_, lda, bs, ldb, wa, wb, (N, Ca, Cb, O, H, W) = ctx.saved
        dyb = dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16)
        dys, lddy = as_nhwc(dyb)
        if lddy % 8 or dys.data_ptr() % 16:
            dyb = dyb.contiguous(memory_format=torch.channels_last)
            dys, lddy = as_nhwc(dyb)
        da = db = dw = None
        # (b first: it is the tensor with other consumers, and its slot is more likely to hold a buffer already)
        if ctx.needs_input_grad[1]:
            db = gemm1x1_dgrad(dyb, wb.t().contiguous(), (N, Cb, H, W), ctx.slots[1])
        if ctx.needs_input_grad[0]:
            da = gemm1x1_dgrad(dyb, wa.t().contiguous(), (N, Ca, H, W), ctx.slots[0])
        if ctx.needs_input_grad[2]:
            halves = []
            for xs, ldx, C in ((as_, lda, Ca), (bs, ldb, Cb)):
                cfg = ConvCfg((O, C, 1, 1), 1, 0, 1, 1)
                g, _ = _conv_backward_weights(dys, lddy, _PlainCfg(cfg), (xs, ldx, (N, C, H, W), True), None, True, False, None)
                halves.append(g)
            dw = torch.cat(halves, 1).to(weight.dtype)
        ctx.saved = ctx.slots = None
        return da, db, dw, None


class SwitchNormFn(torch.autograd.Function):
    """SwitchNorm2d.forward of modeling/snorm.py:41-81 (optionally + ReLU) as one differentiable op."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean_weight, var_weight, p, relu):
        y, saved = sn_forward(x, p, relu)
        ctx.s = saved
        ctx.shapes = (gamma.shape, beta.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dg, db, dmw, dvw = sn_backward(dy, ctx.s)
        return dx, _view(dg, ctx.shapes[0]), _view(db, ctx.shapes[1]), dmw, dvw, None, None


class ConvSnReluFn(torch.autograd.Function):
    """The ConvSnRelu block of modeling/UNet3p_SNws.py:29-33 fused: the conv epilogue hands SwitchNorm its
    partial moments, SwitchNorm applies the ReLU, and only the conv output is kept for backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, mean_weight, var_weight, prep, p, relu, out=None):
        conv_out, stat, saved = conv_forward(x, prep, bias, want_stats=True)
        shift = None
        if stat is not None:
            shift = _f32c(bias) if bias is not None else torch.zeros(weight.shape[0], dtype=torch.float32,
                                                                       device=x.device)
        y, sn_saved = sn_forward(conv_out, p, relu, stat_part=stat, stat_shift=shift, out=out)
        ctx.prep, ctx.saved, ctx.sn, ctx.has_bias = prep, saved, sn_saved, bias is not None
        ctx.bias_key, ctx.in_segs, ctx.gslot = grad_key(bias), input_segments(x), grad_slot(x)
        nb = bias is not None and ctx.needs_input_grad[2]
        sn_saved["dx_sum_req"] = (bool(nb), ctx.bias_key if nb else None)
        ctx.shapes = (gamma.shape, beta.shape)
        ctx.save_for_backward(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        # the gradient of a bias added right in front of the normalisation is the per-channel sum of dconv, which the
        # SwitchNorm backward has in closed form: the weight-gradient kernel need not sum the gradient tile's columns
        nx, nw, nb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        res = sn_backward(dy, ctx.sn, want_dx_sum=nb, dx_sum_key=ctx.bias_key)
        dconv, dg, dbeta, dmw, dvw = res[:5]
        ctx.sn = None
        dx, dw, _ = conv_backward(dconv, ctx.prep, ctx.saved, weight, nx, nw, False, ctx.bias_key, ctx.in_segs, ctx.gslot)
        db = res[5] if nb else None
        ctx.in_segs = ctx.gslot = None
        ctx.saved = None
        if dw is not None:
            dw = dw.to(weight.dtype)
        return dx, dw, db, _view(dg, ctx.shapes[0]), _view(dbeta, ctx.shapes[1]), dmw, dvw, None, None, None, None


HEAD_FUSION = os.environ.get("SNWS_HEAD_FUSION", "1") != "0"


def head_fusable(x, conv_cfg, head):
    """Can ConvSnRelu + the 1x1 classification head run as the fused op?  bf16 MFMA path, 64 or 320 channels into a
    bias-or-not 1x1 stride-1 nn.Conv2d with at most 16 classes."""
    return HEAD_FUSION and x.dtype == torch.bfloat16 and conv_cfg.mfma_shape_ok() and conv_cfg.O in (64, 320) and \
        tuple(head.kernel_size) == (1, 1) and tuple(head.stride) == (1, 1) and tuple(head.padding) == (0, 0) and \
        head.groups == 1 and head.in_channels == conv_cfg.O and head.out_channels <= 16


class ConvSnReluHeadFn(torch.autograd.Function):
    """ConvSnRelu (modeling/UNet3p_SNws.py:29-33) followed by the 1x1 classification head
    (modeling/UNet3p_SNws.py:93,158) as one op that never stores the normalised activation or its gradient: the
    SwitchNorm apply pass multiplies by the head filter on the fly, and the backward passes rebuild
    W_head^T * dlogits where they need it.  Returns the raw logits (N, 16, H, W) bf16 channels-last; channels
    >= n_class are padding."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, mean_weight, var_weight, head_w, head_b, prep, p):
        conv_out, stat, saved = conv_forward(x, prep, bias, want_stats=True)
        if saved[3] is not True or stat is None:
            raise RuntimeError("ConvSnReluHeadFn needs the bf16 MFMA convolution path")
        shift = _f32c(bias) if bias is not None else torch.zeros(weight.shape[0], dtype=torch.float32, device=x.device)
        _, sn = sn_forward(conv_out, p, True, stat_part=stat, stat_shift=shift, apply=False)
        N, C, H, W = conv_out.shape
        ncls = head_w.shape[0]
        hw16 = torch.zeros((16, C), dtype=torch.bfloat16, device=x.device)
        hw16[:ncls] = head_w.detach().reshape(ncls, C)
        hb16 = None
        if head_b is not None:
            hb16 = torch.zeros(16, dtype=torch.float32, device=x.device)
            hb16[:ncls] = head_b.detach()
        logits = new_nhwc(N, 16, H, W, torch.bfloat16, x.device)
        _launch("sn_apply_head", 2.0 * N * H * W * (C + 16), "snws_sn_apply_head", _ptr(sn["xs"]), sn["ldx"], _ptr(sn["ctx"]),
                _ptr(sn["beta"]), _ptr(hw16), _ptr(hb16), _ptr(logits), N, H * W, C, 1, _stream())
        ctx.prep, ctx.saved, ctx.sn, ctx.has_bias = prep, saved, sn, bias is not None
        ctx.bias_key, ctx.in_segs, ctx.gslot = grad_key(bias), input_segments(x), grad_slot(x)
        ctx.shapes = (gamma.shape, beta.shape)
        ctx.hw16, ctx.ncls, ctx.head_keys = hw16, ncls, (grad_key(head_w), grad_key(head_b))
        ctx.head_shape, ctx.has_head_bias = tuple(head_w.shape), head_b is not None
        ctx.save_for_backward(weight)
        return logits

    @staticmethod
    def backward(ctx, dl):
        (weight,) = ctx.saved_tensors
        s = ctx.sn
        N, C, H, W = s["shape"]
        HW = H * W
        lib, st, dev = L.lib(), _stream(), dl.device
        dls, lddl = as_nhwc(dl if dl.dtype == torch.bfloat16 else dl.to(torch.bfloat16))
        if lddl != 16 or dls.data_ptr() % 16:
            dls = dls.contiguous(memory_format=torch.channels_last).clone(memory_format=torch.channels_last)
        T = lib.snws_sn_head_tiles(HW, C, 1)
        part = torch.empty((N, T, C, 2), dtype=torch.float32, device=dev)
        hwp = torch.empty((N * T, 16, C), dtype=torch.float32, device=dev)
        hbp = torch.empty((N * T, 16), dtype=torch.float32, device=dev)
        _launch("sn_head_bwd_reduce", 2.0 * N * HW * (C + 16), "snws_sn_head_bwd_reduce", _ptr(dls), _ptr(s["xs"]), s["ldx"],
                _ptr(s["ctx"]), _ptr(s["beta"]), _ptr(ctx.hw16), _ptr(part), _ptr(hwp), _ptr(hbp), N, HW, C, 1, st)
        # the head's own gradients: fixed-order sums of the per-wave partials
        dhw = dhb = None
        if ctx.needs_input_grad[7]:
            tmp = torch.empty((16, C), dtype=torch.float32, device=dev)
            L.call("snws_partial_sum", _ptr(hwp), N * T, 16 * C, _ptr(tmp), 0, st)
            dhw = tmp[:ctx.ncls].reshape(ctx.head_shape)
        if ctx.has_head_bias and ctx.needs_input_grad[8]:
            tmpb = torch.empty(16, dtype=torch.float32, device=dev)
            L.call("snws_partial_sum", _ptr(hbp), N * T, 16, _ptr(tmpb), 0, st)
            dhb = tmpb[:ctx.ncls]
        p1 = sn_backward_pass1(None, s, partials=(part, T))
        if s["synced"]:
            _allreduce_sum(p1["packb"], s["group"])
        xs, ldx, hw16 = s["xs"], s["ldx"], ctx.hw16

        def apply_fn(ctxb):
            dx = new_nhwc(N, C, H, W, torch.bfloat16, dev)
            _launch("sn_head_bwd_apply", 2.0 * N * HW * (2 * C + 16), "snws_sn_head_bwd_apply", _ptr(dls), _ptr(xs), ldx,
                    _ptr(s["ctx"]), _ptr(ctxb), _ptr(s["beta"]), _ptr(hw16), _ptr(dx), C, N, HW, C, 1, st)
            return dx
        nx, nw, nb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        res = sn_backward_pass2(s, p1, want_dx_sum=nb, dx_sum_key=ctx.bias_key, apply_fn=apply_fn)
        dconv, dg, dbeta, dmw, dvw = res[:5]
        ctx.sn = None
        dx, dw, _ = conv_backward(dconv, ctx.prep, ctx.saved, weight, nx, nw, False, ctx.bias_key, ctx.in_segs, ctx.gslot)
        db = res[5] if nb else None
        ctx.in_segs = ctx.gslot = None
        ctx.saved = None
        if dw is not None:
            dw = dw.to(weight.dtype)
        return dx, dw, db, _view(dg, ctx.shapes[0]), _view(dbeta, ctx.shapes[1]), dmw, dvw, dhw, dhb, None, None


class ConvSnAddReluFn(torch.autograd.Function):
    """relu(SwitchNorm2d(Conv2dws(x)) + residual): the second half of Resnet_BasicBlock
    (modeling/UNet3p_res_ocr_SNws.py:152-159) as one fused op.  The residual add and the ReLU run in the
    SwitchNorm apply kernel; backward takes the ReLU mask from the saved block output."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, mean_weight, var_weight, prep, p, residual):
        conv_out, stat, saved = conv_forward(x, prep, bias, want_stats=True)
        shift = None
        if stat is not None:
            shift = _f32c(bias) if bias is not None else torch.zeros(weight.shape[0], dtype=torch.float32,
                                                                       device=x.device)
        y, sn_saved = sn_forward(conv_out, p, True, stat_part=stat, stat_shift=shift, residual=residual)
        ctx.prep, ctx.saved, ctx.sn, ctx.has_bias = prep, saved, sn_saved, bias is not None
        ctx.bias_key, ctx.in_segs, ctx.gslot = grad_key(bias), input_segments(x), grad_slot(x)
        ctx.shapes = (gamma.shape, beta.shape)
        ctx.res_dtype, ctx.res_slot = residual.dtype, grad_slot(residual)
        ctx.save_for_backward(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        y = ctx.sn["y"]
        want_res = ctx.needs_input_grad[9]
        nx, nw, nb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        res = sn_backward(dy, ctx.sn, want_dx_sum=nb, dx_sum_key=ctx.bias_key, dres_dtype=ctx.res_dtype if want_res else None)
        dconv, dg, dbeta, dmw, dvw = res[:5]
        dres = ctx.sn.pop("dres", None)
        if want_res and dres is None:
            dres = torch.where(y > 0, dy.to(y.dtype), torch.zeros((), dtype=y.dtype, device=y.device)).to(ctx.res_dtype)
        ctx.sn = None
        dx, dw, _ = conv_backward(dconv, ctx.prep, ctx.saved, weight, nx, nw, False, ctx.bias_key, ctx.in_segs, ctx.gslot)
        db = res[5] if nb else None
        ctx.in_segs = ctx.gslot = None
        ctx.saved = None
        if dw is not None:
            dw = dw.to(weight.dtype)
        dres, ctx.res_slot = deliver_grad(ctx.res_slot, dres), None
        return dx, dw, db, _view(dg, ctx.shapes[0]), _view(dbeta, ctx.shapes[1]), dmw, dvw, None, None, dres


# ------------------------------------------------------------------------------------------------ OCR attention
OCR_KERNELS = os.environ.get("SNWS_OCR_KERNELS", "1") != "0"
OCR_GATHER_C, OCR_ATTEND_C = 512, 256      # channel counts the fused kernels are built for


def _rows16(t, K):
    """(B, C, K, 1) class-major tensor -> dense bf16 (B, 16, C) matrix with rows >= K zero."""
    B, C = t.shape[0], t.shape[1]
    out = torch.zeros((B, 16, C), dtype=torch.bfloat16, device=t.device)
    out[:, :K] = t.detach().permute(0, 2, 3, 1).reshape(B, K, C)
    return out


def _from_rows(m, K, dtype):
    """(B, 16, C) -> logical (B, C, K, 1) with channels-last storage."""
    B, _, C = m.shape
    return m[:, :K].to(dtype).contiguous().view(B, K, 1, C).permute(0, 3, 1, 2)


def _class_rows(probs):
    """Class scores (B, K, H, W) as an NHWC tensor whose pixel rows hold at least 16 bf16 values."""
    ps, ldp = as_nhwc(probs)
    if ps.dtype != torch.bfloat16 or ldp < 16 or ldp % 8 or ps.data_ptr() % 16:
        B, K, H, W = probs.shape
        buf = torch.zeros((B, H, W, 16), dtype=torch.bfloat16, device=probs.device)
        buf[..., :K] = probs.detach().permute(0, 2, 3, 1)
        ps, ldp = buf.permute(0, 3, 1, 2), 16
    return ps, ldp


def ocr_gather_ok(feats, probs):
    return OCR_KERNELS and feats.is_cuda and feats.dtype == torch.bfloat16 and feats.shape[1] == OCR_GATHER_C and probs.shape[1] <= 16


def ocr_attend_ok(q, key):
    return OCR_KERNELS and q.is_cuda and q.dtype == torch.bfloat16 and q.shape[1] == OCR_ATTEND_C and key.shape[2] <= 16


class OcrGatherFn(torch.autograd.Function):
    """SpatialGather_Module.forward (modeling/UNet3p_res_ocr_SNws.py:21-28): ctx = softmax_pixels(scale * probs) @ feats
    as two streaming kernels (soft-max statistics, weighted sum) and one for the gradients.  Returns (B, C, K, 1)."""

    @staticmethod
    def forward(ctx, feats, probs, scale):
        lib, st = L.lib(), _stream()
        B, C, H, W = feats.shape
        K, HW = probs.shape[1], H * W
        fs, ldf = as_nhwc(feats)
        if ldf % 8 or fs.data_ptr() % 16:
            fs, ldf = _pad_channels(fs, ldf, C)
        ps, ldp = _class_rows(probs)
        Ts, Tn = lib.snws_ocr_tiles(HW, 0), lib.snws_ocr_tiles(HW, 1)
        stat = torch.empty((B, Ts, 16, 2), dtype=torch.float32, device=feats.device)
        L.call("snws_ocr_gather_stats", _ptr(ps), ldp, _ptr(stat), B, HW, K, float(scale), st)
        part = torch.empty((B, Tn, 16, C), dtype=torch.float32, device=feats.device)
        _launch("ocr_gather_fwd", 2.0 * B * HW * C, "snws_ocr_gather_fwd", _ptr(fs), ldf, _ptr(ps), ldp, _ptr(stat), _ptr(part),
                B, HW, C, K, float(scale), st)
        c32 = torch.empty((B, 16, C), dtype=torch.float32, device=feats.device)
        L.call("snws_partial_sum_batched", _ptr(part), B, Tn, 16 * C, _ptr(c32), 0, st)
        ctx.saved = (fs, ldf, ps, ldp, stat, c32, (B, C, H, W), K, float(scale))
        ctx.gslot = grad_slot(feats)
        return _from_rows(c32, K, feats.dtype)

    @staticmethod
    def backward(ctx, dctx):
        fs, ldf, ps, ldp, stat, c32, (B, C, H, W), K, scale = ctx.saved
        ctx.saved = None
        d16 = _rows16(dctx, K)
        dot = (c32 * d16.float()).sum(-1).contiguous()
        dprobs = new_nhwc(B, 16, H, W, torch.bfloat16, dctx.device)
        slot, ctx.gslot = ctx.gslot, None
        tgt = slot_buffer(slot, (B, C, H, W), torch.bfloat16)
        if tgt is not None:     # the feature map's other consumers have parked their gradient: add this share in the kernel
            ts, ldt = as_nhwc(tgt)
            _launch("ocr_gather_bwd", 4.0 * B * H * W * C, "snws_ocr_gather_bwd_acc", _ptr(fs), ldf, _ptr(ps), ldp, _ptr(stat),
                    _ptr(d16), _ptr(dot), _ptr(ts), ldt, _ptr(dprobs), B, H * W, C, K, scale, _stream())
            return None, dprobs[:, :K], None
        dfeats = new_nhwc(B, C, H, W, torch.bfloat16, dctx.device)
        _launch("ocr_gather_bwd", 4.0 * B * H * W * C, "snws_ocr_gather_bwd", _ptr(fs), ldf, _ptr(ps), ldp, _ptr(stat), _ptr(d16),
                _ptr(dot), _ptr(dfeats), C, _ptr(dprobs), B, H * W, C, K, scale, _stream())
        return deliver_grad(slot, dfeats), dprobs[:, :K], None


class OcrAttendFn(torch.autograd.Function):
    """The attention of _ObjectAttentionBlock.forward (modeling/UNet3p_res_ocr_SNws.py:87-100) between f_pixel /
    f_object / f_down and f_up: softmax_classes(scale * query @ key) @ value, one streaming kernel per direction.
    q (B, Ck, H, W); key, value (B, Ck, K, 1); returns (B, Ck, H, W)."""

    @staticmethod
    def forward(ctx, q, key, value, scale):
        B, C, H, W = q.shape
        K = key.shape[2]
        qs, ldq = as_nhwc(q)
        if ldq % 8 or qs.data_ptr() % 16:
            qs, ldq = _pad_channels(qs, ldq, C)
        k16, v16 = _rows16(key, K), _rows16(value, K)
        out = new_nhwc(B, C, H, W, torch.bfloat16, q.device)
        _launch("ocr_attend_fwd", 4.0 * B * H * W * C, "snws_ocr_attend_fwd", _ptr(qs), ldq, _ptr(k16), _ptr(v16), _ptr(out), C, B,
                H * W, C, K, float(scale), _stream())
        ctx.saved = (qs, ldq, k16, v16, (B, C, H, W), K, float(scale), key.dtype, value.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        qs, ldq, k16, v16, (B, C, H, W), K, scale, kdt, vdt = ctx.saved
        ctx.saved = None
        lib, st, dev = L.lib(), _stream(), dout.device
        ds, lddo = as_nhwc(dout if dout.dtype == torch.bfloat16 else dout.to(torch.bfloat16))
        if lddo % 8 or ds.data_ptr() % 16:
            ds, lddo = _pad_channels(ds, lddo, C)
        Tn = lib.snws_ocr_tiles(H * W, 2)
        dq = new_nhwc(B, C, H, W, torch.bfloat16, dev)
        dkp = torch.empty((B, Tn, 16, C), dtype=torch.float32, device=dev)
        dvp = torch.empty((B, Tn, 16, C), dtype=torch.float32, device=dev)
        _launch("ocr_attend_bwd", 6.0 * B * H * W * C, "snws_ocr_attend_bwd", _ptr(qs), ldq, _ptr(ds), lddo, _ptr(k16), _ptr(v16),
                _ptr(dq), C, _ptr(dkp), _ptr(dvp), B, H * W, C, K, scale, st)
        dk = torch.empty((B, 16, C), dtype=torch.float32, device=dev)
        dv = torch.empty((B, 16, C), dtype=torch.float32, device=dev)
        L.call("snws_partial_sum_batched", _ptr(dkp), B, Tn, 16 * C, _ptr(dk), 0, st)
        L.call("snws_partial_sum_batched", _ptr(dvp), B, Tn, 16 * C, _ptr(dv), 0, st)
        return dq, _from_rows(dk, K, kdt), _from_rows(dv, K, vdt), None


def ocr32_ok(t, n_class):
    """Float32 pixel tensors on the GPU with at most 64 classes: the generic float32 attention kernels."""
    return t.is_cuda and t.dtype == torch.float32 and n_class <= 64


def _f32_nhwc(t):
    """(B, C, H, W) float32 -> its NHWC view and pixel stride (a copy when the storage is not pixel-major)."""
    ts, ld = as_nhwc(t if t.dtype == torch.float32 else t.float())
    return ts, ld


def _rows32(t, K):
    """(B, C, K, 1) -> dense float32 (B, K, C)."""
    B, C = t.shape[0], t.shape[1]
    return t.detach().float().reshape(B, C, K).permute(0, 2, 1).contiguous()


class OcrGather32Fn(torch.autograd.Function):
    """OcrGatherFn in float32 (any channel count, K <= 64): snws_ocr32_gather_fwd / _bwd."""

    @staticmethod
    def forward(ctx, feats, probs, scale):
        B, C, H, W = feats.shape
        K, HW = probs.shape[1], H * W
        fs, ldf = _f32_nhwc(feats)
        ps, ldp = _f32_nhwc(probs)
        stat = torch.empty((B, K, 2), dtype=torch.float32, device=feats.device)
        c32 = torch.empty((B, K, C), dtype=torch.float32, device=feats.device)
        _launch("ocr32_gather_fwd", 4.0 * B * HW * C, "snws_ocr32_gather_fwd", _ptr(fs), ldf, _ptr(ps), ldp, _ptr(stat), _ptr(c32),
                B, HW, C, K, float(scale), _stream())
        ctx.saved = (fs, ldf, ps, ldp, stat, c32, (B, C, H, W), K, float(scale))
        ctx.gslot = grad_slot(feats)
        return c32.permute(0, 2, 1).unsqueeze(3)                       # (B, C, K, 1)

    @staticmethod
    def backward(ctx, dctx):
        fs, ldf, ps, ldp, stat, c32, (B, C, H, W), K, scale = ctx.saved
        ctx.saved = None
        d32 = _rows32(dctx, K)
        dot = (c32 * d32).sum(-1).contiguous()
        dprobs = new_nhwc(B, K, H, W, torch.float32, dctx.device)
        dps, lddp = as_nhwc(dprobs)
        slot, ctx.gslot = ctx.gslot, None
        tgt = slot_buffer(slot, (B, C, H, W), torch.float32) if slot is not None else None
        dfeats = tgt if tgt is not None else new_nhwc(B, C, H, W, torch.float32, dctx.device)
        dfs, lddf = as_nhwc(dfeats)
        _launch("ocr32_gather_bwd", 8.0 * B * H * W * C, "snws_ocr32_gather_bwd", _ptr(fs), ldf, _ptr(ps), ldp, _ptr(stat), _ptr(d32),
                _ptr(dot), _ptr(dfs), lddf, _ptr(dps), lddp, B, H * W, C, K, scale, int(tgt is not None), _stream())
        if tgt is not None:
            return None, dprobs, None
        return deliver_grad(slot, dfeats), dprobs, None


class OcrAttend32Fn(torch.autograd.Function):
    """OcrAttendFn in float32 (any channel count, K <= 64): snws_ocr32_attend_fwd / _bwd."""

    @staticmethod
    def forward(ctx, q, key, value, scale):
        B, C, H, W = q.shape
        K = key.shape[2]
        qs, ldq = _f32_nhwc(q)
        k32, v32 = _rows32(key, K), _rows32(value, K)
        out = new_nhwc(B, C, H, W, torch.float32, q.device)
        _launch("ocr32_attend_fwd", 8.0 * B * H * W * C, "snws_ocr32_attend_fwd", _ptr(qs), ldq, _ptr(k32), _ptr(v32), _ptr(out), C,
                B, H * W, C, K, float(scale), _stream())
        ctx.saved = (qs, ldq, k32, v32, (B, C, H, W), K, float(scale), key.dtype, value.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        qs, ldq, k32, v32, (B, C, H, W), K, scale, kdt, vdt = ctx.saved
        ctx.saved = None
        lib, st, dev = L.lib(), _stream(), dout.device
        ds, lddo = _f32_nhwc(dout)
        runs = lib.snws_ocr32_runs(H * W)
        dq = new_nhwc(B, C, H, W, torch.float32, dev)
        dkp = torch.empty((B, runs, K, C), dtype=torch.float32, device=dev)
        dvp = torch.empty((B, runs, K, C), dtype=torch.float32, device=dev)
        _launch("ocr32_attend_bwd", 12.0 * B * H * W * C, "snws_ocr32_attend_bwd", _ptr(qs), ldq, _ptr(ds), lddo, _ptr(k32),
                _ptr(v32), _ptr(dq), C, _ptr(dkp), _ptr(dvp), B, H * W, C, K, scale, st)
        dk = torch.empty((B, K, C), dtype=torch.float32, device=dev)
        dv = torch.empty((B, K, C), dtype=torch.float32, device=dev)
        L.call("snws_partial_sum_batched", _ptr(dkp), B, runs, K * C, _ptr(dk), 0, st)
        L.call("snws_partial_sum_batched", _ptr(dvp), B, runs, K * C, _ptr(dv), 0, st)
        to4 = lambda m, dt: m.to(dt).permute(0, 2, 1).unsqueeze(3)       # noqa: E731   (B, K, C) -> (B, C, K, 1)
        return dq, to4(dk, kdt), to4(dv, vdt), None


# ------------------------------------------------------------------------------------------------ conv of an up-sampled map
class PreparedTaps:
    """The standardized filter of a Conv2dws re-laid as a 1x1 filter with taps*O output channels (tap-major):
    Z = W_taps x at the SOURCE resolution is all the matrix work conv(upsample(x)) needs."""

    def __init__(self, weight, cfg, want_mfma):
        require_gpu(weight, "Conv2dws")
        w = _f32c(weight)
        what = torch.empty_like(w)
        L.call("snws_ws_fwd", _ptr(w), cfg.O, cfg.Ig, cfg.kh, cfg.kw, _ptr(what), None, None, None, _stream())
        wre = what.permute(2, 3, 0, 1).reshape(cfg.kh * cfg.kw * cfg.O, cfg.Ig, 1, 1).contiguous()
        self.cfg = cfg
        self.cfg_re = ConvCfg(wre.shape, 1, 0, 1, 1)
        self.prep = PreparedWeight(wre, self.cfg_re, want_mfma, standardize=False)
        # the same filter as a plain (taps*O, I) bf16 matrix for the wide 1x1 kernel, when it takes the shape
        self.gemm_w = None
        if want_mfma and gemm1x1_ok(wre.shape[0], cfg.Ig):
            self.gemm_w = wre.reshape(wre.shape[0], cfg.Ig).to(torch.bfloat16)


class TapConvFn(torch.autograd.Function):
    """x (N,I,h,w) -> Z (N, taps*O, h, w): the 1x1 half of conv(upsample(x)).  Backward receives the sum of dZ over
    every scale that used Z and runs dgrad / wgrad / the standardisation backward once."""

    @staticmethod
    def forward(ctx, x, weight, taps_prep):
        if taps_prep.gemm_w is not None and x.dtype == torch.bfloat16:
            z, xs, ldx = gemm1x1(x, taps_prep.gemm_w, None, tag="gemm1x1_taps")
            saved = (xs, ldx, tuple(x.shape), True)
        else:
            z, _, saved = conv_forward(x, taps_prep.prep, None)
        ctx.tp, ctx.saved, ctx.in_segs, ctx.gslot = taps_prep, saved, input_segments(x), grad_slot(x)
        ctx.save_for_backward(weight)
        return z

    @staticmethod
    def backward(ctx, dz):
        (weight,) = ctx.saved_tensors
        tp = ctx.tp
        cfg = tp.cfg
        dx, dwre, _ = conv_backward(dz, tp.prep, ctx.saved, None, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                    False, None, ctx.in_segs, ctx.gslot)
        ctx.in_segs = ctx.gslot = None
        dw = None
        if dwre is not None:
            g = dwre.view(cfg.kh, cfg.kw, cfg.O, cfg.Ig).permute(2, 0, 1, 3).contiguous()
            w32 = _f32c(weight)
            dw = grad_out(grad_key(weight), (cfg.O, cfg.Ig, cfg.kh, cfg.kw), g.device, acc_ok=True)
            L.call("snws_ws_bwd", _ptr(w32), _ptr(g), cfg.Ig, _ptr(dw), cfg.O, cfg.Ig, cfg.kh, cfg.kw,
                   int(grad_accumulates(dw)), _stream())
            dw = grad_fold(dw)
            if dw is not None:
                dw = dw.to(weight.dtype)
        ctx.saved = None
        return dx, dw, None


class UpTapSnReluFn(torch.autograd.Function):
    """Z -> SwitchNorm(ReLU)(bias + sum_t up_k(Z_t) shifted by t): the memory-bound half of
    ConvSnRelu(nn.Upsample(k)(x)), modeling/UNet3p_SNws.py:131-155, with the block's SwitchNorm + ReLU."""

    @staticmethod
    def forward(ctx, z, bias, gamma, beta, mean_weight, var_weight, p, k, ksize, c_out, relu, out=None):
        require_gpu(z, "ConvSnRelu")
        ctx.gslot = grad_slot(z)
        N, CZ, h, w = z.shape
        zs, ldz = as_nhwc(z)
        conv_out = new_nhwc(N, c_out, h * k, w * k, z.dtype, z.device)
        b32 = None if bias is None else _f32c(bias)
        work = float(z.element_size()) * N * c_out * (h * w * ksize * ksize + h * k * w * k)
        stat = shift = None
        if z.dtype == torch.bfloat16 and ksize == 3 and c_out % 64 == 0 and k in (2, 4, 8, 16) and ldz % 8 == 0:
            # the LDS-staged kernel also hands SwitchNorm the per-strip moments of its (rounded) output
            T = L.lib().snws_upconv_stat_tiles(h * k, w * k)
            stat = torch.empty((N, T, c_out, 2), dtype=torch.float32, device=z.device)
            try:
                _launch("upconv_fwd", work, "snws_upconv_fwd_stats", _ptr(zs), ldz, _ptr(conv_out), c_out, _ptr(b32),
                        dtype_code(z), N, h, w, c_out, k, ksize, _ptr(stat), _stream())
                shift = b32 if b32 is not None else torch.zeros(c_out, dtype=torch.float32, device=z.device)
            except L.SnwsUnsupported:
                stat = None
        if stat is None:
            _launch("upconv_fwd", work, "snws_upconv_fwd", _ptr(zs), ldz, _ptr(conv_out), c_out, _ptr(b32), dtype_code(z),
                    N, h, w, c_out, k, ksize, _stream())
        y, sn_saved = sn_forward(conv_out, p, relu, stat_part=stat, stat_shift=shift, out=out)
        ctx.sn, ctx.meta, ctx.has_bias = sn_saved, (N, h, w, k, ksize, c_out, CZ), bias is not None
        ctx.bias_key = grad_key(bias)
        sn_saved["dx_sum_req"] = (True, ctx.bias_key)
        ctx.shapes = (gamma.shape, beta.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, h, w, k, ksize, c_out, CZ = ctx.meta
        dconv, dg, dbeta, dmw, dvw, dxsum = sn_backward(dy, ctx.sn, want_dx_sum=True, dx_sum_key=ctx.bias_key)
        ctx.sn = None
        work = float(dconv.element_size()) * N * c_out * (h * w * ksize * ksize + h * k * w * k)
        slot, ctx.gslot = ctx.gslot, None
        dz, added = None, False
        parked = None if slot is None else slot.buf
        if parked is not None and k == 2 and parked.dtype == dconv.dtype == torch.bfloat16 and \
                tuple(parked.shape) == (N, CZ, h, w):
            # a later consumer of a shared tap map: the kernel adds its share into the parked gradient where it can
            ps, ldp = as_nhwc(parked)
            if ps.data_ptr() == parked.data_ptr():
                try:
                    _launch("upconv_bwd", work, "snws_upconv_bwd_acc", _ptr(dconv), c_out, _ptr(ps), ldp, dtype_code(dconv),
                            N, h, w, c_out, k, ksize, _stream())
                    global GRAD_SHARE_HITS
                    GRAD_SHARE_HITS += 1
                    added = True
                except L.SnwsUnsupported:
                    pass
        if not added:
            dz = new_nhwc(N, CZ, h, w, dconv.dtype, dconv.device)
            _launch("upconv_bwd", work, "snws_upconv_bwd", _ptr(dconv), c_out, _ptr(dz), CZ, dtype_code(dconv), N, h, w,
                    c_out, k, ksize, _stream())
            dz = deliver_grad(slot, dz)
        db = dxsum if ctx.has_bias else None
        return dz, db, _view(dg, ctx.shapes[0]), _view(dbeta, ctx.shapes[1]), dmw, dvw, None, None, None, None, None, None


class AssembleFn(torch.autograd.Function):
    """torch.cat((parts...), 1) without the copy: the parts were written by their producers straight into channel
    slices of `buf` (the `out=` argument of the fused blocks); forward hands out `buf`, backward hands every
    producer its channel slice of the gradient as a strided view (the kernels take a pixel stride)."""

    @staticmethod
    def forward(ctx, buf, *parts):
        off = 0
        for t in parts:
            c = t.shape[1]
            if t.data_ptr() != buf[:, off:off + c].data_ptr() or t.shape[2:] != buf.shape[2:]:
                raise ValueError("AssembleFn: part is not the matching channel slice of the buffer")
            off += c
        if off != buf.shape[1]:
            raise ValueError("AssembleFn: parts cover %d of %d channels" % (off, buf.shape[1]))
        ctx.widths = [t.shape[1] for t in parts]
        ctx.tags = None
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, off = [None], 0
        for c in ctx.widths:
            outs.append(g[:, off:off + c])
            off += c
        tags, ctx.tags = ctx.tags, None
        if tags:
            # the parts are SwitchNorm block outputs whose ONLY consumer is this concatenation, and all their incoming
            # gradients exist now: the SwitchNorm backward of all of them runs here - one batched launch per pass and
            # round (a module used for several parts keeps its order: _rounds), the [2C] payloads of a data-parallel
            # run in ONE all-reduce - and each block's own backward picks its result up ("pass2"; with SNWS_SN_BATCH=0
            # only the first half runs here and the block continues from "pass1")
            synced = tags[0]["synced"]
            slots = torch.empty(sum(2 * s["shape"][1] for s in tags), dtype=torch.float32, device=g.device) if synced else None
            o, pairs = 0, []
            for i, s in enumerate(tags):
                C = s["shape"][1]
                pairs.append((s, sn_backward_pass1(outs[1 + i], s, packb=slots[o:o + 2 * C] if synced else None,
                                                   launch=not SN_BATCHING)))
                o += 2 * C
            if not SN_BATCHING:
                for s, p1 in pairs:
                    s["pass1"] = p1
                _allreduce_sum(slots, tags[0]["group"])
                return tuple(outs)
            launches = [grp for rnd in _rounds(pairs, lambda sp: _addr(sp[0]["gamma"])) for grp in _groups(rnd, _sn_bwd_sig)]
            for grp in launches:
                _sn_bwd_a(grp)
            if synced:
                _allreduce_sum(slots, tags[0]["group"])
            for grp in launches:
                _sn_bwd_b(grp)
        return tuple(outs)


def assemble(buf, parts):
    """AssembleFn over the parts written into `buf`, carrying their SwitchNorm tags over to the assembled tensor (at
    their channel offsets) when every part has one."""
    y = AssembleFn.apply(buf, *parts)
    segs, off = [], 0
    for p in parts:
        tag = getattr(p, "_snws_seg", None)
        if not tag:
            return y
        segs += [(off + c0, c, s) for c0, c, s in tag]
        off += p.shape[1]
    y._snws_seg = segs
    # grouped SwitchNorm backward of all the parts (see AssembleFn.backward) when every part is exactly one SwitchNorm
    # layer without a residual input, all synchronised over the same process group or none of them
    states = [s for _, _, s in segs]
    if y.grad_fn is not None and len(states) == len(parts) and all(s["y"] is None for s in states) and \
            (SN_BATCHING or states[0]["synced"]) and all(s["synced"] == states[0]["synced"] for s in states) and \
            all(s["group"] is states[0]["group"] for s in states) and \
            all(c0 == sum(q.shape[1] for q in parts[:i]) and c == parts[i].shape[1] for i, (c0, c, _) in enumerate(segs)):
        y.grad_fn.tags = states
    return y


# ------------------------------------------------------------------------------------------------ loss / inference tiles
class FusedCEFn(torch.autograd.Function):
    """scale * sum over non-ignored pixels of w[target] * CrossEntropy(logits, target) (utils/loss.py:146-159 with the
    division by the normaliser - the valid-pixel count, or the sum of w[target] with class weights - and the batch
    size folded into `scale`, a device scalar), forward and backward in one pass.  logits: (N, Cp, H, W)
    channels-last, the first n_class channels are real (Cp > n_class: the padded MFMA head); the gradient of the
    padding channels is zero.  weight: None or n_class float32 class weights (nn.CrossEntropyLoss(weight=...))."""

    @staticmethod
    def forward(ctx, logits, target, n_class, scale, ignore_index, weight=None):
        part, sc, dl = ce_partials(logits, target, n_class, ignore_index, weight, scale, want_grad=True)
        ctx.dl = dl
        return part.sum() * sc[0]

    @staticmethod
    def backward(ctx, g):
        dl, ctx.dl = ctx.dl, None
        return dl.mul_(g.to(dl.dtype)), None, None, None, None, None


def ce_partials(logits, target, n_class, ignore_index=255, weight=None, scale=None, want_grad=False):
    """One launch of the cross-entropy kernel: per-block partial sums of w[target] * CE over the non-ignored pixels
    (float32, fixed order), the scale as a (1,) device tensor and - with want_grad - d(scale * sum)/d(logits).
    Without want_grad this is the forward-only form validation uses (train.py:291)."""
    require_gpu(logits, "cross entropy")
    N, Cp, H, W = logits.shape
    xs, ld = as_nhwc(logits)
    tg = target.detach().to(torch.float32).contiguous()
    if tuple(tg.shape) != (N, H, W):
        raise ValueError("cross entropy: target shape %s does not match logits %s" % (tuple(tg.shape), (N, H, W)))
    wt = None
    if weight is not None:
        wt = weight.detach().to(device=logits.device, dtype=torch.float32).contiguous()
        if wt.numel() != int(n_class):
            raise ValueError("cross entropy: %d class weights for %d classes" % (wt.numel(), int(n_class)))
    npix = N * H * W
    part = torch.empty(L.lib().snws_ce_blocks(npix), dtype=torch.float32, device=logits.device)
    dl = new_nhwc(N, Cp, H, W, xs.dtype, logits.device) if want_grad else None
    sc = None if scale is None else scale.detach().to(torch.float32).reshape(1).contiguous()
    nbytes = float(npix) * ((2.0 if want_grad else 1.0) * xs.element_size() * Cp + 4.0)
    _launch("ce_fwd_bwd", nbytes, "snws_ce_fwd_bwd_w", _ptr(xs), dtype_code(xs), ld, _ptr(tg), npix, int(n_class),
            int(ignore_index), _ptr(sc) if sc is not None else None, _ptr(wt) if wt is not None else None, _ptr(part),
            _ptr(dl) if dl is not None else None, Cp, Cp, _stream())
    return part, sc, dl


def argmax_stitch(logits, n_class, pred, y0, x0, remap_background=False):
    """pred[y0:y0+h, x0:x0+w] = argmax over the first n_class channels of ONE tile's logits (1, Cp, h, w), clipped
    to pred's extent (predict.py:196,248-253).  pred: int64 (H, W) on the device."""
    require_gpu(logits, "argmax_stitch")
    if logits.dim() != 4 or logits.shape[0] != 1:
        raise ValueError("argmax_stitch takes one tile (1, C, h, w)")
    if pred.dtype != torch.int64 or pred.dim() != 2 or not pred.is_contiguous():
        raise ValueError("argmax_stitch: pred must be a contiguous int64 (H, W) tensor")
    xs, ld = as_nhwc(logits)
    _, _, h, w = logits.shape
    L.call("snws_argmax_stitch", _ptr(xs), dtype_code(xs), ld, h, w, int(n_class), _ptr(pred), pred.shape[0],
           pred.shape[1], int(y0), int(x0), int(bool(remap_background)), _stream())
    return pred


# ------------------------------------------------------------------------------------------------ resampling
_MODE = {"copy": 0, "maxpool": 1, "bilinear": 2}


class ResampleFn(torch.autograd.Function):
    """nn.MaxPool2d(k,k,ceil_mode=True) / nn.Upsample(k,'bilinear') of modeling/UNet3p_SNws.py:49-55."""

    @staticmethod
    def forward(ctx, x, mode, k):
        require_gpu(x, "resample")
        N, C, H, W = x.shape
        xs, ld = as_nhwc(x)
        m = _MODE[mode]
        OH, OW = (H, W) if m == 0 else ((-(-H // k), -(-W // k)) if m == 1 else (H * k, W * k))
        y = new_nhwc(N, C, OH, OW, x.dtype, x.device)
        _launch("resample_fwd", float(x.element_size()) * N * C * (H * W + OH * OW), "snws_resample_fwd", _ptr(xs), ld,
                _ptr(y), C, dtype_code(x), N, H, W, C, m, k, _stream())
        ctx.meta = (m, k, (N, C, H, W), ld)
        ctx.gslot = grad_slot(x)
        ctx.save_for_backward(xs if m == 1 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        m, k, (N, C, H, W), ld = ctx.meta
        (xs,) = ctx.saved_tensors
        dys, ldd = as_nhwc(dy)
        OH, OW = dy.shape[2], dy.shape[3]
        slot, ctx.gslot = ctx.gslot, None
        acc = slot.buf if slot is not None else None
        if acc is not None and acc.dtype == dys.dtype and tuple(acc.shape) == (N, C, H, W):
            a_, lda = as_nhwc(acc)
            if a_.data_ptr() == acc.data_ptr():
                # a later contributor to a shared gradient: the kernel adds into the parked tensor (accumulate = 1)
                _launch("resample_bwd", float(dys.element_size()) * N * C * (2 * H * W + OH * OW), "snws_resample_bwd", _ptr(dys),
                        ldd, _ptr(xs), ld, _ptr(a_), lda, dtype_code(dys), N, H, W, C, m, k, 1, _stream())
                return None, None, None
        dx = new_nhwc(N, C, H, W, dys.dtype, dys.device)      # written in full by the kernel (accumulate = 0)
        _launch("resample_bwd", float(dys.element_size()) * N * C * (H * W + OH * OW), "snws_resample_bwd", _ptr(dys),
                ldd, _ptr(xs), ld, _ptr(dx), C, dtype_code(dys), N, H, W, C, m, k, 0, _stream())
        return deliver_grad(slot, dx), None, None


def max_pool_ceil(x, k):
    return ResampleFn.apply(x, "maxpool", int(k))


def upsample_bilinear(x, k):
    return ResampleFn.apply(x, "bilinear", int(k))
