"""Weight-standardized convolution, same module API as the reference's modeling/weightstd.py:8-18.

`Conv2dws` is an `nn.Conv2d` subclass with the reference's constructor signature and state_dict
(`weight`, `bias`).  forward() standardizes the filter per output channel (mean 0, unbiased std 1 over the
fan-in, eps 1e-5 added to the std) and convolves, all in HIP kernels for gfx950: a one-pass standardize +
operand-pack kernel, then the bf16 MFMA implicit-GEMM kernels for 3x3/1x1 stride-1 "same" convolutions, or the
direct kernels for every other stride / padding / dilation / groups combination and for float32.
There is no CPU path: a CPU tensor raises.
"""
import torch.nn as nn

from snws import ops


class Conv2dws(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super(Conv2dws, self).__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self._prep = None
        self._prep_key = None
        self._taps = None
        self._taps_key = None
        self.force_direct = False   # True pins the exact direct kernels (float32 tier / debugging)
        self.cache_filters = False  # a training driver that announces weight updates switches this on (snws.ops)

    def prepared(self, want_mfma):
        """Standardized filter images.  Rebuilt on every call, like the reference, unless a training driver has
        switched on this module's cache_filters and announces weight updates with snws.ops.weights_changed(); then
        a module shared by several call sites in one forward standardizes once per step."""
        w = self.weight
        key = ops.filter_cache_key(w, want_mfma, self.cache_filters)
        if self._prep is None or key is None or self._prep_key != key:
            cfg = ops.ConvCfg(w.shape, self.stride, self.padding, self.dilation, self.groups)
            self._prep = ops.PreparedWeight(w, cfg, want_mfma, reuse=self._prep if self.cache_filters else None)
            self._prep_key = key
        return self._prep

    def prepared_taps(self, want_mfma):
        """The standardized filter re-laid as a (kh*kw*O, I, 1, 1) filter for conv-of-an-up-sampled-map
        (snws.ops.TapConvFn); cached per weight version like prepared()."""
        w = self.weight
        key = ops.filter_cache_key(w, want_mfma, self.cache_filters)
        if self._taps is None or key is None or self._taps_key != key:
            cfg = ops.ConvCfg(w.shape, self.stride, self.padding, self.dilation, self.groups)
            if cfg.stride != 1 or cfg.dil != 1 or cfg.groups != 1 or cfg.kh != cfg.kw or cfg.kh not in (1, 3) or \
                    cfg.pad != (cfg.kh - 1) // 2:
                raise NotImplementedError("conv-of-upsample needs a stride-1 'same' 1x1 or 3x3 convolution")
            self._taps = ops.PreparedTaps(w, cfg, want_mfma)
            self._taps_key = key
        return self._taps

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(x.dim()))
        ops.require_gpu(x, "Conv2dws")
        import torch
        prep = self.prepared(x.dtype == torch.bfloat16 and not self.force_direct)
        return ops.Conv2dwsFn.apply(x, self.weight, self.bias, prep, self.force_direct)
