"""AdamW over ONE flat parameter buffer: the reference's torch.optim.AdamW(model.parameters(), lr, weight_decay)
(train.py:81, stepped at train.py:254) as a single fused HIP launch.

All parameters are re-pointed to views of one float32 buffer and their .grad to views of a second one, so autograd
accumulates straight into the flat gradient, zero_grad is one memset, a data-parallel all-reduce can run on slices
of it without flatten/unflatten copies, and the update is one pass over four flat arrays.  state_dict() /
load_state_dict() speak torch.optim.AdamW's format, so the 'optimizer' entry of a reference checkpoint
(train.py:226, 272-277) loads and saves unchanged.  Like torch, a parameter that received no gradient in a step is
left alone in that step (no weight decay, no moment decay)."""
import torch

from . import _lib as L
from . import ops


class FlatAdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW: no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW runs on the MI355X HIP kernels only (there is no CPU fallback)")
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("FlatAdamW: float32 parameters on one device expected")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # every view starts 16-byte aligned
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self._touched = [False] * len(self.params)
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
                p.register_post_accumulate_grad_hook(self._make_hook(i))
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                  amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                                  fused=None)]
        self.steps = [0] * len(self.params)          # per-parameter update counts, as torch keeps them

    def _make_hook(self, i):
        def hook(_p):
            self._touched[i] = True
        return hook

    # -------------------------------------------------------------------------------------------- torch.optim surface
    def zero_grad(self, set_to_none=False):
        """Always keeps the flat gradient views (set_to_none is accepted for API compatibility and ignored)."""
        self.flat_g.zero_()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self._touched = [False] * len(self.params)

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        # contiguous runs of parameters that received a gradient and share an update count -> one launch each
        runs, i, n = [], 0, len(self.params)
        while i < n:
            if not self._touched[i]:
                i += 1
                continue
            j = i
            while j + 1 < n and self._touched[j + 1] and self.steps[j + 1] == self.steps[i]:
                j += 1
            end = self.offsets[j + 1] if j + 1 < n else self.numel
            runs.append((self.offsets[i], end, self.steps[i] + 1))
            for q in range(i, j + 1):
                self.steps[q] += 1
            i = j + 1
        st = ops._stream()
        for lo, hi, step in runs:
            L.call("snws_adamw_step", self.flat_p.data_ptr() + 4 * lo, self.flat_g.data_ptr() + 4 * lo,
                   self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, hi - lo, float(g["lr"]),
                   float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), int(step), float(grad_scale), st)

    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if self.steps[i] == 0:
                continue
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.steps[i])),
                        "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError("FlatAdamW.load_state_dict: %d parameters in the checkpoint, %d here" % (len(ids), len(self.params)))
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.steps = [0] * len(self.params)
        for pos, key in enumerate(ids):
            st = sd["state"].get(key)
            if st is None:
                continue
            p, o = self.params[pos], self.offsets[pos]
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.steps[pos] = int(float(st["step"]))
