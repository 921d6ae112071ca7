"""AdamW over ONE flat parameter buffer: the reference's torch.optim.AdamW(model.parameters(), lr, weight_decay)
(train.py:81, stepped at train.py:254) as a single fused HIP launch.

All parameters are re-pointed to views of one float32 buffer, and every parameter owns a slot in a second one, the
flat gradient.  The slots are registered as gradient sinks (snws.ops.GRAD_SINKS): the backward kernels write
d(loss)/d(parameter) straight into the slot and autograd adopts that view as .grad, so no per-parameter add or copy
is launched; a gradient that arrives any other way (a stock torch op, a second use of the parameter) is added or
copied into the slot by the post-accumulate hook.  zero_grad is one memset, a data-parallel all-reduce runs on slices
of the flat gradient without flatten/unflatten copies, and the update is one pass over four flat arrays.  state_dict() /
load_state_dict() speak torch.optim.AdamW's format, so the 'optimizer' entry of a reference checkpoint
(train.py:226, 272-277) loads and saves unchanged.  Like torch, a parameter that received no gradient in a step is
left alone in that step (no weight decay, no moment decay)."""
import weakref

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
        self._hooks = []
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = None
                ops.GRAD_SINKS[p.data_ptr()] = (self.flat_g, o, n, [False], weakref.ref(self))
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                  amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                                  fused=None)]
        self.steps = [0] * len(self.params)          # per-parameter update counts, as torch keeps them

    def _make_hook(self, i):
        o = self.offsets[i]

        def hook(p):
            self._touched[i] = True
            if p.grad is None:      # every share of this gradient was added inside the slot by the producing kernels
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
            elif p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:     # not produced in place: move it into the slot
                with torch.no_grad():
                    slot = self.flat_g[o:o + p.numel()].view(p.shape)
                    slot.copy_(p.grad)
                    p.grad = slot
        return hook

    def release(self):
        """Detach from the parameters: their gradient slots and hooks are removed, so a later backward outside this
        optimizer produces ordinary .grad tensors (the parameters keep living in the flat parameter buffer)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            sink = ops.GRAD_SINKS.get(p.data_ptr())
            if sink is not None and sink[4]() is self:
                del ops.GRAD_SINKS[p.data_ptr()]
            p.grad = None

    # -------------------------------------------------------------------------------------------- torch.optim surface
    def zero_grad(self, set_to_none=True):
        """Clears the flat gradient (one memset) and re-arms every parameter's gradient sink.  .grad is always reset
        to None so that autograd adopts the view a backward kernel wrote into (set_to_none is accepted for API
        compatibility)."""
        self.flat_g.zero_()
        for p in self.params:
            p.grad = None
            sink = ops.GRAD_SINKS.get(p.data_ptr())
            if sink is not None:
                sink[3][0] = False
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
        # the raw kernel does not bump tensor version counters: tell the standardized-filter caches
        ops.weights_changed()

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
        ops.weights_changed()
