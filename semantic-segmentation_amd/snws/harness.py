"""Training / validation step harness that reproduces Trainer.training and Trainer.validation of the
reference's train.py:237-341 for the SNws models, one process per GPU.

Reference semantics kept: poly LR set every iteration (utils/lr_scheduler.py:43-70), optimizer.zero_grad ->
forward -> CrossEntropy(mean over non-ignored pixels, ignore_index 255) / batch size (utils/loss.py:146-159)
-> backward -> AdamW(lr 0.002, weight_decay 5e-4) step (train.py:81,384,393); validation = eval-mode forward,
channel argmax, Evaluator.add_batch (train.py:279-304).

Data parallelism replaces nn.DataParallel (train.py:212): every rank holds a replica and its own shard of the
batch; SwitchNorm's BN branch all-reduces its per-channel sums inside forward/backward (snws.ops), the loss is
normalised by the GLOBAL number of valid pixels and the GLOBAL batch size, and gradients are summed over ranks
in buckets that are launched as soon as their last gradient is produced, so the all-reduces overlap the rest of
backward.
"""
import math
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F


def init_distributed():
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns
    (rank, world, device).  With WORLD_SIZE unset or 1 nothing is initialised."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("SNWS_DP_SHARE_GPU") == "1"     # rehearsal: every rank on GPU 0, collectives over gloo
    if torch.cuda.is_available():
        if share:
            local = 0
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    # SNWS_DP_SELFTEST=1: a one-rank group, so a single-GPU box exercises every collective of the multi-GPU path
    if (world > 1 or os.environ.get("SNWS_DP_SELFTEST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if torch.cuda.is_available() and not share else "gloo"
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def _dp(group=None):
    from . import ops
    return ops.dp_collectives(group)


def poly_lr(base_lr, it, total_iters):
    """utils/lr_scheduler.py:48 - lr = base * (1 - T/N)^0.9."""
    return base_lr * pow((1 - 1.0 * it / total_iters), 0.9)


def _loss_normalisers(target, n, ignore_index, weight, device, group, reduce=True):
    """[sum over non-ignored pixels of w[target] (their count without class weights), batch size], float32 on the
    device, summed over the ranks of `group`: the two divisors of utils/loss.py:151-157 for the GLOBAL batch."""
    t = target.to(torch.int64)
    valid = t != ignore_index
    if weight is None:
        norm = valid.sum().to(torch.float32)
    else:       # nn.CrossEntropyLoss(weight=w, reduction='mean') divides by the sum of w[target] over valid pixels
        w = weight.to(device=device, dtype=torch.float32)
        norm = (w[t.clamp(0, w.numel() - 1)] * valid).sum()
    # (torch.full is a device fill; torch.tensor(..., device=cuda) would be a blocking host-to-device copy that
    # stalls the host until the forward pass has drained)
    cnt = torch.stack([norm, torch.full((), float(n), dtype=torch.float32, device=device)])
    if reduce and _dp(group):
        from . import ops
        ops.all_reduce(cnt, "sum", group)
    return cnt


def global_ce_loss(logits, target, ignore_index=255, group=None, weight=None):
    """The reference loss of the GLOBAL batch, as this rank's additive share: sum of the rank's per-pixel (class-
    weighted) CE over the global normaliser, over the global batch size.  Summed over ranks it equals
    CrossEntropyLoss(weight, mean)(all logits)/N_global (utils/loss.py:146-157 applied to DataParallel's gathered
    output, train.py:250-252)."""
    w = None if weight is None else weight.to(device=logits.device, dtype=torch.float32)
    ce_sum = F.cross_entropy(logits.float(), target.long(), weight=w, ignore_index=ignore_index, reduction="sum")
    cnt = _loss_normalisers(target, logits.shape[0], ignore_index, weight, logits.device, group)
    return ce_sum / cnt[0] / cnt[1]


def global_ce_loss_fused(raw_logits, n_class, target, ignore_index=255, group=None, weight=None):
    """global_ce_loss on the head kernel's raw output (N, Cp, H, W) through the fused forward+backward kernel: the
    soft-max, the loss and d(loss)/d(logits) are produced in one pass over the logits."""
    from . import ops
    cnt = _loss_normalisers(target, raw_logits.shape[0], ignore_index, weight, raw_logits.device, group)
    scale = 1.0 / (cnt[0] * cnt[1])
    return ops.FusedCEFn.apply(raw_logits, target, n_class, scale, ignore_index, weight)


class GradBuckets:
    """Sum-all-reduce of parameter gradients in ~bucket_mb buckets, overlapped with backward.

    Parameters are bucketed in REVERSE registration order (the order backward tends to finish them).  A
    post-accumulate-grad hook counts a bucket's parameters; when the last one has its final gradient (shared
    modules fire once, after all their uses are summed) the bucket is flattened and all-reduced asynchronously.
    finish() waits and copies the sums back.  Parameters that received no gradient (unused modules) are skipped
    consistently on every rank because usage is data-independent."""

    def __init__(self, params, bucket_mb=25.0, group=None, flat_grad=None, flat_offsets=None):
        """flat_grad / flat_offsets: the flat gradient buffer of snws.optim.FlatAdamW and each parameter's offset in
        it; a bucket is then a contiguous slice that is all-reduced in place (no flatten / unflatten copies)."""
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.flat_grad = flat_grad
        self.slices = None
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_mb * (1 << 20):
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.where = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self.where[p] = bi
        if flat_grad is not None:
            index = {p: i for i, p in enumerate(self.params)}
            ends = list(flat_offsets[1:]) + [flat_grad.numel()]
            self.slices = []
            for b in self.buckets:           # reverse registration order: b[-1] has the lowest offset
                lo, hi = index[b[-1]], index[b[0]]
                self.slices.append((flat_offsets[lo], ends[hi]))
        self.pending = [0] * len(self.buckets)
        self.work = []
        self.enabled = _dp(group)
        if self.enabled:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
        self.reset()

    def reset(self):
        self.pending = [len(b) for b in self.buckets]
        self.work = []

    def _launch(self, bi):
        from . import ops
        if self.slices is not None:
            lo, hi = self.slices[bi]
            flat = self.flat_grad[lo:hi]
            ops.trace_collective("bucket%d" % bi, flat.numel())
            self.work.append((ops.all_reduce(flat, "sum", self.group, async_op=True), None, ()))
            return
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
        ops.trace_collective("bucket%d" % bi, flat.numel())
        self.work.append((ops.all_reduce(flat, "sum", self.group, async_op=True), flat, ps))

    def _hook(self, p):
        bi = self.where[p]
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        if not self.enabled:
            return
        for bi, left in enumerate(self.pending):     # buckets with parameters that never got a gradient
            if left > 0:
                self._launch(bi)
        for h, flat, ps in self.work:
            if h is not None:
                h.wait()
            off = 0
            for p in ps:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self.reset()


class Trainer:
    """One-replica training driver with the reference's hyper-parameters as defaults (train.py:384-429)."""

    def __init__(self, model, lr=0.002, weight_decay=5e-4, total_iters=1000, ignore_index=255, group=None,
                 bucket_mb=25.0, fused_loss=None, optimizer="flat", class_weight=None):
        """class_weight: None or a float tensor of per-class loss weights (train.py:55-67, --use-balanced-weights)."""
        self.model = model
        self.class_weight = class_weight
        self.val_loss = None
        # classes = output channels of the last registered 1x1 classification conv the forward uses
        head = getattr(model, "cls_head", None) or getattr(model, "output_seg_map")
        self.n_class = head.out_channels
        self.fused_loss = hasattr(model, "raw_logits") if fused_loss is None else bool(fused_loss)
        self.base_lr, self.total_iters, self.ignore_index, self.group = lr, total_iters, ignore_index, group
        if optimizer == "flat":      # one fused launch over flat parameter / gradient buffers (snws.optim)
            from .optim import FlatAdamW
            self.opt = FlatAdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
        else:                        # torch's multi-tensor AdamW
            try:
                self.opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay, fused=True)
            except (RuntimeError, TypeError):
                self.opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
        if optimizer == "flat":
            self.buckets = GradBuckets(self.opt.params, bucket_mb, group, self.opt.flat_g, self.opt.offsets)
        else:
            self.buckets = GradBuckets(list(model.parameters()), bucket_mb, group)
        self.it = 0
        from . import ops
        # this driver owns the weights of ITS model and announces every update below: the model's Conv2dws modules
        # may keep their standardized filters between calls (modules of other models are not affected)
        ops.set_filter_cache(model, True)
        ops.weights_changed()

    def close(self):
        """Give the model back: its Conv2dws modules standardize on every call again, like the reference, and the
        flat optimizer lets go of the parameters' gradient slots (the Trainer must not step again afterwards)."""
        from . import ops
        ops.set_filter_cache(self.model, False)
        if hasattr(self.opt, "release"):
            self.opt.release()

    def step(self, image, target):
        """train.py:246-255.  Returns this rank's share of the global loss (a 0-d tensor; no host sync)."""
        lr = poly_lr(self.base_lr, self.it, self.total_iters)
        for g in self.opt.param_groups:
            g["lr"] = lr
        self.opt.zero_grad(set_to_none=True)
        self.model.train()
        if self.fused_loss:
            self.model.raw_logits = True
            try:
                out = self.model(image)
            finally:
                self.model.raw_logits = False
            loss = global_ce_loss_fused(out, self.n_class, target, self.ignore_index, self.group, self.class_weight)
        else:
            out = self.model(image)
            loss = global_ce_loss(out, target, self.ignore_index, self.group, self.class_weight)
        loss.backward()
        self.buckets.finish()
        self.opt.step()
        from . import ops
        ops.weights_changed()
        self.it += 1
        return loss.detach()

    @torch.no_grad()
    def validate(self, batches, evaluator):
        """train.py:279-304 with the argmax + confusion matrix on the device.  Returns the four metrics; the
        criterion of every validation batch (train.py:291-292, summed over the batches like `val_loss`) is left in
        self.val_loss, its mean over the batches (the 'Loss:' line, train.py:313) in self.val_loss_mean.  The loss
        comes from the forward-only form of the fused cross-entropy kernel on the same logits; under data
        parallelism each batch's sums are global (one all-reduce for the whole validation)."""
        from . import ops
        self.model.eval()
        evaluator.reset()
        sums = []
        for image, target in batches:
            logits = self.model(image)
            evaluator.add_batch_logits(target, logits)
            part, _, _ = ops.ce_partials(logits, target, self.n_class, self.ignore_index, self.class_weight)
            cnt = _loss_normalisers(target, logits.shape[0], self.ignore_index, self.class_weight, logits.device,
                                    self.group, reduce=False)
            sums.append(torch.cat([part.sum().reshape(1), cnt]))
        evaluator.sync(group=self.group, all_reduce=True)
        if sums:
            s = torch.stack(sums)
            if _dp(self.group):
                ops.all_reduce(s, "sum", self.group)
            per_batch = s[:, 0] / s[:, 1] / s[:, 2]
            self.val_loss = float(per_batch.sum())
            self.val_loss_mean = self.val_loss / len(sums)
        else:
            self.val_loss, self.val_loss_mean = 0.0, float("nan")
        return (evaluator.Pixel_Accuracy(), evaluator.Pixel_Accuracy_Class(),
                evaluator.Mean_Intersection_over_Union(), evaluator.Frequency_Weighted_Intersection_over_Union())


def synthetic_batch(batch, n_channels, size, n_class, device, seed=1, dtype=torch.float32):
    """Synthetic FY-4A-shaped tiles: images uniform in [0,255] (the data range after the per-tile min-max x255 of
    dataloaders/datasets/cloud_dataset.py:127-131), labels uniform integers 0..n_class-1 stored as float
    (cloud_dataset.py:132-137)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = torch.rand((batch, n_channels, size, size), generator=g) * 255.0
    lab = torch.randint(0, n_class, (batch, size, size), generator=g).float()
    return img.to(device=device, dtype=dtype), lab.to(device)
