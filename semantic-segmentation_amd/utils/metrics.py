"""Segmentation Evaluator, same API as the reference's utils/metrics.py:4-52.

`confusion_matrix` stays a public float64 (C,C) ndarray (rows = ground truth, cols = prediction) and the four
metric methods use the same numpy expressions as the reference, so every reported number is bit-identical.
The counting itself (reference: mask, astype('int'), C*gt+pred, np.bincount) runs on the GPU as an
LDS-privatized integer-atomic histogram; counts are int64 and exact.

add_batch(gt, pred) takes what train.py:294-298 passes (numpy arrays; gt float32 labels, pred int64 argmax) or
torch tensors.  add_batch_logits(gt, logits) is an extension that takes the network output itself and does
the channel argmax on the device, removing the logits D2H copy and the CPU argmax of train.py:294-296.
"""
import ctypes

import numpy as np
import torch

from snws import _lib as L

_GT = {torch.float32: L.SNWS_F32, torch.float64: L.SNWS_F64, torch.int64: L.SNWS_I64, torch.int32: L.SNWS_I32,
       torch.uint8: L.SNWS_U8}
_PRED = {torch.int64: L.SNWS_I64, torch.int32: L.SNWS_I32, torch.uint8: L.SNWS_U8}


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("Evaluator counts on the MI355X HIP histogram kernel; no GPU is visible "
                           "(there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a, table, what):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    if t.dtype not in table:
        if t.dtype in (torch.float16, torch.bfloat16):
            t = t.float()
        elif t.dtype in (torch.int8, torch.int16, torch.bool):
            t = t.to(torch.int32)
        else:
            raise TypeError("Evaluator: unsupported %s dtype %s" % (what, t.dtype))
    if not t.is_cuda:
        t = t.to(_device(), non_blocking=False)
    return t.contiguous()


class Evaluator(object):
    def __init__(self, num_class):
        self.num_class = num_class
        self.confusion_matrix = np.zeros((self.num_class,) * 2)
        self._counts = None     # device int64 [C*C + 2], pending counts not yet folded into confusion_matrix
        self._local = np.zeros((self.num_class,) * 2)   # counts of THIS rank folded since the last all-reduce

    # ---- metrics: the reference's expressions, verbatim semantics (metrics.py:9-38)
    def Pixel_Accuracy(self):
        cm = self._matrix()
        return np.diag(cm).sum() / cm.sum()

    def Pixel_Accuracy_Class(self):
        cm = self._matrix()
        return np.diag(cm) / cm.sum(axis=1)

    def Mean_Intersection_over_Union(self):
        cm = self._matrix()
        iu = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))
        return np.nanmean(iu)

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self._matrix()
        freq = np.sum(cm, axis=1) / np.sum(cm)
        iu = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))
        return (freq[freq > 0] * iu[freq > 0]).sum()

    # ---- counting
    def _pending(self, device):
        C2 = self.num_class * self.num_class
        if self._counts is None or self._counts.device != device:
            self.sync()
            self._counts = torch.zeros(C2 + 2, dtype=torch.int64, device=device)
        return self._counts

    def sync(self, group=None, all_reduce=False):
        """Fold the device-side counts into `confusion_matrix` (one small D2H copy).  With all_reduce=True the
        counts are first summed over the data-parallel ranks (one 8*(C*C+2)-byte all-reduce); a rank with nothing
        pending - an empty validation shard, counts already folded by a metric call or a numpy add_batch - still
        takes part, with zeros, so the collective is matched on every rank."""
        from snws import ops
        reduce_now = bool(all_reduce) and ops.dp_collectives(group)
        if self._counts is None and not reduce_now:
            return
        c = self._counts
        C2 = self.num_class * self.num_class
        if reduce_now:
            if c is None:
                c = torch.zeros(C2 + 2, dtype=torch.int64, device=_device())
            # what this rank folded locally since the last reduction (metric calls, numpy add_batch) joins the sum too
            c[:C2] += torch.from_numpy(self._local.reshape(-1).astype(np.int64)).to(c.device)
            ops.all_reduce(c, "sum", group)
        host = c.cpu().numpy()
        self._counts = None
        if host[C2] or host[C2 + 1]:
            # the reference's bincount/reshape raises for a prediction outside the matrix (metrics.py:42-44)
            raise ValueError("Evaluator.add_batch: %d prediction(s) outside [0, num_class) for a valid label"
                             % int(host[C2] + host[C2 + 1]))
        add = host[:C2].reshape(self.num_class, self.num_class)
        if reduce_now:
            self.confusion_matrix += add - self._local      # the local part is already in the matrix
            self._local = np.zeros_like(self._local)
        else:
            self.confusion_matrix += add
            self._local += add

    def _matrix(self):
        self.sync()
        return self.confusion_matrix

    def add_batch(self, gt_image, pre_image):
        assert gt_image.shape == pre_image.shape
        gt = _to_dev(gt_image, _GT, "label")
        pr = _to_dev(pre_image, _PRED, "prediction")
        if pr.device != gt.device:
            pr = pr.to(gt.device)
        counts = self._pending(gt.device)
        with torch.cuda.device(gt.device):
            L.call("snws_confusion_hist", ctypes.c_void_p(gt.data_ptr()), _GT[gt.dtype], ctypes.c_void_p(pr.data_ptr()),
                   _PRED[pr.dtype], gt.numel(), self.num_class, ctypes.c_void_p(counts.data_ptr()),
                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if isinstance(gt_image, np.ndarray):
            self.sync()     # numpy callers read .confusion_matrix right away, like with the reference

    def add_batch_logits(self, gt_image, logits):
        """gt (N,H,W); logits (N,C,H,W) float32/bfloat16 GPU tensor (any memory format)."""
        assert tuple(gt_image.shape) == (logits.shape[0],) + tuple(logits.shape[2:])
        if logits.shape[1] != self.num_class:
            raise ValueError("logits have %d channels, Evaluator has %d classes" % (logits.shape[1], self.num_class))
        gt = _to_dev(gt_image, _GT, "label")
        lg = logits.detach()
        if lg.dtype not in (torch.float32, torch.bfloat16):
            lg = lg.float()
        lg = lg.permute(0, 2, 3, 1).contiguous()        # no copy when the network output is channels-last
        if gt.device != lg.device:
            gt = gt.to(lg.device)
        counts = self._pending(lg.device)
        code = L.SNWS_F32 if lg.dtype == torch.float32 else L.SNWS_BF16
        with torch.cuda.device(lg.device):
            L.call("snws_confusion_hist_argmax", ctypes.c_void_p(gt.data_ptr()), _GT[gt.dtype],
                   ctypes.c_void_p(lg.data_ptr()), code, self.num_class, gt.numel(), self.num_class,
                   ctypes.c_void_p(counts.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def reset(self):
        self._counts = None
        self._local = np.zeros((self.num_class,) * 2)
        self.confusion_matrix = np.zeros((self.num_class,) * 2)
