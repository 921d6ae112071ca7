"""Input-side tile preparation on the device: what CLOUDdataset.__getitem__ does with numpy on the host
(dataloaders/datasets/cloud_dataset.py:127-134 of the reference) for tiles that already live in HBM - per-tile
min-max scaling to [0, 255] (NaN -> 0 first) and the label remap NaN / 10 / 255 -> 0 - bit-identical to the numpy
arithmetic.  There is no CPU path."""
import ctypes

import torch

from . import _lib as L
from . import ops

_SRC = {torch.float32: L.SNWS_F32, torch.float64: L.SNWS_F64, torch.int64: L.SNWS_I64, torch.int32: L.SNWS_I32,
        torch.uint8: L.SNWS_U8}


def normalize_tiles(tiles, out_dtype=torch.float32, return_minmax=False):
    """tiles: (n, C, H, W) or (C, H, W) float32 GPU tensor, every tile contiguous.  Returns a tensor of the same shape
    and layout holding ((x - min) / (max - min)) * 255 per TILE (minimum and maximum over all of the tile's channels,
    NaN counted as 0: cloud_dataset.py:127-131), in float32 or - rounded from the float32 result - bfloat16."""
    ops.require_gpu(tiles, "normalize_tiles")
    if tiles.dim() not in (3, 4):
        raise ValueError("normalize_tiles takes (C, H, W) or (n, C, H, W), got %s" % (tuple(tiles.shape),))
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("normalize_tiles: out_dtype must be float32 or bfloat16")
    x = tiles if tiles.dtype == torch.float32 else tiles.float()
    x = x.contiguous()
    n = 1 if x.dim() == 3 else x.shape[0]
    if n == 0:
        return torch.empty_like(x, dtype=out_dtype)
    elems = x.numel() // n
    lib = L.lib()
    B = lib.snws_tile_minmax_blocks(elems)
    part = torch.empty(2 * n * B, dtype=torch.float32, device=x.device)
    mm = torch.empty((n, 2), dtype=torch.float32, device=x.device) if return_minmax else None
    y = torch.empty_like(x, dtype=out_dtype)
    with torch.cuda.device(x.device):
        L.call("snws_tile_minmax_scale", x.data_ptr(), y.data_ptr(), L.SNWS_F32 if out_dtype == torch.float32 else L.SNWS_BF16,
               n, elems, part.data_ptr(), None if mm is None else mm.data_ptr(), ops._stream())
    return (y, mm) if return_minmax else y


def remap_labels(mask):
    """Label map of any shape (float32 / float64 / int64 / int32 / uint8 GPU tensor) -> float32 labels with NaN, 10
    and 255 mapped to 0 (cloud_dataset.py:132-137)."""
    ops.require_gpu(mask, "remap_labels")
    if mask.dtype not in _SRC:
        raise TypeError("remap_labels: unsupported label dtype %s" % mask.dtype)
    m = mask.contiguous()
    out = torch.empty(m.shape, dtype=torch.float32, device=m.device)
    with torch.cuda.device(m.device):
        L.call("snws_label_remap", m.data_ptr(), _SRC[m.dtype], out.data_ptr(), m.numel(), ops._stream())
    return out
