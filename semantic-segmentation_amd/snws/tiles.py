"""Full-disk tiled inference on the device: tile offsets, batched eval-mode forward, arg-max and stitch, sharded
round-robin over the ranks of a data-parallel job.

Mirrors the reference's flow for one FY-4A disk (predict.py:122-130 builds the tile list with
utils/gen_tiles_offs.py:3-27, predict.py:183-215 runs the network tile by tile and takes torch.max over the class
axis, predict.py:248-253 writes every tile's prediction into the full array in list order, so a later tile
overwrites an earlier one where they overlap).  Here the disk stays in HBM, tiles are cut by pointer arithmetic,
several tiles go through the network per forward, and the arg-max lands directly in the stitched prediction.
"""
import torch
import torch.distributed as dist

from . import ops


def gen_tiles_offs(xsize, ysize, block, overlap):
    """[(xoff, yoff), ...] covering an xsize x ysize raster with block x block tiles that advance by block-overlap;
    the last tile of each axis is pulled back to end at the border (and dropped if it repeats the one before).
    Order: x outer, y inner - the order tiles are stitched in (utils/gen_tiles_offs.py:3-27)."""
    step = block - overlap
    if block <= 0 or step <= 0:
        raise ValueError("gen_tiles_offs: need block > overlap >= 0")

    def axis(size):
        n = int((size - block) / step) + 1
        offs = [step * j for j in range(n)] + [size - block]
        if len(offs) >= 2 and offs[-1] == offs[-2]:
            offs.pop()
        return offs

    return [(x, y) for x in axis(xsize) for y in axis(ysize)]


@torch.no_grad()
def predict_disk(model, disk, block=512, overlap=0, n_class=None, tiles_per_forward=12, remap_background=False,
                 group=None, normalize="minmax"):
    """Class map (H, W) int64 of one (C, H, W) disk that already lives on the device.

    model: a network of this package in eval mode semantics (model.eval() is called); disk: float tensor (C, H, W),
    H = ysize, W = xsize.  normalize: what is applied before the network -
    "minmax" (default): to each (n, C, block, block) tile batch, the per-tile min-max x 255 the TRAINING tiles went
    through (dataloaders/datasets/cloud_dataset.py:127-131), on the HIP kernel of snws.data - the network then sees
    what it was trained on; "minmax_disk": one min-max x 255 over the whole disk before tiling, which is what the
    reference's own inference dataset does with its image file (utils/preddataset.py:42-56); a callable applied to
    each tile batch; or None (the disk is already scaled).  A constant tile (max == min: an all-space corner block)
    scales to 0/0 = NaN exactly as the numpy lines do; NaNs are set to 0 afterwards as utils/preddataset.py:71 does,
    so such a tile gets the class of an all-zero input instead of an arg-max over NaN logits.
    Under torch.distributed the tile list is dealt round-robin to the ranks; every rank returns the full map: the
    ranks' partial maps are merged with one all-reduce(max) of (tile index, class) keys, which reproduces "later
    tile wins".  No halo exchange is needed: every tile is a self-contained forward."""
    ops.require_gpu(disk, "predict_disk")
    if disk.dim() != 3:
        raise ValueError("predict_disk: disk must be (C, H, W)")
    C, H, W = disk.shape
    if H < block or W < block:
        raise ValueError("predict_disk: disk %dx%d is smaller than one %d tile" % (H, W, block))
    offs = gen_tiles_offs(W, H, block, overlap)
    world, rank = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    merge = ops.dp_collectives(group)
    if n_class is None:
        head = getattr(model, "cls_head", None) or getattr(model, "output_seg_map")
        n_class = head.out_channels
    was_training = model.training
    raw = hasattr(model, "raw_logits")      # take the head kernel's own output: no float32 copy of the logits
    model.eval()
    if raw:
        model.raw_logits = True
    cd = torch.bfloat16 if getattr(model, "compute_dtype", None) == torch.bfloat16 else torch.float32
    if normalize == "minmax_disk":
        from . import data
        disk = torch.nan_to_num_(data.normalize_tiles(disk, cd), nan=0.0)
    mine = [(i, o) for i, o in enumerate(offs) if i % world == rank]
    pred = torch.zeros((H, W), dtype=torch.int64, device=disk.device)
    tile_pred = torch.empty((block, block), dtype=torch.int64, device=disk.device) if merge else None
    try:
        for b0 in range(0, len(mine), tiles_per_forward):
            chunk = mine[b0:b0 + tiles_per_forward]
            batch = torch.stack([disk[:, y:y + block, x:x + block] for _, (x, y) in chunk])
            if normalize == "minmax":
                from . import data
                batch = torch.nan_to_num_(data.normalize_tiles(batch, cd), nan=0.0)
            elif normalize == "minmax_disk":
                pass
            elif normalize is not None:
                batch = normalize(batch)
            logits = model(batch)
            for j, (i, (x, y)) in enumerate(chunk):
                if not merge:       # list order: a later tile overwrites an earlier one
                    ops.argmax_stitch(logits[j:j + 1], n_class, pred, y, x, remap_background)
                else:               # key = (tile index + 1) * 256 + class: max over ranks = the LAST tile in list order
                    ops.argmax_stitch(logits[j:j + 1], n_class, tile_pred, 0, 0, remap_background)
                    region = pred[y:y + block, x:x + block]
                    torch.maximum(region, tile_pred + (i + 1) * 256, out=region)
    finally:
        model.train(was_training)
        if raw:
            model.raw_logits = False
    if merge:
        ops.all_reduce(pred, "max", group)
        pred = pred & 255
    return pred
