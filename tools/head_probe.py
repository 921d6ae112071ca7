"""Per-kernel timing of the last block + classification head at the bench shape, fused and unfused."""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
from modeling import blocks  # noqa: E402
from snws import ops  # noqa: E402


def main():
    N, C, H, W, ncls = 16, 320, 512, 512, 10
    if len(sys.argv) > 1:
        C = int(sys.argv[1])
    dev = "cuda:0"
    torch.manual_seed(0)
    blk = blocks.ConvSnRelu(C, C).to(dev).train()
    head = torch.nn.Conv2d(C, ncls, 1).to(dev)
    x = torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dl = torch.randn(N, 16, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dl[:, ncls:] = 0
    for fused in (True, False):
        for it in range(4):
            ops.PROFILE = [] if it == 3 else None
            xg = x.detach().requires_grad_(True)
            raw = blk.with_head(xg, head) if fused else blocks.head_conv(head, blk(xg), raw=True)
            raw.backward(dl)
            torch.cuda.synchronize()
        agg = collections.OrderedDict()
        for tag, work, e0, e1 in ops.PROFILE:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(tag, [0.0, 0.0, 0])
            a[0] += ms
            a[1] += work
            a[2] += 1
        ops.PROFILE = None
        print("fused" if fused else "unfused")
        tot = 0.0
        for tag, (ms, work, n) in agg.items():
            tot += ms
            print("  %-24s n=%d %8.3f ms   %8.1f G(work)/s" % (tag, n, ms, work / ms / 1e6))
        print("  total profiled %.3f ms" % tot)


if __name__ == "__main__":
    main()
