"""Throughput of the wide 1x1 kernel (snws_gemm1x1) on the shapes the models use, beside torch.mm (vendor GEMM)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
from snws import _lib as L  # noqa: E402

SHAPES = [
    # M, N, K, what
    (16 * 512 * 512, 320, 320, "ocr aux_head"),
    (16 * 512 * 512, 256, 512, "ocr f_pixel[0] / f_up dgrad"),
    (16 * 512 * 512, 256, 256, "ocr f_pixel[1]"),
    (16 * 512 * 512, 512, 256, "ocr f_up / f_pixel[0] dgrad"),
    (16 * 512 * 512, 512, 1024, "ocr conv_bn_dropout"),
    (16 * 512 * 512, 1024, 512, "ocr conv_bn_dropout dgrad"),
    (16 * 256 * 256, 576, 128, "taps convd2_cat"),
    (16 * 128 * 128, 576, 256, "taps convd3_cat"),
    (16 * 64 * 64, 576, 512, "taps convd4_cat"),
    (16 * 32 * 32, 576, 1024, "taps conv5_cat"),
    (16 * 256 * 256, 576, 320, "taps convu_cat @256"),
    (16 * 128 * 128, 576, 320, "taps convu_cat @128"),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda:0"
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    only = sys.argv[1:] and int(sys.argv[1])
    for i, (M, N, K, what) in enumerate(SHAPES):
        if sys.argv[1:] and i != only:
            continue
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        y2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_own = timeit(lambda: L.call("snws_gemm1x1", x.data_ptr(), K, w.data_ptr(), None, y.data_ptr(), N, M, N, K, st))
        t_lib = timeit(lambda: torch.mm(x, w.t(), out=y2))
        err = float((y.float() - y2.float()).norm() / y2.float().norm())
        fl = 2.0 * M * N * K
        by = 2.0 * M * (N + K)
        print("%-28s M=%8d N=%4d K=%4d  own %7.3f ms %6.0f TF %5.2f TB/s | torch.mm %7.3f ms %6.0f TF | diff %.1e"
              % (what, M, N, K, t_own, fl / t_own / 1e9, by / t_own / 1e9, t_lib, fl / t_lib / 1e9, err), flush=True)
        del x, w, y, y2


if __name__ == "__main__":
    main()
