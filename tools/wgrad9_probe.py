#!/usr/bin/env python3
"""Times snws_conv_mfma_wgrad on the 3x3 shapes of UNet3p_SNws (batch 16, 512^2 input) that carry most of its
weight-gradient work.  SNWS_WGRAD_BLOCKS=n (read once per process) forces the workgroup count of every launch:
    for n in 0 512 1024 2048 4096 8192; do SNWS_WGRAD_BLOCKS=$n python tools/wgrad9_probe.py; done"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from snws import _lib as L
from snws import ops

N = 16
forced = os.environ.get("SNWS_WGRAD_BLOCKS", "0")
for cin, cout, S in ((320, 320, 512), (320, 64, 512), (64, 64, 512), (320, 320, 256), (320, 64, 256), (128, 128, 256),
                     (64, 128, 256), (128, 64, 256), (256, 256, 128), (320, 320, 128), (512, 512, 64), (1024, 1024, 32)):
    x = torch.randn(N, S, S, cin, device="cuda:0").to(torch.bfloat16)
    dy = torch.randn(N, S, S, cout, device="cuda:0").to(torch.bfloat16)
    g = torch.zeros(cout, 3, 3, cin, device="cuda:0")
    # no bias gradient, as in the training step (inside a Conv2dws -> SwitchNorm block it comes from the SwitchNorm
    # backward in closed form); SNWS_PROBE_BIAS=1 asks for it
    db = torch.zeros(cout, device="cuda:0") if os.environ.get("SNWS_PROBE_BIAS") == "1" else None
    cfg = ops.ConvCfg((cout, cin, 3, 3), 1, 1, 1, 1)
    d = cfg.desc(N, S, S, cin, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    nbytes = L.lib().snws_conv_mfma_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes // 4, device="cuda:0")

    def run_atomic():
        L.call("snws_conv_mfma_wgrad", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(), db.data_ptr() if db is not None else None, st)

    def run_det():      # partial tiles + fixed-order reduction (both kernels are inside the timed region)
        L.call("snws_conv_mfma_wgrad_det", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(), db.data_ptr() if db is not None else None,
               ws.data_ptr(), nbytes, st)
    res = []
    for run in (run_atomic, run_det):
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * 9 * N * S * S * cin * cout
    print("blocks %5s  wgrad 3x3 %4d->%4d @%3d^2: atomic %.3f ms %6.0f TFLOP/s | partial tiles + reduce %.3f ms %6.0f TFLOP/s "
          "(workspace %.0f MB)" % (forced, cin, cout, S, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9, nbytes / 1e6),
          flush=True)
    del x, dy, g, ws
