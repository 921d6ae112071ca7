#!/usr/bin/env python3
"""Where the phases of the ping-pong 3x3 wgrad kernel go (diagnostics build only: csrc/conv_mfma.hip compiled with
-DSNWS_TIMING into lib/libsnws_timing.so; run as SNWS_LIB=libsnws_timing.so python tools/wgrad9p_phase_probe.py).
Per pixel tile: shader-clock cycles group A multiplies, waits at the barrier behind its matrix phase, and spends in
the other phase; cycles group B needs to issue its copies, waits for them to land, and waits at the barrier."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from snws import _lib as L
from snws import ops

N = 16
lib = L.lib()
if not hasattr(lib, "snws_dbg_read"):
    sys.exit("needs the -DSNWS_TIMING build")
out = (ctypes.c_ulonglong * 16)()
for cin, cout, S in ((320, 320, 512), (320, 64, 512), (64, 64, 512), (128, 128, 256), (512, 512, 64)):
    x = torch.randn(N, S, S, cin, device="cuda:0").to(torch.bfloat16)
    dy = torch.randn(N, S, S, cout, device="cuda:0").to(torch.bfloat16)
    g = torch.zeros(cout, 3, 3, cin, device="cuda:0")
    cfg = ops.ConvCfg((cout, cin, 3, 3), 1, 1, 1, 1)
    d = cfg.desc(N, S, S, cin, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = lib.snws_conv_mfma_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes // 4, device="cuda:0")

    def run():
        L.call("snws_conv_mfma_wgrad_det", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(), None,
               ws.data_ptr(), nbytes, st)
    run()
    lib.snws_dbg_read(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    lib.snws_dbg_read(out)
    n = max(out[13], 1)
    fl = 2.0 * 9 * N * S * S * cin * cout
    print("wgrad9p %4d->%4d @%3d: %.3f ms %5.0f TFLOP/s | per tile: A multiply %6.0f, A at barrier %6.0f, A other phase %6.0f | "
          "B issue %6.0f, B wait copies %6.0f, B at barrier %6.0f cycles"
          % (cin, cout, S, ms, fl / ms / 1e9, out[8] / n, out[9] / n, out[14] / n, out[10] / n, out[11] / n, out[12] / n), flush=True)
