#!/usr/bin/env python3
"""Where the workgroups of the 3x3 forward kernel spend their time (diagnostics build only: compile csrc/conv_mfma.hip
with -DSNWS_TIMING).  Prints, per shape, the average shader-clock cycles a workgroup waits for staged data, multiplies
and spends in the epilogue.   SNWS_PRIO=0/1 python tools/conv_phase_probe.py"""
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
    sys.exit("this probe needs the diagnostics build: rebuild csrc/conv_mfma.hip with -DSNWS_TIMING "
             "(make FLAGS=\"... -DSNWS_TIMING\"), run it, then rebuild the shipped library without the flag")
out = (ctypes.c_ulonglong * 16)()
for cin, cout, S in ((320, 320, 512), (320, 320, 256), (64, 64, 512), (128, 128, 256), (512, 512, 64)):
    x = torch.randn(N, S, S, cin, device="cuda:0").to(torch.bfloat16)
    y = torch.empty(N, S, S, cout, device="cuda:0", dtype=torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device="cuda:0") * 0.05
    cfg = ops.ConvCfg((cout, cin, 3, 3), 1, 1, 1, 1)
    pw = ops.PreparedWeight(w, cfg, True)
    d = cfg.desc(N, S, S, cin, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        L.call("snws_conv_mfma_fwd", ctypes.byref(d), x.data_ptr(), pw.pack_fwd.data_ptr(), None, y.data_ptr(), None, st)
    for _ in range(2):
        run()
    lib.snws_dbg_read(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    lib.snws_dbg_read(out)
    n = max(out[3], 1)
    fl = 2.0 * 9 * N * S * S * cin * cout
    print("fwd 3x3 %4d->%4d @%3d: %.3f ms %5.0f TFLOP/s | per workgroup: wait %7.0f  matrix %7.0f  epilogue %7.0f cycles (%d chunks)"
          % (cin, cout, S, ms, fl / ms / 1e9, out[0] / n, out[1] / n, out[2] / n, cin // 32), flush=True)

# ---- 3x3 weight gradient: cycles per pixel tile waiting for the staged tiles / multiplying
for cin, cout, S in ((320, 320, 512), (320, 320, 256), (64, 64, 512), (128, 128, 256), (512, 512, 64)):
    x = torch.randn(N, S, S, cin, device="cuda:0").to(torch.bfloat16)
    dy = torch.randn(N, S, S, cout, device="cuda:0").to(torch.bfloat16)
    g = torch.zeros(cout, 3, 3, cin, device="cuda:0")
    db = torch.zeros(cout, device="cuda:0")
    cfg = ops.ConvCfg((cout, cin, 3, 3), 1, 1, 1, 1)
    d = cfg.desc(N, S, S, cin, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = lib.snws_conv_mfma_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes // 4, device="cuda:0")

    def run():
        L.call("snws_conv_mfma_wgrad_det", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(), db.data_ptr(),
               ws.data_ptr(), nbytes, st)
    for _ in range(2):
        run()
    lib.snws_dbg_read(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    lib.snws_dbg_read(out)
    nt = max(out[6], 1)
    fl = 2.0 * 9 * N * S * S * cin * cout
    print("wgrad 3x3 %4d->%4d @%3d: %.3f ms %5.0f TFLOP/s | per pixel tile: wait %6.0f  matrix %6.0f cycles (%.1f tiles per workgroup)"
          % (cin, cout, S, ms, fl / ms / 1e9, out[4] / nt, out[5] / nt, nt / max(out[7], 1)), flush=True)
