#!/usr/bin/env python3
"""Per-parameter gradient difference between the fused (dgrad-epilogue) and the stand-alone SwitchNorm backward
reduction, plus run-to-run difference of the unfused path as the noise floor."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch
from modeling.UNet3p_SNws import UNet3p_SNws
from snws import ops

DEV = torch.device("cuda", 0)
torch.manual_seed(11)
model = UNet3p_SNws(14, 64, 10, 1, 1).to(DEV).train()
model.compute_dtype = torch.bfloat16
img = (torch.rand(2, 14, 64, 64) * 255.0).to(DEV)
lab = torch.randint(0, 10, (2, 64, 64)).to(DEV)


def run(fuse):
    ops.FUSE_SN_BWD = fuse
    ops.FUSED_SN_BWD_HITS = 0
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model(img), lab) / 2
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters()}, ops.FUSED_SN_BWD_HITS


a, ha = run(False)
b, hb = run(True)
c, hc = run(False)
print("hits", ha, hb, hc)
rows = []
for n in a:
    d = float((a[n].double() - b[n].double()).norm() / max(float(a[n].double().norm()), 1e-30))
    d0 = float((a[n].double() - c[n].double()).norm() / max(float(a[n].double().norm()), 1e-30))
    rows.append((d, d0, n, float(a[n].norm())))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3e  (unfused rerun %.3e)  %s  |g|=%.3e" % r)

# ---- per layer: the epilogue's partial sums against the stand-alone reduce kernel on the same gradient
import ctypes
from snws import _lib as L
orig = ops.fused_bwd_partials


def checked(s, dys, lddy):
    r = orig(s, dys, lddy)
    if r is None:
        return r
    part, T = r
    N, C, H, W = s["shape"]
    lib = L.lib()
    Tr = lib.snws_sn_stats_tiles(N, H * W, C)
    ref = torch.empty((N, Tr, C, 2), dtype=torch.float32, device=dys.device)
    L.call("snws_sn_bwd_reduce", dys.data_ptr(), lddy, s["xs"].data_ptr(), s["ldx"], None, 0, L.SNWS_BF16,
           s["ctx"].data_ptr(), s["gamma"].data_ptr(), s["beta"].data_ptr(), N, H * W, C, int(s["relu"]), ref.data_ptr(),
           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    a, b = part.double().sum(1), ref.double().sum(1)
    print("layer N%d C%d %dx%d ld %d: max |diff| %.3e of max %.3e  (P %.3e, Q %.3e)" % (
        N, C, H, W, lddy, float((a - b).abs().max()), float(b.abs().max()),
        float((a - b)[..., 0].abs().max()), float((a - b)[..., 1].abs().max())))
    return r


ops.fused_bwd_partials = checked
run(True)

# ---- layer by layer: dy into and dx out of every SwitchNorm backward, unfused vs fused, in execution order
ops.fused_bwd_partials = orig
orig_snb = ops.sn_backward
trace = []


def traced(dy, s, *a, **k):
    hit0 = ops.FUSED_SN_BWD_HITS
    r = orig_snb(dy, s, *a, **k)
    trace.append((tuple(s["shape"]), dy.detach().float().clone(), r[0].detach().float().clone(), ops.FUSED_SN_BWD_HITS - hit0,
                  [t.detach().float().clone() for t in r[1:5]]))
    return r


ops.sn_backward = traced
run(False)
t0, trace = trace, []
ops.sn_backward = traced
run(True)
t1 = trace
for (sh, dy0, dx0, _, g0), (sh1, dy1, dx1, hit, g1) in zip(t0, t1):
    rel = lambda a, b: float((a - b).norm() / max(float(a.norm()), 1e-30))
    print("SN %s fused=%d  dy diff %.2e  dx diff %.2e  dgamma %.2e dbeta %.2e dmw %.2e dvw %.2e" % (
        sh, hit, rel(dy0, dy1), rel(dx0, dx1), rel(g0[0], g1[0]), rel(g0[1], g1[1]), rel(g0[2], g1[2]), rel(g0[3], g1[3])))
