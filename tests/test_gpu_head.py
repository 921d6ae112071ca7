"""GPU parity tests of the fused last block + classification head (ConvSnRelu -> 1x1 nn.Conv2d,
modeling/UNet3p_SNws.py:93,158 and modeling/UNet_SNws.py:31,76 of the reference) against the unfused path through
the same library and against the float64 oracle."""
import os
import statistics

import pytest
import torch

from oracle import snws_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def l2err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def maxerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _block_and_head(cin, c, ncls, bias, seed):
    from modeling.blocks import ConvSnRelu
    torch.manual_seed(seed)
    blk = ConvSnRelu(cin, c).to(DEV).train()
    head = torch.nn.Conv2d(c, ncls, 1, bias=bias).to(DEV)
    with torch.no_grad():
        blk[1].weight.copy_(torch.rand(1, c, 1, 1) + 0.5)
        blk[1].bias.copy_(torch.randn(1, c, 1, 1) * 0.3)
        blk[1].mean_weight.copy_(torch.randn(3))
        blk[1].var_weight.copy_(torch.randn(3))
        head.weight.copy_(head.weight.to(torch.bfloat16).float())     # the head filter as the kernels hold it
    return blk, head


PARAMS = ["0.weight", "0.bias", "1.weight", "1.bias", "1.mean_weight", "1.var_weight"]


def _run(blk, head, x, dl, fused):
    from modeling import blocks
    for m in (blk, head):
        m.zero_grad(set_to_none=True)
    rm, rv = blk[1].running_mean.clone(), blk[1].running_var.clone()
    xg = x.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ncls = head.out_channels
    if fused:
        assert blk.head_fusable(xg, head)
        raw = blk.with_head(xg, head)
        assert raw.shape[1] == 16 and raw.dtype == torch.bfloat16
        if ncls < 16:
            assert float(raw.detach()[:, ncls:].float().abs().max()) == 0.0      # padding channels are exact zeros
        logits = blocks.finish_logits(raw, ncls)
    else:
        logits = blocks.head_conv(head, blk(xg))
    logits.backward(dl)
    out = dict(logits=logits.detach().float(), dx=xg.grad.float(), hw=head.weight.grad.float().clone(),
               rm=blk[1].running_mean.clone(), rv=blk[1].running_var.clone())
    if head.bias is not None:
        out["hb"] = head.bias.grad.float().clone()
    for n in PARAMS:
        out[n] = blk.get_parameter(n).grad.float().clone()
    with torch.no_grad():       # both runs start from the same running statistics
        blk[1].running_mean.copy_(rm)
        blk[1].running_var.copy_(rv)
    return out


CASES = [
    # N, Cin, C, H, W, classes, head bias, compare with the float64 oracle too
    (2, 320, 320, 32, 40, 10, True, True),
    (1, 64, 64, 24, 24, 2, True, True),
    (2, 128, 64, 9, 11, 16, False, True),          # ragged: 99 positions per image
    (1, 320, 320, 100, 100, 10, True, False),      # more than one backward tile per image, last tile ragged
    (3, 64, 64, 96, 96, 4, True, False),
]


@pytest.mark.parametrize("case", CASES)
def test_fused_block_and_head_matches_unfused_path_and_oracle(case):
    """Logits and every gradient (input, filter, conv bias, the four SwitchNorm parameters, head filter and bias) of
    the fused op against the unfused ConvSnRelu -> head_conv path of the same library, relative L2 1e-2 (both round
    the same tensors to bf16; the fused backward keeps W^T dlogits in float32 where the unfused one stores it as
    bf16), and against the float64 oracle with the bf16 storage roundings, bar 2e-2 (4e-2 for the mixing weights)."""
    N, cin, c, H, W, ncls, bias, with_oracle = case
    blk, head = _block_and_head(cin, c, ncls, bias, 5)
    x = (torch.randn(N, cin, H, W) * 1.5 + 0.3).to(torch.bfloat16)
    dl = torch.randn(N, ncls, H, W).to(torch.bfloat16).float()
    a = _run(blk, head, x.to(DEV), dl.to(DEV), fused=True)
    b = _run(blk, head, x.to(DEV), dl.to(DEV), fused=False)
    for k in b:
        tol = 3e-2 if k in ("1.mean_weight", "1.var_weight") else 1e-2
        assert l2err(a[k], b[k]) < tol, (k, l2err(a[k], b[k]))
    assert maxerr(a["rm"], b["rm"]) < 1e-6 and maxerr(a["rv"], b["rv"]) < 1e-6
    if not with_oracle:
        return
    rnd = lambda t: t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())  # noqa: E731
    w = blk[0].weight.detach().cpu().double().requires_grad_(True)
    cb = blk[0].bias.detach().cpu().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    conv = rnd(torch.nn.functional.conv2d(xr, rnd(O.ws_standardize(w)), cb, 1, 1))
    st = O.SNState(c, dtype=torch.float64)
    st.gamma = blk[1].weight.detach().cpu().double().view(-1).requires_grad_(True)
    st.beta = blk[1].bias.detach().cpu().double().view(-1).requires_grad_(True)
    st.mean_weight = blk[1].mean_weight.detach().cpu().double().requires_grad_(True)
    st.var_weight = blk[1].var_weight.detach().cpu().double().requires_grad_(True)
    yr, _ = O.switchnorm_fwd(conv, st, True, relu=True)
    hw = head.weight.detach().cpu().double().requires_grad_(True)
    hb = None if head.bias is None else head.bias.detach().cpu().double().requires_grad_(True)
    lr = torch.nn.functional.conv2d(rnd(yr), hw, hb)
    lr.backward(dl.double())
    assert maxerr(a["logits"], lr) < 2e-2
    assert maxerr(a["dx"], xr.grad) < 2e-2
    assert maxerr(a["0.weight"], w.grad) < 2e-2
    # the conv bias only acts through what the normalisation does not remove: a small, cancelling sum
    assert maxerr(a["0.bias"], cb.grad) < 5e-2
    assert maxerr(a["1.weight"].view(-1), st.gamma.grad) < 2e-2 and maxerr(a["1.bias"].view(-1), st.beta.grad) < 2e-2
    assert maxerr(a["1.mean_weight"], st.mean_weight.grad) < 4e-2
    assert maxerr(a["1.var_weight"], st.var_weight.grad) < 4e-2
    assert maxerr(a["hw"], hw.grad) < 2e-2
    if hb is not None:
        assert maxerr(a["hb"], hb.grad) < 1e-3


def test_fused_head_is_deterministic():
    """Two runs of the fused op on the same inputs give bit-identical logits and gradients (per-workgroup partials
    summed in a fixed order, no float atomics)."""
    blk, head = _block_and_head(320, 320, 10, True, 9)
    x = (torch.randn(2, 320, 48, 48) * 1.5).to(torch.bfloat16).to(DEV)
    dl = torch.randn(2, 10, 48, 48).to(DEV)
    a = _run(blk, head, x, dl, fused=True)
    b = _run(blk, head, x, dl, fused=True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_unet3p_fused_head_on_and_off():
    """UNet3p_SNws(n_filters=64), bf16 tier: the model takes the fused op by default (kernel launch counted), the
    logits agree with the unfused run to bf16 rounding, and the parameter gradients agree as far as this network's
    amplification of a rounding-level change allows (median 2e-2, 90th percentile 6e-2)."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import ops
    torch.manual_seed(3)
    model = UNet3p_SNws(14, 64, 10, 1, 1).to(DEV).train()
    model.compute_dtype = torch.bfloat16
    img = (torch.rand(2, 14, 64, 64) * 255.0).to(DEV)
    lab = torch.randint(0, 10, (2, 64, 64)).to(DEV)
    res = {}
    saved = ops.HEAD_FUSION
    state = {k: v.clone() for k, v in model.state_dict().items()}
    try:
        for on in (True, False):
            ops.HEAD_FUSION = on
            model.load_state_dict(state)
            model.zero_grad(set_to_none=True)
            ops.PROFILE = []
            try:
                logits = model(img)
                launched = any(rec[0] == "sn_apply_head" for rec in ops.PROFILE)
            finally:
                ops.PROFILE = None
            assert launched == on
            assert logits.shape == (2, 10, 64, 64) and logits.dtype == torch.float32
            (torch.nn.functional.cross_entropy(logits, lab) / 2).backward()
            res[on] = (logits.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()})
    finally:
        ops.HEAD_FUSION = saved
    assert l2err(res[True][0], res[False][0]) < 1e-2
    errs = {n: l2err(res[True][1][n], res[False][1][n]) for n in res[True][1]}
    worst = max(errs, key=errs.get)
    ranked = sorted(errs.values())
    print("fused head on/off: logits %.2e, gradients median %.2e, 90th percentile %.2e, worst %.2e (%s)"
          % (l2err(res[True][0], res[False][0]), statistics.median(ranked), ranked[int(0.9 * len(ranked))], errs[worst], worst))
    # an untrained 34-block bf16 network amplifies a rounding-level change of one layer's gradient by several percent
    # in the most sensitive (smallest) parameter gradients; the bulk must agree closely
    assert statistics.median(ranked) < 2e-2 and ranked[int(0.9 * len(ranked))] < 6e-2 and errs[worst] < 0.35, worst


def test_head_entry_points_reject_unsupported_shapes():
    from snws import _lib as L
    lib = L.lib()
    x = torch.zeros(1, 8, 8, 128, dtype=torch.bfloat16, device=DEV)
    f = torch.zeros(4096, dtype=torch.float32, device=DEV)
    w = torch.zeros(16, 128, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device=DEV)
    rc = lib.snws_sn_apply_head(x.data_ptr(), 128, f.data_ptr(), f.data_ptr(), w.data_ptr(), None, out.data_ptr(), 1, 64,
                                128, 1, None)
    assert rc == L.ERR_UNSUPPORTED       # 128 channels: not a built variant
    # backward partial rows per image: 2048-pixel runs, x4 pixel groups at C = 64
    assert lib.snws_sn_head_tiles(2048, 320, 1) == 1 and lib.snws_sn_head_tiles(2049, 320, 1) == 2
    assert lib.snws_sn_head_tiles(2048, 64, 1) == 4 and lib.snws_sn_head_tiles(2049, 320, 0) == 2


# ================================================================== shared gradients of multi-consumer tensors
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)])
def test_shared_gradient_slots_match_autograd_accumulation(dtype, tol):
    """A tensor consumed by two fused conv blocks, a max-pool and a tap convolution + three up-sampled consumers of its
    tap map: with ops.share_grad the consumers add their gradients in place (dgrad epilogue, pooling backward, the
    k = 2 up-sampled-conv backward's own epilogue in bf16, snws_add_inplace for the others); the input gradient and every parameter gradient must equal what autograd's own accumulation
    gives (float32: to rounding of the summation order; bf16: one bf16 rounding per partial sum differs)."""
    from modeling.blocks import ConvSnRelu
    from snws import ops
    torch.manual_seed(21)
    N, C, H, W = 2, 64, 32, 32
    prod = ConvSnRelu(C, C).to(DEV).train()
    c1, c2, ct = ConvSnRelu(C, C).to(DEV).train(), ConvSnRelu(C, 2 * C).to(DEV).train(), ConvSnRelu(C, C).to(DEV).train()
    mods = [prod, c1, c2, ct]
    x = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    g = [torch.randn(N, C, H, W, device=DEV), torch.randn(N, 2 * C, H, W, device=DEV), torch.randn(N, C, H // 2, W // 2, device=DEV),
         torch.randn(N, C, 2 * H, 2 * W, device=DEV), torch.randn(N, C, 4 * H, 4 * W, device=DEV),
         torch.randn(N, C, 8 * H, 8 * W, device=DEV)]
    res = {}
    saved = ops.GRAD_SHARE
    try:
        for share in (True, False):
            ops.GRAD_SHARE = share
            for m in mods:
                m.zero_grad(set_to_none=True)
            xg = x.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
            ops.PROFILE = []
            t = ops.share_grad(prod(xg))
            assert (ops.grad_slot(t) is not None) == share
            z = ops.share_grad(ct.taps(t))
            outs = [c1(t), c2(t), ops.max_pool_ceil(t, 2), ct.up(z, 2), ct.up(z, 4), ct.up(z, 8)]
            loss = sum((o.float() * gi).sum() for o, gi in zip(outs, g))
            hits0 = ops.GRAD_SHARE_HITS
            loss.backward()
            tags = [r[0] for r in ops.PROFILE]
            ops.PROFILE = None
            # the tap map's consumers report in reverse order: k = 8 parks its gradient, k = 4 adds with snws_add_inplace,
            # k = 2 adds in its own kernel (bf16) or with snws_add_inplace as well (float32)
            assert ("grad_add" in tags) == share
            if share:
                assert tags.count("grad_add") == 1 if dtype == torch.bfloat16 else tags.count("grad_add") >= 2
            assert (ops.GRAD_SHARE_HITS - hits0 >= 5) == share     # 3 later consumers of t, 2 of the tap map
            res[share] = [xg.grad.float().clone()] + [p.grad.float().clone() for m in mods for p in m.parameters()]
    finally:
        ops.GRAD_SHARE = saved
        ops.PROFILE = None
    for a, b in zip(res[True], res[False]):
        assert l2err(a, b) < tol, (a.shape, l2err(a, b))


def test_shared_gradient_guard_catches_an_unaware_consumer():
    """A consumer that does not know about shared gradients (here: a plain torch op) makes autograd sum out of place;
    the identity node of share_grad notices that the gradient it receives is not the parked tensor and raises instead
    of silently losing a contribution."""
    from modeling.blocks import ConvSnRelu
    from snws import ops
    torch.manual_seed(2)
    blk, c1 = ConvSnRelu(64, 64).to(DEV).train(), ConvSnRelu(64, 64).to(DEV).train()
    x = torch.randn(1, 64, 16, 16, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    t = ops.share_grad(blk(x))
    loss = c1(t).float().sum() + (t.float() * 2).sum()
    with pytest.raises(RuntimeError, match="share_grad"):
        loss.backward()
