"""GPU parity tests of the fused OCR attention kernels (SpatialGather_Module.forward and the attention inside
_ObjectAttentionBlock.forward, modeling/UNet3p_res_ocr_SNws.py:21-28,87-100 of the reference) against the same
arithmetic written with torch ops in float64 on the same bf16 inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def l2err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("B,H,W,K,padded", [(2, 40, 50, 10, True), (1, 16, 16, 3, False), (3, 97, 11, 16, True)])
def test_ocr_gather_vs_float64(B, H, W, K, padded):
    """ctx = softmax over pixels (scale * probs) @ feats and its gradients wrt feats and probs.  Bars: the forward
    and d feats within 1e-2 (probabilities enter the product as bf16), d probs within 2e-2."""
    from snws import ops
    torch.manual_seed(B * 100 + K)
    C = ops.OCR_GATHER_C
    feats = _bf(torch.randn(B, H, W, C, device=DEV)).permute(0, 3, 1, 2)
    if padded:      # the class scores as the head kernel leaves them: 16-wide pixel rows, K real channels
        pbuf = _bf(torch.randn(B, H, W, 16, device=DEV) * 3)
        probs = pbuf.permute(0, 3, 1, 2)[:, :K]
    else:
        probs = _bf(torch.randn(B, K, H, W, device=DEV) * 3)
    dctx = _bf(torch.randn(B, C, K, 1, device=DEV))
    f1, p1 = feats.detach().requires_grad_(True), probs.detach().requires_grad_(True)
    assert ops.ocr_gather_ok(f1, p1)
    out = ops.OcrGatherFn.apply(f1, p1, 1.0)
    assert out.shape == (B, C, K, 1) and out.dtype == torch.bfloat16
    out.backward(dctx)
    f2, p2 = feats.detach().double().requires_grad_(True), probs.detach().double().requires_grad_(True)
    pr = torch.softmax(p2.reshape(B, K, -1), dim=2)
    ref = torch.matmul(pr, f2.permute(0, 2, 3, 1).reshape(B, H * W, C)).permute(0, 2, 1).unsqueeze(3)
    ref.backward(dctx.double())
    assert l2err(out, ref) < 1e-2
    assert l2err(f1.grad, f2.grad) < 1e-2
    assert l2err(p1.grad, p2.grad) < 2e-2


@pytest.mark.parametrize("B,H,W,K", [(2, 40, 50, 10), (1, 16, 16, 2), (2, 33, 31, 16)])
def test_ocr_attend_vs_float64(B, H, W, K):
    """out = softmax over classes (Ck^-0.5 * q . key) @ value and its gradients wrt q, key, value; bars 1e-2 for the
    forward, 2e-2 for the gradients (probabilities and their gradients enter the second products as bf16)."""
    from snws import ops
    torch.manual_seed(B * 10 + K)
    C = ops.OCR_ATTEND_C
    q = _bf(torch.randn(B, H, W, C, device=DEV)).permute(0, 3, 1, 2)
    key = _bf(torch.randn(B, K, 1, C, device=DEV)).permute(0, 3, 1, 2)          # (B, C, K, 1), channels-last
    val = _bf(torch.randn(B, K, 1, C, device=DEV)).permute(0, 3, 1, 2)
    dout = _bf(torch.randn(B, H, W, C, device=DEV)).permute(0, 3, 1, 2)
    scale = C ** -0.5
    a = [t.detach().requires_grad_(True) for t in (q, key, val)]
    assert ops.ocr_attend_ok(a[0], a[1])
    out = ops.OcrAttendFn.apply(a[0], a[1], a[2], scale)
    out.backward(dout)
    r = [t.detach().double().requires_grad_(True) for t in (q, key, val)]
    query = r[0].permute(0, 2, 3, 1).reshape(B, H * W, C)
    sim = torch.softmax(torch.matmul(query, r[1].reshape(B, C, K)) * scale, dim=-1)
    ref = torch.matmul(sim, r[2].reshape(B, C, K).permute(0, 2, 1)).reshape(B, H, W, C).permute(0, 3, 1, 2)
    ref.backward(dout.double())
    assert l2err(out, ref) < 1e-2
    for got, want, name in zip(a, r, ("q", "key", "value")):
        assert got.grad.shape == want.grad.shape, name
        assert l2err(got.grad, want.grad) < 2e-2, (name, l2err(got.grad, want.grad))


def test_ocr_kernels_are_deterministic_and_reject_other_widths():
    from snws import _lib as L
    from snws import ops
    torch.manual_seed(1)
    B, H, W, K, C = 2, 64, 48, 10, ops.OCR_ATTEND_C
    q = _bf(torch.randn(B, H, W, C, device=DEV)).permute(0, 3, 1, 2)
    key = _bf(torch.randn(B, K, 1, C, device=DEV)).permute(0, 3, 1, 2)
    val = _bf(torch.randn(B, K, 1, C, device=DEV)).permute(0, 3, 1, 2)
    dout = _bf(torch.randn(B, H, W, C, device=DEV)).permute(0, 3, 1, 2)
    runs = []
    for _ in range(2):
        a = [t.detach().requires_grad_(True) for t in (q, key, val)]
        out = ops.OcrAttendFn.apply(a[0], a[1], a[2], C ** -0.5)
        out.backward(dout)
        runs.append([out.detach()] + [t.grad for t in a])
    for x, y in zip(*runs):
        assert torch.equal(x, y)
    lib = L.lib()
    z = torch.zeros(4096, device=DEV, dtype=torch.bfloat16)
    rc = lib.snws_ocr_attend_fwd(z.data_ptr(), 128, z.data_ptr(), z.data_ptr(), z.data_ptr(), 128, 1, 16, 128, 4, 1.0, None)
    assert rc == L.ERR_UNSUPPORTED


def test_gemm1x1_acc_adds_into_its_destination():
    """snws_gemm1x1_acc: y <- round(y + x w^T) (f32 sum of the stored bf16 value and the f32 product, rounded once)."""
    from snws import ops
    torch.manual_seed(5)
    N, H, W, K, O = 2, 20, 13, 128, 192
    x = _bf(torch.randn(N, H, W, K, device=DEV)).permute(0, 3, 1, 2)
    w = _bf(torch.randn(O, K, device=DEV) / K ** 0.5)
    y0 = _bf(torch.randn(N, H, W, O, device=DEV)).permute(0, 3, 1, 2)
    y = y0.clone(memory_format=torch.preserve_format)
    out, _, _ = ops.gemm1x1(x, w, None, out=y, accumulate=True)
    assert out.data_ptr() == y.data_ptr()
    ref = y0.double() + torch.einsum("nkhw,ok->nohw", x.double(), w.double())
    assert l2err(y, ref) < 4e-3                    # one bf16 rounding of the sum
    assert l2err(y, ref) < l2err(y0, ref) / 50


def test_ocr_gather_bwd_acc_equals_separate_add():
    from snws import ops
    torch.manual_seed(6)
    B, H, W, K, C = 2, 24, 20, 10, ops.OCR_GATHER_C
    feats = _bf(torch.randn(B, H, W, C, device=DEV)).permute(0, 3, 1, 2)
    probs = _bf(torch.randn(B, H, W, 16, device=DEV) * 2).permute(0, 3, 1, 2)[:, :K]
    dctx = _bf(torch.randn(B, C, K, 1, device=DEV))
    other = _bf(torch.randn(B, H, W, C, device=DEV)).permute(0, 3, 1, 2)      # another consumer's share of d feats
    f1 = feats.detach().requires_grad_(True)
    ops.OcrGatherFn.apply(f1, probs, 1.0).backward(dctx)
    want = (f1.grad.float() + other.float()).to(torch.bfloat16)
    f2 = feats.detach().requires_grad_(True)
    shared = ops.share_grad(f2 * 1)                 # (a non-leaf producer, as in the model)
    slot = ops.grad_slot(shared)
    out = ops.OcrGatherFn.apply(shared, probs, 1.0)
    slot.buf = other.clone(memory_format=torch.preserve_format)   # what an earlier consumer would have parked
    parked = slot.buf
    # the gather is the only real consumer here: feed the parked tensor to the producer by hand
    (g,) = torch.autograd.grad(out, shared, dctx, allow_unused=True)
    assert g is None                                # delivered in place
    assert torch.equal(parked, want)


@pytest.mark.parametrize("Ca,Cb,O", [(512, 512, 512), (128, 256, 192)])
def test_conv1x1_pair_vs_float64_conv_of_concat(Ca, Cb, O):
    """conv1x1(cat([a, b]), w) as two products into one output, and its three gradients; bars: 1e-2 (bf16 operands, one
    extra bf16 rounding of the first partial product)."""
    from snws import ops
    torch.manual_seed(Ca + O)
    N, H, W = 2, 18, 22
    a = _bf(torch.randn(N, H, W, Ca, device=DEV)).permute(0, 3, 1, 2)
    b = _bf(torch.randn(N, H, W, Cb, device=DEV)).permute(0, 3, 1, 2)
    w = (torch.randn(O, Ca + Cb, 1, 1, device=DEV) / (Ca + Cb) ** 0.5)
    dy = _bf(torch.randn(N, H, W, O, device=DEV)).permute(0, 3, 1, 2)
    conv = torch.nn.Conv2d(Ca + Cb, O, 1, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w)
    assert ops.conv1x1_pair_ok(a, b, conv)
    a1, b1 = a.detach().requires_grad_(True), b.detach().requires_grad_(True)
    y = ops.Conv1x1PairFn.apply(a1, b1, conv.weight)
    y.backward(dy)
    a2, b2 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    w2 = conv.weight.detach().to(torch.bfloat16).double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(torch.cat([a2, b2], 1), w2)
    ref.backward(dy.double())
    assert l2err(y, ref) < 1e-2
    assert l2err(a1.grad, a2.grad) < 1e-2 and l2err(b1.grad, b2.grad) < 1e-2
    assert l2err(conv.weight.grad, w2.grad) < 1e-2


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2)])
def test_res_ocr_shared_gradients_match_autograd_accumulation(dtype, tol):
    """UNet3p_res_ocr_SNws with and without shared gradient buffers (decoder map -> auxiliary head + OCR branch; OCR
    features -> region gather + pixel branch + final 1x1 convolution): same loss, same gradients to bf16 rounding of the
    partial sums (median over parameters; the worst is bounded loosely - some gradients are tiny differences)."""
    from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws
    from snws import ops
    torch.manual_seed(3)
    model = UNet3p_res_ocr_SNws(14, 16, 10, 1, 1).to(DEV).train()
    model.compute_dtype = dtype
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    x = torch.randn(2, 14, 64, 64, device=DEV)
    lab = torch.randint(0, 10, (2, 64, 64), device=DEV)
    res = {}
    saved = ops.GRAD_SHARE
    try:
        for share in (True, False):
            ops.GRAD_SHARE = share
            model.zero_grad(set_to_none=True)
            hits0 = ops.GRAD_SHARE_HITS
            loss = torch.nn.functional.cross_entropy(model(x).float(), lab)
            loss.backward()
            # decoder map: 1 later contributor; OCR features: 2; encoder maps, pooled copies and tap maps: many more
            assert (ops.GRAD_SHARE_HITS - hits0 >= 10) == share, ops.GRAD_SHARE_HITS - hits0
            res[share] = (loss.item(), [p.grad.float().clone() for p in model.parameters() if p.grad is not None])
    finally:
        ops.GRAD_SHARE = saved
    assert abs(res[True][0] - res[False][0]) < 1e-3 * abs(res[False][0])
    errs = sorted(l2err(a, b) for a, b in zip(res[True][1], res[False][1]))
    assert len(errs) > 100
    assert errs[len(errs) // 2] < tol and errs[-1] < 0.35, (errs[len(errs) // 2], errs[-1])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_batchnorm_relu_dropout2d_in_one_pass(dtype, tol):
    """HipBatchNorm2d.with_dropout == nn.Dropout2d(nn.ReLU(nn.BatchNorm2d(x))) (modeling/UNet3p_res_ocr_SNws.py:118-124)
    for the mask a seeded F.dropout2d draws: output, input gradient, d gamma, d beta; whole (image, channel) maps are
    zero exactly where torch zeroes them."""
    from modeling.blocks import HipBatchNorm2d
    torch.manual_seed(11)
    N, C, H, W = 4, 64, 12, 10
    x = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    bn = HipBatchNorm2d(C, fuse_relu=True).to(DEV).train()
    ref_bn = torch.nn.BatchNorm2d(C).to(DEV).double().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5)
        bn.bias.copy_(torch.randn(C) * 0.3)
        ref_bn.weight.copy_(bn.weight.double())
        ref_bn.bias.copy_(bn.bias.double())
    drop = torch.nn.Dropout2d(0.3).train()
    x1 = x.detach().requires_grad_(True)
    torch.manual_seed(77)
    y = bn.with_dropout(x1, drop)
    y.backward(dy)
    # the reference composition; its mask comes from the same seeded draw in x's dtype
    torch.manual_seed(77)
    noise = x.new_empty((N, C, 1, 1)).bernoulli_(0.7).div_(0.7)
    torch.manual_seed(77)
    probe = torch.nn.functional.dropout2d(torch.ones_like(x), 0.3, True)
    assert torch.equal(probe, noise.expand_as(probe).to(probe.dtype))           # F.dropout2d draws exactly this mask
    x2 = x.detach().double().requires_grad_(True)
    ref = torch.relu(ref_bn(x2)) * noise.double()
    ref.backward(dy.double())
    dropped = (noise.reshape(N, C) == 0)
    assert dropped.any() and not dropped.all()
    assert (y.float().abs().amax(dim=(2, 3))[dropped] == 0).all()
    assert l2err(y, ref) < tol
    assert l2err(x1.grad, x2.grad) < (tol if dtype == torch.float32 else 2e-2)
    assert l2err(bn.weight.grad, ref_bn.weight.grad) < (tol if dtype == torch.float32 else 2e-2)
    assert l2err(bn.bias.grad, ref_bn.bias.grad) < (tol if dtype == torch.float32 else 2e-2)
    # p = 0 and evaluation mode: plain BatchNorm + ReLU
    bn.eval()
    assert torch.equal(bn.with_dropout(x, torch.nn.Dropout2d(0.0).train()), bn(x))
    assert torch.equal(bn.with_dropout(x, drop.eval()), bn(x))


@pytest.mark.parametrize("B,H,W,C,K", [(2, 20, 30, 512, 10), (1, 7, 9, 40, 3), (2, 33, 17, 100, 20)])
def test_ocr_float32_kernels_vs_float64(B, H, W, C, K):
    """The float32 tier's two attention steps (ops.OcrGather32Fn / ops.OcrAttend32Fn: any channel count, up to 64
    classes) against the reference's arithmetic (F.softmax + torch.matmul, modeling/UNet3p_res_ocr_SNws.py:21-28,
    87-100) in float64 on the same float32 inputs: forward and every gradient within 1e-5 (L2)."""
    from snws import ops
    torch.manual_seed(B * 1000 + C + K)
    feats = torch.randn(B, H, W, C, device=DEV).permute(0, 3, 1, 2)
    probs = (torch.randn(B, H, W, K, device=DEV) * 3).permute(0, 3, 1, 2)
    dctx = torch.randn(B, C, K, 1, device=DEV)
    f1, p1 = feats.detach().requires_grad_(True), probs.detach().requires_grad_(True)
    assert ops.ocr32_ok(f1, K) and not ops.ocr_gather_ok(f1, p1)
    out = ops.OcrGather32Fn.apply(f1, p1, 1.0)
    assert out.shape == (B, C, K, 1) and out.dtype == torch.float32
    out.backward(dctx)
    f2, p2 = feats.detach().double().requires_grad_(True), probs.detach().double().requires_grad_(True)
    pr = torch.softmax(p2.reshape(B, K, -1), dim=2)
    ref = torch.matmul(pr, f2.permute(0, 2, 3, 1).reshape(B, H * W, C)).permute(0, 2, 1).unsqueeze(3)
    ref.backward(dctx.double())
    assert l2err(out, ref) < 1e-5
    assert l2err(f1.grad, f2.grad) < 1e-5
    assert l2err(p1.grad, p2.grad) < 1e-4          # differences of nearly equal sums: a few float32 ulps of the larger terms

    q = torch.randn(B, H, W, C, device=DEV).permute(0, 3, 1, 2)
    key = torch.randn(B, K, 1, C, device=DEV).permute(0, 3, 1, 2)
    val = torch.randn(B, K, 1, C, device=DEV).permute(0, 3, 1, 2)
    dout = torch.randn(B, H, W, C, device=DEV).permute(0, 3, 1, 2)
    scale = C ** -0.5
    a = [t.detach().requires_grad_(True) for t in (q, key, val)]
    y = ops.OcrAttend32Fn.apply(a[0], a[1], a[2], scale)
    y.backward(dout)
    r = [t.detach().double().requires_grad_(True) for t in (q, key, val)]
    query = r[0].permute(0, 2, 3, 1).reshape(B, H * W, C)
    sim = torch.softmax(torch.matmul(query, r[1].reshape(B, C, K)) * scale, dim=-1)
    ctx = torch.matmul(sim, r[2].reshape(B, C, K).permute(0, 2, 1))
    yr = ctx.reshape(B, H, W, C).permute(0, 3, 1, 2)
    yr.backward(dout.double())
    assert l2err(y, yr) < 1e-5
    for got, want in zip(a, r):
        assert l2err(got.grad, want.grad) < 1e-5
    # deterministic: the same launch twice gives the same bits
    a2 = [t.detach().requires_grad_(True) for t in (q, key, val)]
    ops.OcrAttend32Fn.apply(a2[0], a2[1], a2[2], scale).backward(dout)
    for u, v in zip(a, a2):
        assert torch.equal(u.grad, v.grad)


def test_ocr_modules_raise_on_unsupported_input():
    """No attention step falls back to stock torch ops: inputs neither kernel family takes raise."""
    from modeling.UNet3p_res_ocr_SNws import SpatialGather_Module
    feats = torch.randn(1, 8, 4, 4, device=DEV, dtype=torch.float64)
    probs = torch.randn(1, 3, 4, 4, device=DEV, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        SpatialGather_Module(3)(feats, probs)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,acc,bias", [(256, 512, False, True), (512, 256, False, False), (320, 320, False, True),
                                           (512, 512, True, False), (64, 128, False, True)])
def test_gemm1x1_statistics_epilogue(N, K, acc, bias):
    """snws_gemm1x1_stats: the product must equal snws_gemm1x1 / snws_gemm1x1_acc bit for bit, and the per-tile moments
    must be the sums of the STORED bf16 values over each 256-pixel tile (float64 reference from the stored tensor), for
    both channel-tile widths (256, 128 incl. a ragged last tile at 320), with a bias and in accumulate mode."""
    import ctypes
    from snws import _lib as L
    torch.manual_seed(N + K)
    imgs, HW = 3, 512
    M = imgs * HW
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=DEV) if bias else None
    base = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())      # noqa: E731
    want = base.clone()
    if acc:
        L.call("snws_gemm1x1_acc", P(x), K, P(w), P(want), N, M, N, K, st)
    else:
        L.call("snws_gemm1x1", P(x), K, P(w), P(b), P(want), N, M, N, K, st)
    T = L.lib().snws_gemm1x1_stat_tiles(HW)
    assert T == 2 and L.lib().snws_gemm1x1_stat_tiles(300) == -1
    got = base.clone()
    stat = torch.full((imgs, T, N, 2), float("nan"), device=DEV)
    L.call("snws_gemm1x1_stats", P(x), K, P(w), P(b), P(got), N, M, N, K, int(acc), HW, P(stat), st)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    y = got.double().view(imgs, T, 256, N)
    ref = torch.stack([y.sum(2), (y * y).sum(2)], -1)
    assert bool(torch.isfinite(stat).all())
    err = (stat.double() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-6, float(err)
    with pytest.raises(L.SnwsUnsupported):
        L.call("snws_gemm1x1_stats", P(x), K, P(w), P(b), P(got), N, M, N, K, int(acc), 300, P(stat), st)


@pytest.mark.gpu
def test_ocr_head_batchnorm_takes_moments_from_the_convolutions():
    """UNet3p_res_ocr_SNws in bf16: the BatchNorm layers behind the wide 1x1 convolutions, the 3x3 convolution and the
    two-product 1x1 convolution take their moments from the producing kernel's epilogue (ops.BN_STAT_HITS counts them);
    logits, running statistics and gradients agree with the run that re-reads every tensor for its statistics
    (SNWS_BN_STAT_FUSION=0 path) to bf16 rounding of the running buffers' inputs."""
    from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws
    from snws import ops

    def run(fuse):
        torch.manual_seed(8)
        model = UNet3p_res_ocr_SNws(14, 16, 10, 1, 1).to(DEV).train()
        model.compute_dtype = torch.bfloat16
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        g = torch.Generator().manual_seed(4)
        img = (torch.rand(2, 14, 64, 64, generator=g) * 255.0).to(DEV)
        lab = torch.randint(0, 10, (2, 64, 64), generator=g).to(DEV)
        old, ops.BN_STAT_FUSION = ops.BN_STAT_FUSION, fuse
        hits0 = ops.BN_STAT_HITS
        try:
            out = model(img)
            logits = out[1] if isinstance(out, (tuple, list)) else out
            aux = out[0] if isinstance(out, (tuple, list)) else None
            loss = torch.nn.functional.cross_entropy(logits.float(), lab)
            if aux is not None:
                loss = loss + 0.4 * torch.nn.functional.cross_entropy(aux.float(), lab)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            ops.BN_STAT_FUSION = old
        bufs = {k: v.detach().float().cpu().clone() for k, v in model.named_buffers() if "running" in k}
        grads = torch.cat([p.grad.detach().float().flatten().cpu() for p in model.parameters() if p.grad is not None])
        return logits.detach().float().cpu(), bufs, grads, ops.BN_STAT_HITS - hits0

    l1, b1, g1, hits = run(True)
    l0, b0, g0, hits0 = run(False)
    assert hits0 == 0 and hits >= 6, (hits, hits0)
    assert l2err(l1, l0) < 2e-2
    for k in b1:
        assert float((b1[k] - b0[k]).abs().max()) <= 2e-3 * float(b0[k].abs().max()) + 1e-5, k
    assert torch.isfinite(g1).all()
