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
