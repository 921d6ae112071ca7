"""GPU parity tests: every HIP kernel, called through the C ABI (via the host modules), against the CPU oracle and
the reference-generated golden fixtures.  Run on the MI355X box with `-m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import snws_oracle as O
import patterns

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def relerr(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


# ================================================================== Evaluator / histogram (bit-exact)
def _eval_check(ev, g, case):
    assert np.array_equal(ev.confusion_matrix, g[case + "_cm"])
    with np.errstate(all="ignore"):
        for fn, key in ((ev.Pixel_Accuracy, "_pa"), (ev.Pixel_Accuracy_Class, "_pac"),
                        (ev.Mean_Intersection_over_Union, "_miou"),
                        (ev.Frequency_Weighted_Intersection_over_Union, "_fwiou")):
            got = np.asarray(fn(), np.float64)
            assert np.array_equal(got.view(np.uint64), g[case + key].view(np.uint64)), (case, key)


def test_evaluator_golden_small():
    from utils.metrics import Evaluator
    g = load("g6_evaluator")
    ev = Evaluator(10)
    ev.add_batch(g["a_gt"], g["a_pred"])
    ev.add_batch(g["a_gt"][::-1].copy(), g["a_pred"])
    _eval_check(ev, g, "a")
    ev = Evaluator(10)
    ev.add_batch(g["b_gt"], g["b_pred"])        # 255 / -1 / fractional / NaN / 10.0 labels
    _eval_check(ev, g, "b")
    ev = Evaluator(3)
    ev.add_batch(g["e_gt"], g["e_pred"])
    _eval_check(ev, g, "e")
    ev = Evaluator(10)
    ev.add_batch(np.full((1, 8, 8), 255.0, np.float32), np.zeros((1, 8, 8), np.int64))
    assert np.array_equal(ev.confusion_matrix, g["c_cm"])
    ev.reset()
    assert ev.confusion_matrix.sum() == 0


def test_evaluator_full_disk_bit_exact():
    from utils.metrics import Evaluator
    g = load("g6_evaluator")
    rng = np.random.default_rng(int(g["d_seed"]))
    gt = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.float32)
    pr = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.int64)
    ev = Evaluator(10)
    ev.add_batch(gt, pr)
    _eval_check(ev, g, "d")
    # device tensors, several dtypes, accumulated twice == 2x
    ev2 = Evaluator(10)
    ev2.add_batch(torch.from_numpy(gt).to(DEV), torch.from_numpy(pr).to(DEV))
    ev2.add_batch(torch.from_numpy(gt).to(DEV).long(), torch.from_numpy(pr).to(DEV).int())
    assert np.array_equal(ev2._matrix(), 2 * g["d_cm"])


def test_evaluator_many_classes_and_unaligned():
    """64 classes (LDS histogram copies are limited) and inputs whose base address is not 16-byte aligned."""
    from utils.metrics import Evaluator
    rng = np.random.default_rng(5)
    gt = rng.integers(0, 70, size=(3, 33, 47)).astype(np.float32)
    pr = rng.integers(0, 64, size=(3, 33, 47)).astype(np.int64)
    ev = Evaluator(64)
    ev.add_batch(gt, pr)
    assert np.array_equal(ev.confusion_matrix, O.confusion_matrix(gt, pr, 64).astype(np.float64))
    g1 = torch.from_numpy(gt.reshape(-1)).to(DEV)[1:]
    p1 = torch.from_numpy(pr.reshape(-1)).to(DEV)[1:]
    ev = Evaluator(64)
    ev.add_batch(g1, p1)
    assert np.array_equal(ev._matrix(), O.confusion_matrix(gt.reshape(-1)[1:], pr.reshape(-1)[1:], 64).astype(np.float64))


def test_evaluator_edge_inputs():
    """Empty batch, all-ignored labels (255 / -1 / NaN), label and prediction dtypes other than f32 / i64, and the
    shape-mismatch assertion of utils/metrics.py:48 - each against the oracle's restatement of _generate_matrix."""
    from utils.metrics import Evaluator
    ev = Evaluator(10)
    ev.add_batch(np.zeros((0, 4, 4), np.float32), np.zeros((0, 4, 4), np.int64))          # nothing to count
    assert ev.confusion_matrix.sum() == 0
    ign = np.array([[255.0, -1.0, np.nan, 10.0, 300.0]], np.float32).reshape(1, 1, 5)
    ev.add_batch(ign, np.zeros((1, 1, 5), np.int64))
    assert ev._matrix().sum() == 0
    rng = np.random.default_rng(11)
    for gdt, pdt in ((np.uint8, np.int32), (np.int64, np.uint8), (np.float64, np.int64), (np.int32, np.int64)):
        gt = rng.integers(0, 12, size=(2, 19, 23)).astype(gdt)          # 10, 11 are outside the class range: ignored
        pr = rng.integers(0, 10, size=(2, 19, 23)).astype(pdt)
        e2 = Evaluator(10)
        e2.add_batch(gt, pr)
        assert np.array_equal(e2._matrix(), O.confusion_matrix(gt, pr, 10).astype(np.float64)), (gdt, pdt)
    with pytest.raises(AssertionError):
        ev.add_batch(np.zeros((1, 4, 4), np.float32), np.zeros((1, 4, 5), np.int64))
    # accumulation over calls and reset()
    e3 = Evaluator(3)
    a, b = np.array([[0.0, 1.0, 2.0, 2.0]], np.float32), np.array([[0, 1, 1, 2]], np.int64)
    e3.add_batch(a, b)
    e3.add_batch(a, b)
    assert np.array_equal(e3._matrix(), 2 * O.confusion_matrix(a, b, 3))
    e3.reset()
    assert e3._matrix().sum() == 0


def test_evaluator_out_of_range_prediction_raises():
    from utils.metrics import Evaluator
    ev = Evaluator(10)
    with pytest.raises(ValueError):
        ev.add_batch(np.array([[9.0]], np.float32), np.array([[10]], np.int64))
    ev = Evaluator(10)
    with pytest.raises(ValueError):
        ev.add_batch(np.array([[0.0]], np.float32), np.array([[-1]], np.int64))
    # pred >= C on a small label only lands in another cell, as np.bincount does
    ev = Evaluator(10)
    ev.add_batch(np.array([[1.0]], np.float32), np.array([[12]], np.int64))
    assert ev.confusion_matrix[2, 2] == 1
    with pytest.raises(AssertionError):
        ev.add_batch(np.zeros((2, 2), np.float32), np.zeros((2, 3), np.int64))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_evaluator_fused_argmax(dtype):
    from utils.metrics import Evaluator
    torch.manual_seed(3)
    logits = torch.randn(2, 10, 37, 53).to(dtype)
    logits[0, 3, 5, 5] = logits[0, 7, 5, 5] = 9.0       # tie -> first index
    logits[1, 4, 6, 6] = float("nan")                    # NaN counts as the maximum
    gt = torch.randint(0, 11, (2, 37, 53)).float()
    gt[0, 0, :7] = 255.0
    ev = Evaluator(10)
    ev.add_batch_logits(gt.to(DEV), logits.to(DEV))
    pred = np.argmax(logits.float().numpy(), axis=1)
    assert np.array_equal(ev._matrix(), O.confusion_matrix(gt.numpy(), pred, 10).astype(np.float64))
    evc = Evaluator(10)
    evc.add_batch_logits(gt.to(DEV), logits.to(DEV).contiguous(memory_format=torch.channels_last))
    assert np.array_equal(evc._matrix(), ev._matrix())


# ================================================================== weight standardisation
def test_ws_fwd_bwd_golden():
    from snws import ops
    g = load("g1_conv2dws")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        w = torch.from_numpy(g[pre + "w"]).to(DEV)
        cfg = ops.ConvCfg(w.shape, 1, 0, 1, 1)
        prep = ops.PreparedWeight(w, cfg, want_mfma=True)
        what = prep.ohwi.permute(0, 3, 1, 2)
        assert relerr(what, g[pre + "what"]) < 1e-5
        # closed-form backward against the oracle
        gr = torch.from_numpy(patterns.fill(tuple(w.shape), 50 + ci, 1.0)).float()
        want = O.ws_standardize_bwd(w.cpu(), gr)
        g_ohwi = gr.permute(0, 2, 3, 1).contiguous().to(DEV)
        dw = torch.empty_like(w)
        from snws import _lib as L
        import ctypes
        L.call("snws_ws_bwd", ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(g_ohwi.data_ptr()), w.shape[1],
               ctypes.c_void_p(dw.data_ptr()), w.shape[0], w.shape[1], w.shape[2], w.shape[3], 0, None)
        torch.cuda.synchronize()
        assert relerr(dw, want) < 1e-4
        # accumulate flag: a second call adds the same gradient
        L.call("snws_ws_bwd", ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(g_ohwi.data_ptr()), w.shape[1],
               ctypes.c_void_p(dw.data_ptr()), w.shape[0], w.shape[1], w.shape[2], w.shape[3], 1, None)
        torch.cuda.synchronize()
        assert relerr(dw, 2 * want) < 1e-4


@pytest.mark.parametrize("shape,ldg", [((64, 320, 3, 3), 320), ((32, 1024, 3, 3), 1024), ((8, 1820, 3, 3), 1824),
                                       ((16, 576, 1, 1), 576), ((5, 7, 3, 3), 16)])
def test_ws_bwd_wide_fan_in_vs_oracle(shape, ldg):
    """snws_ws_bwd on the fan-in classes of the benchmark model (16 and 64 elements per thread; 1820 input channels is
    the widest filter the kernel takes: its staged gradient row fills 64 KB of LDS) and with a padded channel stride of
    the incoming gradient, against the closed-form float64 backward of the oracle."""
    import ctypes
    from snws import _lib as L
    torch.manual_seed(4)
    w = torch.randn(shape) * 0.05
    gr = torch.randn(shape)
    want = O.ws_standardize_bwd(w.double(), gr.double())
    Oc, I, kh, kw = shape
    g_ohwi = torch.zeros(Oc, kh, kw, ldg)
    g_ohwi[..., :I] = gr.permute(0, 2, 3, 1)
    wd, gd = w.to(DEV), g_ohwi.to(DEV)
    dw = torch.empty_like(wd)
    L.call("snws_ws_bwd", ctypes.c_void_p(wd.data_ptr()), ctypes.c_void_p(gd.data_ptr()), ldg,
           ctypes.c_void_p(dw.data_ptr()), Oc, I, kh, kw, 0, None)
    torch.cuda.synchronize()
    assert relerr(dw, want) < 1e-5


# ================================================================== direct convolution (exact f32 tier)
@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("ci", range(4))
def test_conv2dws_direct_f32_golden(ci, direct):
    """Conv2dws in float32 against the reference's own outputs and gradients: through the default path (the f32
    matrix-instruction kernels for the stride-1 3x3 / 1x1 cases, the direct kernels for the strided / dilated ones) and
    pinned to the direct kernels."""
    import modeling.weightstd as ws
    g = load("g1_conv2dws")
    pre = "c%d_" % ci
    Oc, I, k, s, p, d, has_b = [int(v) for v in g[pre + "cfg"]]
    conv = ws.Conv2dws(I, Oc, k, s, p, d, 1, bool(has_b)).to(DEV)
    conv.force_direct = direct
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(g[pre + "w"]))
        if has_b:
            conv.bias.copy_(torch.from_numpy(g[pre + "b"]))
    x = torch.from_numpy(g[pre + "x"]).to(DEV).requires_grad_(True)
    y = conv(x)
    assert tuple(y.shape) == g[pre + "y"].shape
    assert relerr(y, g[pre + "y"]) < 1e-4
    y.backward(torch.from_numpy(g[pre + "dy"]).to(DEV))
    assert relerr(x.grad, g[pre + "dx"]) < 1e-4
    assert relerr(conv.weight.grad, g[pre + "dw"]) < 1e-3
    if has_b:
        assert relerr(conv.bias.grad, g[pre + "db"]) < 1e-4


def test_conv2dws_direct_groups_f32():
    import modeling.weightstd as ws
    torch.manual_seed(0)
    conv = ws.Conv2dws(8, 12, 3, 2, 1, 1, 4, True).to(DEV)
    x = torch.randn(2, 8, 9, 11, device=DEV, requires_grad=True)
    dy = torch.randn(2, 12, 5, 6)
    y = conv(x)
    y.backward(dy.to(DEV))
    xr = x.detach().cpu().requires_grad_(True)
    wr = conv.weight.detach().cpu().requires_grad_(True)
    br = conv.bias.detach().cpu().requires_grad_(True)
    yr = O.conv2dws_fwd(xr, wr, br, 2, 1, 1, 4)
    yr.backward(dy)
    assert relerr(y, yr) < 1e-4 and relerr(x.grad, xr.grad) < 1e-4
    assert relerr(conv.weight.grad, wr.grad) < 1e-3 and relerr(conv.bias.grad, br.grad) < 1e-4


# ================================================================== f32 matrix-instruction convolution (float32 tier)
F32_CASES = [
    # N, Cin, Cout, H, W, k
    (2, 64, 64, 32, 32, 3),
    (1, 14, 64, 20, 40, 3),     # first layer: 14 channels (not a multiple of 4: scalar loads), ragged tiles
    (2, 128, 64, 16, 32, 3),
    (1, 320, 320, 16, 32, 3),
    (2, 64, 128, 17, 33, 3),    # odd sizes: partial tiles in both directions
    (3, 24, 40, 9, 70, 3),      # channel tails in both tile directions, three pixel tiles across
    (1, 64, 64, 32, 64, 1),     # 1x1
    (2, 320, 144, 19, 37, 1),
    (1, 18, 8, 5, 5, 1),        # tiny
]


@pytest.mark.parametrize("case", F32_CASES)
def test_conv2dws_f32_mfma_vs_oracle(case):
    """The float32 tier on v_mfma_f32_32x32x2_f32 (forward with the SwitchNorm statistics epilogue, dgrad, wgrad through
    partial tiles + the fixed-order reduction, bias gradient) against float64 arithmetic on the CPU: 1e-5 relative
    (f32 accumulation over up to 2880 terms), far inside the tier's 1e-3 bar; the epilogue's per-tile moments against
    the output it just wrote; and the same results as the direct kernels."""
    import ctypes
    import modeling.weightstd as ws
    from snws import _lib as L
    from snws import ops
    N, Cin, Cout, H, W, k = case
    torch.manual_seed(3)
    conv = ws.Conv2dws(Cin, Cout, k, 1, (k - 1) // 2).to(DEV)
    x = torch.randn(N, Cin, H, W) * 2 + 0.5
    dy = torch.randn(N, Cout, H, W)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    prep = conv.prepared(False)
    y, stat, saved = ops.conv_forward(xg.detach(), prep, conv.bias, want_stats=True)
    assert saved[3] == "f32" and y.dtype == torch.float32
    dx, dw, db = ops.conv_backward(dy.to(DEV), prep, saved, conv.weight, True, True, True)
    w = conv.weight.detach().cpu().double()
    what = O.ws_standardize(w.float()).double()
    b = conv.bias.detach().cpu().double()
    xr = x.double().requires_grad_(True)
    whr = what.clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, whr, b, 1, (k - 1) // 2)
    yr.backward(dy.double())
    assert relerr(y, yr) < 1e-5, "fwd"
    assert relerr(dx, xr.grad) < 1e-5, "dgrad"
    assert relerr(dw, O.ws_standardize_bwd(w, whr.grad)) < 2e-5, "wgrad"
    assert relerr(db, dy.double().sum((0, 2, 3))) < 1e-5, "dbias"
    # statistics epilogue: per-tile sums of (y - bias) and its square, summed over tiles = the plane sums
    yb = (y.double().cpu() - b.view(1, -1, 1, 1))
    assert relerr(stat[..., 0].sum(1).cpu(), yb.sum((2, 3))) < 1e-4
    assert relerr(stat[..., 1].sum(1).cpu(), (yb * yb).sum((2, 3))) < 1e-5
    assert stat.shape[1] == L.lib().snws_conv_f32_tiles(H, W)
    # the direct kernels compute the same thing
    y2, _, saved2 = ops.conv_forward(xg.detach(), prep, conv.bias, force_direct=True)
    assert saved2[3] is False
    dx2, dw2, db2 = ops.conv_backward(dy.to(DEV), prep, saved2, conv.weight, True, True, True)
    assert relerr(y, y2) < 1e-5 and relerr(dx, dx2) < 1e-5 and relerr(dw, dw2) < 2e-5 and relerr(db, db2) < 1e-5
    # bit-reproducible
    dx3, dw3, db3 = ops.conv_backward(dy.to(DEV), prep, saved, conv.weight, True, True, True)
    assert torch.equal(dx, dx3) and torch.equal(dw, dw3) and torch.equal(db, db3)


# ================================================================== MFMA convolution (bf16 tier)
MFMA_CASES = [
    # N, Cin, Cout, H, W, k
    (2, 64, 64, 32, 32, 3),
    (1, 14, 64, 20, 40, 3),     # first-layer shape: 14 channels padded to 16, ragged tiles
    (2, 128, 64, 16, 32, 3),
    (1, 320, 320, 16, 32, 3),   # 5 output tiles, 10 chunks
    (2, 64, 128, 17, 33, 3),    # odd sizes: partial tiles in both directions
    (1, 64, 64, 32, 64, 1),     # 1x1
    (2, 320, 144, 19, 37, 1),   # 1x1, channel tails in both tile directions, ragged pixel tiles
    (1, 1024, 576, 8, 8, 1),    # 1x1, many tile pairs, one pixel tile
    (2, 128, 256, 9, 13, 1),    # 1x1 wide kernel, pixel count not a multiple of its 128-pixel tile
    (1, 112, 248, 24, 40, 1),   # 1x1 wide kernel, channel tails in both tile directions
    (1, 32, 24, 8, 8, 3),       # C_out not a multiple of 64 (direct wgrad), tiny map
]


@pytest.mark.parametrize("case", MFMA_CASES)
def test_conv2dws_mfma_bf16_vs_oracle(case):
    import modeling.weightstd as ws
    N, Cin, Cout, H, W, k = case
    torch.manual_seed(1)
    conv = ws.Conv2dws(Cin, Cout, k, 1, (k - 1) // 2).to(DEV)
    x = bf16_round(torch.randn(N, Cin, H, W) * 2 + 0.5)
    dy = bf16_round(torch.randn(N, Cout, H, W))
    xg = x.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv(xg)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (N, Cout, H, W)
    y.backward(dy.to(DEV).to(torch.bfloat16))
    # oracle on the same bf16-rounded inputs with the bf16-rounded standardized filter
    w = conv.weight.detach().cpu().double()
    what = bf16_round(O.ws_standardize(w.float())).double()
    b = conv.bias.detach().cpu().double()
    xr = x.double().requires_grad_(True)
    whr = what.clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, whr, b, 1, (k - 1) // 2)
    yr.backward(dy.double())
    assert relerr(y.float(), yr) < 1e-2, "fwd"
    assert relerr(xg.grad.float(), xr.grad) < 1e-2, "dgrad"
    dw_ref = O.ws_standardize_bwd(w, whr.grad)
    assert relerr(conv.weight.grad, dw_ref) < 1e-2, "wgrad"
    assert relerr(conv.bias.grad, dy.double().sum((0, 2, 3))) < 1e-2, "dbias"


def test_conv_mfma_exact_small_integers():
    """Integer-valued inputs, filters and output gradients: every product and every partial sum is an integer below
    2^24, exact in bf16 storage / f32 accumulation whatever the summation order, so the MFMA kernels (forward,
    dgrad, wgrad, bias gradient) must reproduce the direct kernels fed the SAME filter - and the exact integer
    convolution computed on the CPU in float64 - bit for bit (catches any fragment / tap / swizzle mix-up that random
    data would blur into a tolerance)."""
    from snws import ops
    torch.manual_seed(5)
    for (N, Cin, Cout, H, W, k) in ((1, 64, 64, 24, 40, 3), (2, 32, 128, 17, 23, 3), (1, 128, 64, 16, 32, 1)):
        w = torch.randint(-2, 3, (Cout, Cin, k, k)).float()
        b = torch.randint(-4, 5, (Cout,)).float()
        x = torch.randint(-3, 4, (N, Cin, H, W)).float()
        dy = torch.randint(-2, 3, (N, Cout, H, W)).float()
        xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, br, 1, (k - 1) // 2)
        yr.backward(dy.double())
        xg = x.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dyg = dy.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wg, bg = w.to(DEV), b.to(DEV)
        cfg = ops.ConvCfg(w.shape, 1, (k - 1) // 2, 1, 1)
        prep = ops.PreparedWeight(wg, cfg, True, standardize=False)
        res = {}
        for name, direct in (("mfma", False), ("direct", True)):
            y, _, saved = ops.conv_forward(xg, prep, bg, force_direct=direct)
            assert saved[3] == (not direct)          # the path that was asked for did run
            dx, dw, db = ops.conv_backward(dyg, prep, saved, wg, True, True, True)
            res[name] = (y, dx, dw, db)
        for a, c, what in zip(res["mfma"], res["direct"], ("fwd", "dgrad", "wgrad", "dbias")):
            assert torch.equal(a, c), (what, N, Cin, Cout, k)
        y, dx, dw, db = res["mfma"]
        assert torch.equal(y.float().cpu(), yr.detach().to(torch.bfloat16).float())
        assert torch.equal(dx.float().cpu(), xr.grad.to(torch.bfloat16).float())
        assert torch.equal(dw.cpu().double(), wr.grad) and torch.equal(db.cpu().double(), br.grad)


def test_conv2dws_mfma_matches_direct_kernels():
    """Conv2dws through the MFMA path against the same module pinned to the direct kernels: the direct path keeps
    the standardized filter in f32, the MFMA path rounds it to bf16, so they agree at bf16 resolution of the filter."""
    import modeling.weightstd as ws
    torch.manual_seed(5)
    N, Cin, Cout, H, W = 1, 64, 64, 24, 40
    conv = ws.Conv2dws(Cin, Cout, 3, 1, 1).to(DEV)
    x = torch.randint(-3, 4, (N, Cin, H, W)).float()
    xg = x.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y_fast = conv(xg).float()
        conv.force_direct = True
        conv._prep = None
        y_direct = conv(xg.float()).float()
    assert relerr(y_fast, y_direct) < 2e-2


@pytest.mark.parametrize("case", [
    (2, 64, 64, 40, 50, 3),      # 3x3, one channel-tile pair (wgrad9)
    (1, 320, 320, 24, 40, 3),    # 3x3, 25 pairs
    (2, 80, 144, 19, 33, 3),     # 3x3, channel tails in both tile directions
    (2, 16, 64, 24, 40, 3),      # C_in = 16: the narrow generic kernel (two pixel-row halves per output)
    (2, 16, 48, 20, 20, 1),      # 1x1 with C_in = 16
    (2, 64, 576, 16, 32, 1),     # 1x1, 64-wide kernel
    (2, 320, 16, 32, 32, 1),     # 1x1, the classification head's shape
    (2, 256, 128, 16, 16, 1),    # 1x1, 128-wide kernel
    # 1x1 with a multiple of 192 outputs and >= 128 K pixels: the 192 x 160 tile of the tap convolutions
    (1, 320, 576, 362, 363, 1),  # 6 pairs, a ragged last pixel tile
    (1, 1024, 576, 362, 363, 1), # 7 in-channel tiles, the last one 64 wide
    (1, 176, 192, 362, 363, 1),  # one out tile, an in-channel tail of 16
])
def test_wgrad_deterministic_flush_matches_atomic_flush_and_repeats(case):
    """snws_conv_mfma_wgrad_det (partial tiles + fixed-order reduction) against snws_conv_mfma_wgrad (float atomics)
    through the C ABI: same filter and bias gradient to f32 summation-order accuracy, the deterministic form bit-identical
    over repeated launches with the workspace and the output pre-filled with different garbage, and its bias gradient
    ADDED to what the buffer held."""
    import ctypes
    from snws import _lib as L
    from snws import ops
    N, Cin, Cout, H, W, k = case
    torch.manual_seed(11)
    x = torch.randn(N, H, W, Cin, device=DEV).to(torch.bfloat16)
    dy = torch.randn(N, H, W, Cout, device=DEV).to(torch.bfloat16)
    cfg = ops.ConvCfg((Cout, Cin, k, k), 1, (k - 1) // 2, 1, 1)
    d = cfg.desc(N, H, W, Cin, Cin, Cout)
    g_at = torch.zeros(Cout, k, k, Cin, device=DEV)
    b_at = torch.zeros(Cout, device=DEV)
    L.call("snws_conv_mfma_wgrad", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g_at.data_ptr(), b_at.data_ptr(), None)
    nbytes = L.lib().snws_conv_mfma_wgrad_workspace_bytes(ctypes.byref(d))
    assert nbytes > 0
    outs = []
    for rep, fill in enumerate((0.0, 123.0, float("nan"))):
        ws = torch.full((nbytes // 4,), fill, device=DEV)
        g = torch.full((Cout, k, k, Cin), fill, device=DEV)
        b = torch.full((Cout,), 0.5, device=DEV)
        L.call("snws_conv_mfma_wgrad_det", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(), b.data_ptr(),
               ws.data_ptr(), nbytes, None)
        outs.append((g, b))
    for g, b in outs[1:]:
        assert torch.equal(g, outs[0][0]) and torch.equal(b, outs[0][1])
    g, b = outs[0]
    assert relerr(g, g_at) < 1e-5 and relerr(b - 0.5, b_at) < 1e-5
    # and against the definition, in float64 (on the CPU; the large 1x1 cases as one float64 product on the GPU)
    if k == 1 and N * H * W > 100000:
        want = torch.einsum("po,pi->oi", dy.double().reshape(-1, Cout), x.double().reshape(-1, Cin))
        assert relerr(g.reshape(Cout, Cin), want) < 1e-5
        assert relerr(b - 0.5, dy.double().reshape(-1, Cout).sum(0)) < 1e-5
    else:
        xr = x.float().cpu().double().permute(0, 3, 1, 2)
        dyr = dy.float().cpu().double().permute(0, 3, 1, 2)
        wr = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(xr, wr, None, 1, (k - 1) // 2).backward(dyr)
        assert relerr(g.permute(0, 3, 1, 2), wr.grad) < 1e-5
    with pytest.raises(ValueError, match="workspace"):
        L.call("snws_conv_mfma_wgrad_det", ctypes.byref(d), x.data_ptr(), dy.data_ptr(), g.data_ptr(), b.data_ptr(),
               ws.data_ptr(), nbytes - 16, None)


def test_training_step_is_bit_reproducible():
    """Two runs of three Trainer steps (flat optimizer, fused loss, bf16 MFMA path) from the same state on the same
    batches: every loss, every parameter and every running buffer must be bit-identical - no float atomics are left
    on the training path (weight gradients go through partial tiles and a fixed-order reduction)."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import harness
    batches = [patterns.synth_batch_learnable(4, 14, 64, 200 + j) for j in range(3)]

    def run():
        model = UNet3p_SNws(14, 16, 10, 1, 1)
        patterns.det_init(model)
        model = model.to(DEV)
        model.compute_dtype = torch.bfloat16
        tr = harness.Trainer(model, total_iters=100)
        losses = [tr.step(i.to(DEV), l.to(DEV)).clone() for i, l in batches]
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        tr.close()
        return torch.stack(losses), state
    l0, s0 = run()
    l1, s1 = run()
    assert torch.equal(l0, l1), (l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


# ================================================================== SwitchNorm
def _sn_module(g, pre):
    import modeling.snorm as sn
    N, C, H, W, mov, ubn, lg = [int(v) for v in g[pre + "cfg"]]
    m = sn.SwitchNorm2d(C, using_movavg=bool(mov), using_bn=bool(ubn), last_gamma=bool(lg)).to(DEV)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(g[pre + "gamma"]))
        m.bias.copy_(torch.from_numpy(g[pre + "beta"]))
        m.mean_weight.copy_(torch.from_numpy(g[pre + "mw"]))
        m.var_weight.copy_(torch.from_numpy(g[pre + "vw"]))
    return m, ubn


def _sn_cases():
    g = load("g2_switchnorm")
    return sorted({k[:-4] for k in g.files if k.endswith("_cfg")})


@pytest.mark.parametrize("pre", _sn_cases())
def test_switchnorm_f32_golden(pre):
    g = load("g2_switchnorm")
    pre = pre + "_"
    m, ubn = _sn_module(g, pre)
    m.train()
    x = torch.from_numpy(g[pre + "x"]).to(DEV).requires_grad_(True)
    y = m(x)
    assert relerr(y, g[pre + "y"]) < 1e-4
    y.backward(torch.from_numpy(g[pre + "dy"]).to(DEV))
    assert relerr(x.grad, g[pre + "dx"]) < 1e-3
    assert relerr(m.weight.grad, g[pre + "dgamma"]) < 1e-3
    assert relerr(m.bias.grad, g[pre + "dbeta"]) < 1e-3
    # the two softmax-weight gradients are small differences of large sums: the float32 golden itself carries
    # ~1e-2 relative noise there, so they are held to 2e-2 against it and to 2e-3 against the float64 oracle
    assert relerr(m.mean_weight.grad, g[pre + "dmw"]) < 2e-2
    assert relerr(m.var_weight.grad, g[pre + "dvw"]) < 2e-2
    st = O.SNState(m.weight.numel(), using_movavg=m.using_movavg, using_bn=m.using_bn, dtype=torch.float64)
    st.gamma, st.beta = [torch.from_numpy(g[pre + k]).double().view(-1) for k in ("gamma", "beta")]
    st.mean_weight, st.var_weight = [torch.from_numpy(g[pre + k]).double() for k in ("mw", "vw")]
    _, c64 = O.switchnorm_fwd(torch.from_numpy(g[pre + "x"]).double(), st, True)
    o64 = O.switchnorm_bwd(torch.from_numpy(g[pre + "dy"]).double(), st, c64)
    assert relerr(m.mean_weight.grad, o64[3]) < 2e-3 and relerr(m.var_weight.grad, o64[4]) < 2e-3
    assert relerr(x.grad, o64[0]) < 1e-4
    if ubn:
        assert relerr(m.running_mean, g[pre + "rm1"]) < 1e-5
        assert relerr(m.running_var, g[pre + "rv1"]) < 1e-4
        with torch.no_grad():
            for j in range(3):
                m(x.detach() * (1.0 + 0.5 * (j + 1)) - 0.3 * j)
        assert relerr(m.running_mean, g[pre + "rm4"]) < 1e-5
        assert relerr(m.running_var, g[pre + "rv4"]) < 1e-4
    m.eval()
    with torch.no_grad():
        assert relerr(m(x.detach()), g[pre + "y_eval"]) < 1e-4


@pytest.mark.parametrize("shape,dtype,tol", [((2, 64, 32, 32), torch.bfloat16, 2e-2), ((3, 320, 16, 16), torch.bfloat16, 2e-2),
                                             ((2, 64, 32, 32), torch.float32, 1e-3), ((2, 1024, 8, 8), torch.bfloat16, 2e-2),
                                             ((2, 24, 9, 7), torch.bfloat16, 2e-2),
                                             ((1, 16, 12, 12), torch.bfloat16, 2e-2),      # a batch of one image
                                             ((2, 10, 9, 7), torch.bfloat16, 2e-2),        # channels not a multiple of 8: scalar lanes
                                             ((3, 5, 6, 11), torch.float32, 1e-3),
                                             ((1, 8, 300, 300), torch.float32, 1e-3),      # many tiles per plane
                                             ((16, 64, 128, 128), torch.bfloat16, 2e-2)])  # the benchmark's batch
def test_switchnorm_relu_vs_oracle(shape, dtype, tol):
    import modeling.snorm as sn
    torch.manual_seed(2)
    N, C, H, W = shape
    m = sn.SwitchNorm2d(C).to(DEV)
    m.fuse_relu = True
    st = O.SNState(C, dtype=torch.float64)
    with torch.no_grad():
        m.weight.copy_(torch.rand(1, C, 1, 1) + 0.5)
        m.bias.copy_(torch.randn(1, C, 1, 1) * 0.3)
        m.mean_weight.copy_(torch.randn(3))
        m.var_weight.copy_(torch.randn(3))
    st.gamma, st.beta = m.weight.detach().cpu().double().view(-1), m.bias.detach().cpu().double().view(-1)
    st.mean_weight, st.var_weight = m.mean_weight.detach().cpu().double(), m.var_weight.detach().cpu().double()
    x = (torch.randn(shape) * 1.7 + 0.4).to(dtype)
    dy = torch.randn(shape).to(dtype)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = m(xg)
    y.backward(dy.to(DEV))
    yr, ctx = O.switchnorm_fwd(x.double(), st, True, relu=True)
    dxr, dgr, dbr, dmwr, dvwr = O.switchnorm_bwd(dy.double(), st, ctx)
    assert y.dtype == dtype
    assert relerr(y.float(), yr) < tol
    # The ReLU mask is recomputed from the pre-activation: where that is zero to rounding (a handful of the elements of
    # a large tensor) the two implementations may legitimately disagree on the mask, and dx then differs by the whole
    # A * dy term there.  Every deviating element must be such a tie (output ~ 0 on both sides), and they must be rare.
    dxg = xg.grad.detach().double().cpu()
    bad = (dxg - dxr).abs() > tol * dxr.abs().max()
    ymax = float(yr.abs().max())
    assert int(bad.sum()) <= max(2, int(2e-5 * bad.numel())), int(bad.sum())
    assert bool((y.detach().double().cpu()[bad].abs() <= 2e-5 * ymax).all()) and bool((yr[bad].abs() <= 2e-5 * ymax).all())
    ptol = tol + (5e-3 if bool(bad.any()) else 0.0)      # a flipped tie moves the parameter sums by that element's dy
    assert relerr(m.weight.grad.view(-1), dgr) < ptol and relerr(m.bias.grad.view(-1), dbr) < ptol
    assert relerr(m.mean_weight.grad, dmwr) < 2 * ptol and relerr(m.var_weight.grad, dvwr) < 2 * ptol
    assert relerr(m.running_mean.view(-1), st.running_mean) < 1e-4
    assert relerr(m.running_var.view(-1), st.running_var) < 1e-3


@pytest.mark.parametrize("shape,dtype,tol", [((2, 8, 16, 16), torch.float32, 1e-3), ((2, 4, 64, 64), torch.float32, 1e-3),
                                             ((2, 64, 32, 32), torch.bfloat16, 2e-2)])
def test_resnet_basic_block_vs_oracle(shape, dtype, tol):
    """Resnet_BasicBlock (conv-SN-ReLU, conv-SN, += input, ReLU; UNet3p_res_ocr_SNws.py:134-161) as two fused ops
    against the float64 oracle: output, input gradient (conv path + skip path) and every parameter gradient."""
    from modeling.blocks import Resnet_BasicBlock
    N, C, H, W = shape
    torch.manual_seed(5)
    blk = Resnet_BasicBlock(C, C).to(DEV).train()
    with torch.no_grad():
        for n_ in (blk.bn1, blk.bn2):
            n_.weight.copy_(torch.rand(1, C, 1, 1) + 0.5)
            n_.bias.copy_(torch.randn(1, C, 1, 1) * 0.3)
            n_.mean_weight.copy_(torch.randn(3))
            n_.var_weight.copy_(torch.randn(3))
    x = (torch.randn(shape) * 1.3 + 0.2).to(dtype)
    dy = torch.randn(shape).to(dtype)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = blk(xg)
    y.backward(dy.to(DEV))
    rnd = torch.bfloat16 if dtype == torch.bfloat16 else None
    params = {"b." + k: v.detach().cpu().double().requires_grad_(not O.is_buffer(k)) for k, v in blk.state_dict().items()}
    for k in params:
        if O.is_buffer(k):
            params[k] = torch.zeros_like(params[k])
    net = O.OracleNet(params, act_round=rnd)
    xr = x.double().requires_grad_(True)
    yr = net.basic("b", xr)
    yr.backward(dy.double())
    assert relerr(y.float(), yr) < tol
    assert relerr(xg.grad.float(), xr.grad) < tol
    for k, v in blk.named_parameters():
        ref = params["b." + k].grad
        lim = 2 * tol if k.endswith(("mean_weight", "var_weight")) else tol
        assert relerr(v.grad.float().view(-1), ref.view(-1)) < lim, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_residual_gradient_from_the_apply_pass_equals_threshold_backward(dtype):
    """The gradient of the residual input of relu(SwitchNorm(conv(t)) + x) is [y > 0] * dy.  The backward apply pass
    of the block's second SwitchNorm stores it (snws_sn_bwd_apply_res); it must be taken, and must equal autograd's own
    threshold-backward bit for bit, as must the conv-path gradient next to it."""
    from modeling.blocks import Resnet_BasicBlock
    from snws import _lib as L
    torch.manual_seed(8)
    N, C, H, W = 2, 64, 24, 40
    blk = Resnet_BasicBlock(C, C).to(DEV).train()
    x = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    calls = []
    orig = L.call

    def counted(name, *a):
        calls.append(name)
        return orig(name, *a)
    L.call = counted
    try:
        xg = x.clone().requires_grad_(True)
        y = blk(xg)
        y.backward(dy)
    finally:
        L.call = orig
    assert "snws_sn_bwd_apply_res" in calls
    # the block computes relu(f(x) + x): d/dx = (conv path) + [y > 0] * dy; isolate the skip term with a detached twin
    xh = x.clone().requires_grad_(True)
    blk.zero_grad()
    y2 = blk(xh)        # second pass over the same modules: the running buffers move, outputs and gradients do not
    skip = torch.where(y2.detach() > 0, dy, torch.zeros((), dtype=dtype, device=DEV))
    from snws import ops
    real = ops.sn_backward

    def no_res(dy_, s, **kw):
        kw.pop("dres_dtype", None)
        return real(dy_, s, **kw)
    ops.sn_backward = no_res
    try:
        y2.backward(dy)
    finally:
        ops.sn_backward = real
    assert torch.equal(y2, y)
    assert torch.equal(xh.grad, xg.grad)
    assert float(skip.abs().sum()) > 0


def test_switchnorm_errors():
    import modeling.snorm as sn
    m = sn.SwitchNorm2d(4).to(DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 4, 8, device=DEV))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 4, 8, 8))


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (2, 14, 64, 24, 40), (1, 320, 64, 16, 32), (2, 128, 128, 9, 11)])
def test_fused_conv_sn_relu_block_bf16_vs_oracle(shape):
    """The fused ConvSnRelu op (conv epilogue moments -> SwitchNorm -> ReLU; backward through all three) against
    the oracle on bf16-rounded inputs with the bf16-rounded conv output, bar 2e-2."""
    from modeling.blocks import ConvSnRelu
    N, Cin, Cout, H, W = shape
    torch.manual_seed(11)
    blk = ConvSnRelu(Cin, Cout).to(DEV).train()
    with torch.no_grad():
        blk[1].weight.copy_(torch.rand(1, Cout, 1, 1) + 0.5)
        blk[1].bias.copy_(torch.randn(1, Cout, 1, 1) * 0.3)
        blk[1].mean_weight.copy_(torch.randn(3))
        blk[1].var_weight.copy_(torch.randn(3))
    x = bf16_round(torch.randn(N, Cin, H, W) * 1.5 + 0.3)
    dy = bf16_round(torch.randn(N, Cout, H, W))
    xg = x.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = blk(xg)
    y.backward(dy.to(DEV).to(torch.bfloat16))
    # oracle in float64 with the same two storage roundings (standardized filter, conv output)
    w = blk[0].weight.detach().cpu().double().requires_grad_(True)
    b = blk[0].bias.detach().cpu().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    rnd = lambda t: t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())  # noqa: E731
    conv = rnd(torch.nn.functional.conv2d(xr, rnd(O.ws_standardize(w)), b, 1, 1))
    st = O.SNState(Cout, dtype=torch.float64)
    st.gamma = blk[1].weight.detach().cpu().double().view(-1).requires_grad_(True)
    st.beta = blk[1].bias.detach().cpu().double().view(-1).requires_grad_(True)
    st.mean_weight = blk[1].mean_weight.detach().cpu().double().requires_grad_(True)
    st.var_weight = blk[1].var_weight.detach().cpu().double().requires_grad_(True)
    yr, _ = O.switchnorm_fwd(conv, st, True, relu=True)
    yr.backward(dy.double())
    assert relerr(y.float(), yr) < 2e-2
    assert relerr(xg.grad.float(), xr.grad) < 2e-2
    assert relerr(blk[0].weight.grad, w.grad) < 2e-2 and relerr(blk[0].bias.grad, b.grad) < 2e-2
    assert relerr(blk[1].weight.grad.view(-1), st.gamma.grad) < 2e-2
    assert relerr(blk[1].bias.grad.view(-1), st.beta.grad) < 2e-2
    assert relerr(blk[1].mean_weight.grad, st.mean_weight.grad) < 4e-2
    assert relerr(blk[1].var_weight.grad, st.var_weight.grad) < 4e-2
    assert relerr(blk[1].running_mean.view(-1), st.running_mean) < 1e-3
    assert relerr(blk[1].running_var.view(-1), st.running_var) < 2e-3


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, h, w, k (up-sampling factor), ksize, dtype, tol
    (2, 6, 4, 5, 7, 2, 3, torch.float32, 1e-4), (1, 5, 3, 3, 2, 8, 3, torch.float32, 1e-4),
    (2, 8, 8, 4, 4, 16, 3, torch.float32, 1e-4), (2, 16, 8, 6, 5, 2, 1, torch.float32, 1e-4),
    (2, 64, 64, 8, 8, 4, 3, torch.bfloat16, 2e-2), (1, 320, 64, 8, 16, 2, 3, torch.bfloat16, 2e-2),
    (1, 128, 64, 2, 2, 16, 3, torch.bfloat16, 2e-2), (2, 64, 32, 8, 8, 2, 1, torch.bfloat16, 2e-2),
    (2, 16, 16, 5, 3, 8, 3, torch.float32, 1e-4), (1, 64, 72, 3, 5, 8, 3, torch.bfloat16, 2e-2),
    (1, 32, 64, 3, 3, 16, 3, torch.bfloat16, 2e-2)])
def test_conv_of_upsampled_map_vs_oracle(case):
    """ConvSnRelu.on_upsampled(x, k) - the 1x1 'tap' conv at low resolution followed by the gather kernel - equals
    ConvSnRelu(nn.Upsample(k, bilinear)(x)) of the oracle, forward and every gradient."""
    from modeling.blocks import ConvSnRelu
    N, Cin, Cout, h, w, k, ks, dtype, tol = case
    torch.manual_seed(13)
    blk = ConvSnRelu(Cin, Cout, kernel_size=ks, padding=(ks - 1) // 2).to(DEV).train()
    with torch.no_grad():
        blk[1].weight.copy_(torch.rand(1, Cout, 1, 1) + 0.5)
        blk[1].bias.copy_(torch.randn(1, Cout, 1, 1) * 0.3)
        blk[1].mean_weight.copy_(torch.randn(3))
        blk[1].var_weight.copy_(torch.randn(3))
    x = (torch.randn(N, Cin, h, w) * 1.5 + 0.3).to(dtype)
    dy = torch.randn(N, Cout, h * k, w * k).to(dtype)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = blk.on_upsampled(xg, k)
    y.backward(dy.to(DEV))
    wr = blk[0].weight.detach().cpu().double().requires_grad_(True)
    br = blk[0].bias.detach().cpu().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    st = O.SNState(Cout, dtype=torch.float64)
    st.gamma = blk[1].weight.detach().cpu().double().view(-1).requires_grad_(True)
    st.beta = blk[1].bias.detach().cpu().double().view(-1).requires_grad_(True)
    st.mean_weight = blk[1].mean_weight.detach().cpu().double().requires_grad_(True)
    st.var_weight = blk[1].var_weight.detach().cpu().double().requires_grad_(True)
    if dtype == torch.float32:      # the reference's order of operations
        conv = O.conv2dws_fwd(O.upsample_bilinear(xr, k), wr, br, 1, (ks - 1) // 2)
    else:                           # same storage roundings as the GPU path (filter, Z, conv output) so ReLU masks agree
        rnd = lambda t: t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())  # noqa: E731
        conv = O.conv_of_upsampled_commuted(xr, rnd(O.ws_standardize(wr)), br, k, rnd)
    yr, _ = O.switchnorm_fwd(conv, st, True, relu=True)
    yr.backward(dy.double())
    assert tuple(y.shape) == tuple(yr.shape) and y.dtype == dtype
    assert relerr(y.float(), yr) < tol
    assert relerr(xg.grad.float(), xr.grad) < tol
    assert relerr(blk[0].weight.grad, wr.grad) < tol * 2 and relerr(blk[0].bias.grad, br.grad) < tol * 2
    assert relerr(blk[1].weight.grad.view(-1), st.gamma.grad) < tol
    assert relerr(blk[1].bias.grad.view(-1), st.beta.grad) < tol
    assert relerr(blk[1].mean_weight.grad, st.mean_weight.grad) < 4 * tol + 1e-3
    assert relerr(blk[1].var_weight.grad, st.var_weight.grad) < 4 * tol + 1e-3


# ================================================================== resampling
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode,k,shape", [("maxpool", 2, (2, 16, 8, 12)), ("maxpool", 4, (1, 8, 9, 13)), ("maxpool", 8, (1, 64, 16, 16)),
                                          ("bilinear", 2, (2, 16, 5, 6)), ("bilinear", 4, (1, 8, 4, 4)), ("bilinear", 16, (1, 64, 2, 3)),
                                          ("bilinear", 8, (1, 5, 3, 2))])
def test_resample_vs_oracle(mode, k, shape, dtype):
    from snws import ops
    torch.manual_seed(4)
    x = torch.randn(shape).to(dtype)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x.double().requires_grad_(True)
    if mode == "maxpool":
        y, yr = ops.max_pool_ceil(xg, k), O.maxpool_ceil(xr, k)
    else:
        y, yr = ops.upsample_bilinear(xg, k), O.upsample_bilinear(xr, k)
    dy = torch.randn(yr.shape).to(dtype)
    y.backward(dy.to(DEV))
    yr.backward(dy.double())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert relerr(y.float(), yr) < tol
    assert relerr(xg.grad.float(), xr.grad) < tol


# ================================================================== whole models
def _load_pattern_state(model):
    patterns.det_init(model)


@pytest.mark.parametrize("arch,tag", [("unet", "g4_unet"), ("unet3p", "g4_unet3p")])
def test_model_f32_golden(arch, tag):
    """Reference-generated logits / loss / every gradient / running buffers for the n_filters=4 models
    (float32 tier: direct kernels), tolerance 1e-3 relative as the float32 parity bar."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from modeling.UNet_SNws import UNet_SNws
    g = load(tag)
    model = (UNet3p_SNws if arch == "unet3p" else UNet_SNws)(14, 4, 10, 1, 1)
    assert list(model.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    assert [",".join(map(str, v.shape)) for v in model.state_dict().values()] == [str(s) for s in g["state_shapes"]]
    _load_pattern_state(model)
    model = model.to(DEV).train()
    img, lab = patterns.synth_batch(2, 14, 64, 101)
    logits = model(img.to(DEV))
    assert relerr(logits, g["logits"]) < 1e-3
    loss = torch.nn.functional.cross_entropy(logits, lab.to(DEV).long(), ignore_index=255) / logits.shape[0]
    assert abs(float(loss) - float(g["loss"])) < 1e-3 * float(g["loss"])
    loss.backward()
    for name, p in model.named_parameters():
        ref = g["grad:" + name]
        assert relerr(p.grad, ref) < 2e-3 or float(np.abs(ref).max()) < 1e-6, name
    for name, b in model.named_buffers():
        assert relerr(b, g["buf:" + name]) < 1e-3, name
    model.eval()
    with torch.no_grad():
        assert relerr(model(img.to(DEV)), g["logits_eval"]) < 1e-3


def test_res_ocr_model_f32_golden():
    """UNet3p_res_ocr_SNws (n_filters 4) in float32 against the reference run in float64.  With these pattern weights
    the model amplifies float32 rounding: the reference's OWN float32 run is 1.5e-3 of the logits and, in the median
    over parameters, 1.4e-2 of a gradient's magnitude away from exact (up to 0.3 for cancellation-dominated
    SwitchNorm mixing weights); the fixture records that per tensor as "gap:*".  Logits / loss are held to 1e-3
    relative + 3x the reference's float32 gap; gradients are compared per class of parameter through the median and maximum of their relative errors,
    since which tensors the rounding hits hardest differs between two float32 implementations.  The kernels themselves are
    held to 1e-3 in the per-op and per-block tests.  Dropout2d p = 0 as in the fixture."""
    from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws
    g = load("g4_unet3p_res_ocr")
    model = UNet3p_res_ocr_SNws(14, 4, 10, 1, 1)
    assert list(model.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    assert [",".join(map(str, v.shape)) for v in model.state_dict().values()] == [str(s) for s in g["state_shapes"]]
    assert [n for n, _ in model.named_parameters()] == [str(k) for k in g["param_names"]]
    patterns.det_init_ocr(model)
    model.ocr_distri_head.conv_bn_dropout[3].p = 0.0
    model = model.to(DEV).train()
    img, lab = patterns.synth_batch(2, 14, 64, 101)

    def check(got, key, what):
        ref = np.asarray(g[key], np.float64)
        err = float(np.abs(got.detach().double().cpu().numpy() - ref).max())
        tol = 1e-3 * float(np.abs(ref).max()) + 3.0 * float(g["gap:" + key]) + 1e-7
        assert err <= tol, (what, err, tol)
        return err / max(float(np.abs(ref).max()), 1e-30)

    logits = model(img.to(DEV))
    e = check(logits, "logits", "logits")
    loss = torch.nn.functional.cross_entropy(logits, lab.to(DEV).long(), ignore_index=255) / logits.shape[0]
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-3 * float(g["loss"]) + 3.0 * float(g["gap:loss"])
    loss.backward()
    # gradients, per class of parameter (conv/norm weight, bias, the two SwitchNorm mixing vectors): the median of
    # this implementation's relative errors must stay within 1e-3 + 3x the median of the reference's own float32
    # errors, the maximum within 3x the reference's worst tensor
    mine, theirs = {}, {}
    for name, p in model.named_parameters():
        ref = np.asarray(g["grad:" + name], np.float64)
        if p.grad is None:      # output_seg_map is registered but unused by forward
            assert float(np.abs(ref).max()) == 0.0, name
            continue
        err = float(np.abs(p.grad.detach().double().cpu().numpy() - ref).max())
        mag = float(np.abs(ref).max())
        if mag < 1e-9:          # exactly zero in exact arithmetic (a conv bias in front of BatchNorm)
            assert err <= 3.0 * float(g["gap:grad:" + name]) + 1e-6, name
            continue
        cls = name.rsplit(".", 1)[1]
        mine.setdefault(cls, []).append(err / mag)
        theirs.setdefault(cls, []).append(float(g["gap:grad:" + name]) / mag)
    worst_ref = max(max(v) for v in theirs.values())
    for cls in mine:
        print("res-OCR f32 grads .%s: median rel err %.3g (reference f32 %.3g), max %.3g (%.3g)"
              % (cls, np.median(mine[cls]), np.median(theirs[cls]), max(mine[cls]), max(theirs[cls])))
        assert np.median(mine[cls]) <= 1e-3 + 3.0 * np.median(theirs[cls]), cls
        assert max(mine[cls]) <= 1e-3 + 3.0 * worst_ref, cls        # heavy-tailed: bound by the reference's worst tensor
    print("res-OCR f32: logits rel err %.3g (reference f32 gap %.3g)"
          % (e, float(g["gap:logits"]) / float(np.abs(g["logits"]).max())))
    for name, b in model.named_buffers():
        assert relerr(b, g["buf:" + name]) < 1e-3, name
    model.eval()
    with torch.no_grad():
        check(model(img.to(DEV)), "logits_eval", "eval logits")


def test_res_ocr_bf16_vs_oracle():
    """The same model at MFMA-aligned widths (n_filters 16) in bf16 against the CPU oracle run with the SAME storage
    format (activations and filters rounded to bf16 where the GPU path stores them, arithmetic in f32).  With random
    filters, BatchNorm over two images and two soft-max attention steps on top of the 39 normalised blocks, bf16
    STORAGE alone moves this model's logits by ~0.2 relative L2 (oracle vs oracle, printed); the GPU path must stay
    well inside that: within 0.6x the storage noise of the same-storage oracle, and all gradients finite."""
    from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws
    torch.manual_seed(3)
    model = UNet3p_res_ocr_SNws(14, 16, 10, 1, 1)
    model.ocr_distri_head.conv_bn_dropout[3].p = 0.0
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img = torch.rand(2, 14, 64, 64) * 255.0
    lab = torch.randint(0, 10, (2, 64, 64)).float()
    logits_ref, _, _ = O.model_step("unet3p_res_ocr", {k: v.clone() for k, v in state.items()}, bf16_round(img), lab)
    logits_r16, _, _ = O.model_step("unet3p_res_ocr", {k: v.clone() for k, v in state.items()}, bf16_round(img), lab,
                                    act_round=torch.bfloat16)
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16
    logits = model(img.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, lab.to(DEV).long()) / 2
    loss.backward()
    assert torch.isfinite(logits).all()
    for name, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
    l2 = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())  # noqa: E731
    print("res-OCR bf16: GPU vs bf16-storage oracle %.4g; bf16-storage oracle vs f32 oracle %.4g; GPU vs f32 oracle %.4g"
          % (l2(logits.detach(), logits_r16), l2(logits_r16, logits_ref), l2(logits.detach(), logits_ref)))
    assert l2(logits.detach(), logits_r16) < 0.6 * l2(logits_r16, logits_ref)
    assert l2(logits.detach(), logits_ref) < 1.5 * l2(logits_r16, logits_ref)


def test_res_ocr_bf16_vs_reference_run_with_bf16_storage():
    """Whole-model bf16 bar for UNet3p_res_ocr_SNws against REFERENCE output: fixture g9b holds the reference model
    (14, 16, 10; Dropout2d p = 0) in a trained state, run with bf16 storage roundings applied from outside and float64
    arithmetic between them (the expected values), plus two yardsticks: how far that result moves after a relative
    1e-6 change of the parameters ("noise": logits 0.7 %, whole gradient vector 26 %) and how far the reference's own
    float32 arithmetic lands from it ("f32gap": 0.55 % / 16 %).  Gated for the MFMA path: logits and loss within the
    stated 2e-2 of the bf16-storage reference.  Whole-network gradients of this model are not reproducible under bf16
    storage by any two implementations (the reference's float32 and float64 runs differ by 16 %), so the gradient vector
    is held to 2e-2 + 1.5x the larger yardstick and the per-tensor median likewise; the backward kernels themselves are
    held to 2e-2 per op and per block elsewhere in this file."""
    import statistics
    from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws
    g = load("g9b_bf16_storage_res_ocr")
    state, img, lab, ref_grads = patterns.load_bf16_storage_fixture(g)
    model = UNet3p_res_ocr_SNws(14, int(g["n_filters"]), 10, 1, 1)
    model.ocr_distri_head.conv_bn_dropout[3].p = 0.0
    sd = model.state_dict()
    model.load_state_dict({k: state[k].reshape(sd[k].shape) for k in sd})
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16
    logits = model(img.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, lab.to(DEV).long(), ignore_index=255) / img.shape[0]
    loss.backward()

    def l2(a, b):
        a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
        return float((a - b).norm() / max(float(b.norm()), 1e-30))
    params = dict(model.named_parameters())
    names = [str(n) for n in g["param_names"] if params[str(n)].grad is not None]
    got = torch.cat([params[n].grad.flatten().double().cpu() for n in names])
    ref = torch.cat([ref_grads[n].flatten() for n in names])
    e_b, e_f, e_g = l2(logits, g["logits_bf16"]), l2(logits, g["logits_f32"]), l2(got, ref)
    yard_g = max(float(g["noise_grads_all"]), float(g["f32gap_grads_all"]))
    per = [l2(params[n].grad.reshape(-1), ref_grads[n].reshape(-1)) for n in names]
    per_yard = [max(float(g["noise:" + n]), float(g["f32gap:" + n])) for n in names]
    print("res-OCR bf16 vs reference with bf16 storage: logits %.4g (its rounding-flip noise %.4g, its own float32 gap %.4g; "
          "storage alone %.4g), vs float32 reference %.4g; loss %.6f vs %.6f; gradients (all, L2) %.4g (noise %.4g, "
          "float32 gap %.4g); per tensor: median %.4g (yardstick median %.4g)"
          % (e_b, float(g["noise_logits"]), float(g["f32gap_logits"]), float(g["storage_logits"]), e_f, float(loss.detach()),
             float(g["loss_bf16"]), e_g, float(g["noise_grads_all"]), float(g["f32gap_grads_all"]),
             statistics.median(per), statistics.median(per_yard)))
    assert e_b < 2e-2
    assert abs(float(loss) - float(g["loss_bf16"])) < 2e-2 * float(g["loss_bf16"])
    assert e_g < 2e-2 + 1.5 * yard_g
    assert statistics.median(per) < 2e-2 + 1.5 * statistics.median(per_yard)
    for n in names:
        assert torch.isfinite(params[n].grad).all(), n
    for k, b in model.named_buffers():
        if b.is_floating_point():
            ref_b = torch.from_numpy(g["buf1:" + k]).reshape(b.shape)
            assert float((b.detach().cpu() - ref_b).abs().max()) <= 2e-2 * float(ref_b.abs().max()) + 1e-4, k


def test_unet3p_bf16_vs_oracle():
    """Mid-size UNet3p_SNws (n_filters 16, 64x64, batch 2) in bf16 on the MFMA path.

    bf16 bar: 2e-2 relative (L2) on logits against the CPU oracle run with the SAME storage format (activations
    and the standardized filter rounded to bf16 where the GPU path stores them, all arithmetic in f32).  The
    distance of either of them to the pure-f32 oracle is a property of bf16 storage through 34 stacked
    normalised layers with random filters, not of the kernels; it is printed, and bounded loosely."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    torch.manual_seed(7)
    model = UNet3p_SNws(14, 16, 10, 1, 1)
    for p in model.parameters():            # non-trivial SwitchNorm parameters
        if p.dim() == 1 and p.numel() == 3:
            p.data.copy_(torch.randn(3) * 0.5)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img = torch.rand(2, 14, 64, 64) * 255.0
    lab = torch.randint(0, 10, (2, 64, 64)).float()
    lab[0, :3] = 255.0
    logits_ref, loss_ref, grads_ref = O.model_step("unet3p", state, bf16_round(img), lab)
    logits_r16, loss_r16, grads_r16 = O.model_step("unet3p", state, bf16_round(img), lab, act_round=torch.bfloat16)
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16
    logits = model(img.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, lab.to(DEV).long(), ignore_index=255) / 2
    loss.backward()
    def l2(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / max(float(b.norm()), 1e-12))
    print("logits: L2 rel vs f32 oracle %.4g (max-abs rel %.4g); vs bf16-storage oracle L2 %.4g (max-abs %.4g); "
          "bf16-storage oracle vs f32 oracle L2 %.4g" % (l2(logits, logits_ref), relerr(logits, logits_ref),
                                                        l2(logits, logits_r16), relerr(logits, logits_r16),
                                                        l2(logits_r16, logits_ref)))
    assert l2(logits, logits_r16) < 2e-2
    assert l2(logits, logits_ref) < 1.5 * l2(logits_r16, logits_ref) + 1e-2
    assert abs(float(loss.detach()) - loss_ref) < 2e-2 * abs(loss_ref)
    # Gradients of this random-label, random-filter step are small residuals of large cancelling sums: rounding
    # the FORWARD activations to bf16 alone moves them by ~40 % (median oracle-to-oracle distance, a CPU-only
    # quantity printed here), so whole-model bf16 gradients are reported, not gated; the backward arithmetic is
    # gated per op (conv dgrad/wgrad, SwitchNorm backward, the fused block below at 2e-2) and by the f32 golden
    # model test at 1e-3, and the end-to-end bf16 behaviour by the 200-step loss curve.
    import statistics
    e_all, d_all = [], []
    for name, p in model.named_parameters():
        e_all.append(l2(p.grad, grads_ref[name]))
        d_all.append(l2(grads_r16[name], grads_ref[name]))
        assert torch.isfinite(p.grad).all(), name
    print("gradient L2 rel error vs f32 oracle: median %.3g; bf16-storage oracle vs f32 oracle: median %.3g"
          % (statistics.median(e_all), statistics.median(d_all)))
    assert statistics.median(e_all) < 1.5 * statistics.median(d_all) + 5e-2


def test_unet3p_bf16_vs_reference_run_with_bf16_storage():
    """Whole-model bf16 bar against REFERENCE output: fixture g9 holds the reference UNet3p_SNws(14, 16, 10) in a
    trained state (20 AdamW steps), run (a) in plain float32 and (b) with bf16 storage roundings applied to the
    reference from outside, plus how far (b) moves when the parameters move by a relative 1e-6 (the reference's own
    bf16 self-noise: logits 0.65 %, all gradients together 1.6 %, L2).
    Gated here for the MFMA path: logits and loss within 2e-2 of BOTH reference runs; the whole gradient vector
    within 2e-2 + the recorded self-noise of (b) (two exact evaluations of the same bf16-storage network differ by that
    much), and within 1.5x the storage-vs-float32 distance of the float32 gradient; per-tensor medians likewise."""
    import statistics
    from modeling.UNet3p_SNws import UNet3p_SNws
    g = load("g9_bf16_storage")
    state, img, lab, ref_grads = patterns.load_bf16_storage_fixture(g)
    model = UNet3p_SNws(14, int(g["n_filters"]), 10, 1, 1)
    sd = model.state_dict()
    model.load_state_dict({k: state[k].reshape(sd[k].shape) for k in sd})
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16
    logits = model(img.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, lab.to(DEV).long(), ignore_index=255) / img.shape[0]
    loss.backward()

    def l2(a, b):
        a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
        return float((a - b).norm() / max(float(b.norm()), 1e-30))
    noise_l, noise_g = float(g["noise_logits"]), float(g["noise_grads_all"])
    e_b, e_f = l2(logits, g["logits_bf16"]), l2(logits, g["logits_f32"])
    names = [str(n) for n in g["param_names"]]
    params = dict(model.named_parameters())
    got = torch.cat([params[n].grad.flatten().double().cpu() for n in names])
    ref = torch.cat([ref_grads[n].flatten() for n in names])
    e_g = l2(got, ref)
    per = [l2(params[n].grad, ref_grads[n].reshape(params[n].shape)) for n in names if n.endswith(".0.weight")]
    per_noise = [float(g["noise:" + n]) for n in names if n.endswith(".0.weight")]
    print("logits: vs reference-with-bf16-storage %.4g (its self-noise %.4g), vs float32 reference %.4g (storage alone "
          "%.4g); loss %.6f vs %.6f / %.6f; gradients (all, L2): %.4g (self-noise %.4g, storage alone %.4g); conv "
          "filters per tensor: median %.4g, worst %.4g (self-noise median %.4g, worst %.4g)"
          % (e_b, noise_l, e_f, float(g["storage_logits"]), float(loss), float(g["loss_bf16"]), float(g["loss_f32"]),
             e_g, noise_g, float(g["storage_grads_all"]), statistics.median(per), max(per),
             statistics.median(per_noise), max(per_noise)))
    assert e_b < 2e-2 and e_f < 2e-2
    assert abs(float(loss) - float(g["loss_bf16"])) < 2e-2 * float(g["loss_bf16"])
    assert abs(float(loss) - float(g["loss_f32"])) < 2e-2 * float(g["loss_f32"])
    assert e_g < 2e-2 + noise_g
    assert statistics.median(per) < 2e-2 + statistics.median(per_noise)
    for n in names:
        assert torch.isfinite(params[n].grad).all(), n
    for k, b in model.named_buffers():
        ref_b = torch.from_numpy(g["buf1:" + k])
        assert float((b.detach().cpu() - ref_b).abs().max()) <= 2e-2 * float(ref_b.abs().max()) + 1e-4, k


# ================================================================== 200-step loss curve
def _train_curve(model, batches, steps, dtype):
    from snws import harness
    model = model.to(DEV)
    model.compute_dtype = dtype
    tr = harness.Trainer(model, lr=2e-3, weight_decay=5e-4, total_iters=200)
    dev_batches = [(i.to(DEV), l.to(DEV)) for i, l in batches]
    losses = []
    for it in range(steps):
        img, lab = dev_batches[it % len(dev_batches)]
        losses.append(tr.step(img, lab))
    return torch.stack(losses).cpu().numpy()


def test_loss_curve_f32_tracks_reference():
    """200 AdamW/poly-LR steps of UNet3p_SNws(14,4,10) on the fixed synthetic batches: the float32 HIP path follows
    the reference's recorded loss curve.  Training amplifies summation-order differences (1e-7 at step 1, 1e-2..1e-1 by
    step 200), so the same 200 steps are run on both float32 convolution paths - the f32 matrix-instruction kernels and
    the direct kernels, two exact-f32 evaluation orders of the same arithmetic: each must stay within 1e-2 of the
    reference over the first 20 steps, within 10 % everywhere, and end within 10 % of the reference's final level; how
    far the two HIP runs end up from each other is printed as the yardstick for that last bound."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import ops
    g = load("g5_loss_curve")
    ref = g["losses"]
    batches = [patterns.synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
    curves = {}
    for name, flag in (("f32 mfma", True), ("direct", False)):
        model = UNet3p_SNws(14, 4, 10, 1, 1)
        patterns.det_init(model)
        old = ops.F32_MFMA
        ops.F32_MFMA = flag
        try:
            got = curves[name] = _train_curve(model, batches, 200, torch.float32)
        finally:
            ops.F32_MFMA = old
        rel = np.abs(got - ref) / ref
        print("%s curve: max rel dev %.4g (first 20 steps %.4g), final %.5f vs %.5f"
              % (name, rel.max(), rel[:20].max(), got[-1], ref[-1]))
        assert rel[:20].max() < 1e-2 and rel.max() < 0.1
        assert abs(got[-10:].mean() - ref[-10:].mean()) < 0.1 * ref[-10:].mean()
    a, b = curves["f32 mfma"], curves["direct"]
    print("the two HIP float32 paths end %.4g apart (relative, mean of the last 10 steps); first 20 steps %.4g"
          % (abs(a[-10:].mean() - b[-10:].mean()) / b[-10:].mean(), (np.abs(a - b) / b)[:20].max()))
    assert (np.abs(a - b) / b)[:20].max() < 1e-2


def test_loss_curve_bf16_tracks_reference():
    """The same 200 steps at n_filters=16 (MFMA-aligned widths) in bf16 track (a) the reference's 200-step curve run
    WITH bf16 storage roundings (fixture g5c: the reference model under the g9 hooks, float32 master weights and
    optimizer) and (b) its float32 curve (g5b).  Training amplifies rounding differences: g5c holds the reference's own
    bf16-storage curve three times, from initial parameters a relative 1e-6 apart, and those runs end up to 19 % apart
    at single steps and 17 % in the 10-step moving average - two correct implementations cannot agree better.  Gates
    against (a): the first 5 steps, before the amplification, within 2e-2; every step within 0.10 + the reference's own
    spread; the moving average within 0.05 + its own spread; the final level (last 10 steps) within 0.05 + its own spread (11 %).  Against (b), the float32
    curve, the looser bounds a storage format justifies (5 % / 35 % / 30 % / 15 %)."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    path = os.path.join(G, "g5b_loss_curve_nf16.npz")
    if not os.path.exists(path):
        pytest.skip("g5b fixture not generated")
    g = np.load(path)
    model = UNet3p_SNws(14, 16, 10, 1, 1)
    patterns.det_init(model)
    batches = [patterns.synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
    got = _train_curve(model, batches, 200, torch.bfloat16)
    ref = g["losses"]
    rel = np.abs(got - ref) / ref
    ker = np.ones(10) / 10.0
    sm = np.abs(np.convolve(got, ker, "valid") - np.convolve(ref, ker, "valid")) / np.convolve(ref, ker, "valid")
    print("bf16 curve: max rel dev %.4g (first 20 steps %.4g), moving-average dev %.4g, final %.5f vs %.5f"
          % (rel.max(), rel[:20].max(), sm.max(), got[-1], ref[-1]))
    assert rel[:5].max() < 0.05 and rel.max() < 0.35
    assert sm.max() < 0.30
    assert abs(got[-10:].mean() - ref[-10:].mean()) < 0.15 * ref[-10:].mean()
    assert got[-1] < 0.5 * got[0]
    gc = load("g5c_loss_curve_nf16_bf16_storage")
    refc, others = gc["losses"], gc["losses_perturbed"]
    mov = lambda v: np.convolve(v, ker, "valid")        # noqa: E731
    spread = float(max((np.abs(o - refc) / refc).max() for o in others))
    spread_sm = float(max((np.abs(mov(o) - mov(refc)) / mov(refc)).max() for o in others))
    relc = np.abs(got - refc) / refc
    smc = np.abs(mov(got) - mov(refc)) / mov(refc)
    print("bf16 curve vs the reference's bf16-storage curve: first 5 steps %.4g, max rel dev %.4g (reference's own spread "
          "%.4g), moving-average dev %.4g (own spread %.4g), final %.5f vs %.5f"
          % (relc[:5].max(), relc.max(), spread, smc.max(), spread_sm, got[-10:].mean(), refc[-10:].mean()))
    assert relc[:5].max() < 2e-2
    assert relc.max() < 0.10 + spread
    assert smc.max() < 0.05 + spread_sm
    spread_end = float(max(abs(o[-10:].mean() - refc[-10:].mean()) for o in others)) / refc[-10:].mean()
    assert abs(got[-10:].mean() - refc[-10:].mean()) < (0.05 + spread_end) * refc[-10:].mean()


# ---------------------------------------------------------------- fused loss, tiled inference
@pytest.mark.parametrize("dtype,cp", [(torch.float32, 10), (torch.bfloat16, 16), (torch.bfloat16, 10), (torch.float32, 3)])
def test_fused_cross_entropy_vs_torch(dtype, cp):
    """FusedCEFn == CrossEntropyLoss(ignore_index=255, reduction='sum') * scale and its gradient (utils/loss.py:
    146-159), with float labels, ignored pixels and the padded head (channels n_class..cp-1 get zero gradient)."""
    from snws import ops
    torch.manual_seed(4)
    n_class = min(cp, 10)
    N, H, W = 3, 37, 29
    logits = (torch.randn(N, cp, H, W) * 3).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    target = torch.randint(0, n_class, (N, H, W)).float()
    target[0, :5] = 255.0
    target[2, 7, 3] = 255.0
    target = target.to(DEV)
    scale = torch.tensor(1.0 / (float((target != 255).sum()) * N), device=DEV)
    loss = ops.FusedCEFn.apply(logits, target, n_class, scale, 255)
    (loss * 1.5).backward()
    ref_in = logits.detach()[:, :n_class].float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, target.long(), ignore_index=255, reduction="sum") * scale
    (ref * 1.5).backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert relerr(logits.grad[:, :n_class].float(), ref_in.grad) < tol
    if cp > n_class:
        assert float(logits.grad[:, n_class:].abs().max()) == 0.0
    # the reference's own composition: mean over valid pixels, then / batch
    ref2 = torch.nn.functional.cross_entropy(ref_in.detach(), target.long(), ignore_index=255) / N
    assert abs(float(loss) - float(ref2)) <= 1e-5 * abs(float(ref2))


@pytest.mark.parametrize("dtype,cp", [(torch.float32, 10), (torch.bfloat16, 16)])
def test_fused_cross_entropy_class_weights_golden(dtype, cp):
    """Class-weighted criterion (nn.CrossEntropyLoss(weight=w) / N, utils/loss.py:146-159 with train.py:55-67's
    weights) against the REFERENCE's loss and gradient (fixture g8w), through both loss routes of the harness:
    the fused kernel on (padded) raw logits and the torch route; plus the kernel's forward-only form."""
    from snws import harness, ops
    g = load("g8w_weighted_ce")
    w = torch.from_numpy(g["weight"]).to(DEV)
    target = torch.from_numpy(g["target"]).to(DEV)
    base = torch.from_numpy(g["logit"]).to(dtype)
    N, C, H, W = base.shape
    raw = torch.zeros(N, cp, H, W, dtype=dtype)
    raw[:, :C] = base
    raw = raw.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    loss = harness.global_ce_loss_fused(raw, C, target, 255, None, w)
    loss.backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert abs(float(loss) - float(g["loss"])) <= tol * abs(float(g["loss"]))
    assert relerr(raw.grad[:, :C].float(), g["dlogit"]) < tol
    if cp > C:
        assert float(raw.grad[:, C:].abs().max()) == 0.0
    plain = base.to(DEV).float().requires_grad_(True)
    loss2 = harness.global_ce_loss(plain, target, 255, None, w)
    loss2.backward()
    assert abs(float(loss2) - float(g["loss"])) <= tol * abs(float(g["loss"]))
    assert relerr(plain.grad, g["dlogit"]) < tol
    part, _, dl = ops.ce_partials(raw.detach(), target, C, 255, w)
    assert dl is None
    valid = target.long() != 255
    norm = float((w[target.long().clamp(0, C - 1)] * valid).sum())
    assert abs(float(part.sum()) / norm / N - float(g["loss"])) <= tol * abs(float(g["loss"]))


def test_trainer_validation_loss_and_class_weights():
    """Trainer.validate leaves the criterion summed over the validation batches (train.py:291-292) in val_loss; with
    class weights the training step's loss and gradients equal the torch route's."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import harness
    from utils.metrics import Evaluator
    w = torch.linspace(0.5, 2.0, 10)
    batches = [patterns.synth_batch(2, 14, 64, 101 + j) for j in range(3)]
    batches = [(i.to(DEV), l.to(DEV)) for i, l in batches]
    batches[1][1][0, :7] = 255.0
    grads, losses = [], []
    for fused in (False, True):
        model = UNet3p_SNws(14, 4, 10, 1, 1)
        patterns.det_init(model)
        model = model.to(DEV)
        tr = harness.Trainer(model, fused_loss=fused, class_weight=w)
        tr.opt.step = lambda *a, **k: None
        losses.append(float(tr.step(*batches[1])))
        grads.append({n: p.grad.clone() for n, p in model.named_parameters()})
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[0])
    for n in grads[0]:      # two float32 routes for d(loss)/d(logits): summation-order noise through 34 blocks
        assert relerr(grads[1][n], grads[0][n]) < 5e-4 or float(grads[0][n].abs().max()) < 1e-7, n
    for cw in (None, w):
        tr.class_weight = cw
        ev = Evaluator(10)
        tr.validate(batches, ev)
        model.eval()
        want = 0.0
        with torch.no_grad():
            for img, lab in batches:
                want += float(torch.nn.functional.cross_entropy(
                    model(img).float(), lab.long(), weight=None if cw is None else cw.to(DEV), ignore_index=255)) / img.shape[0]
        assert abs(tr.val_loss - want) <= 1e-5 * abs(want)
        assert abs(tr.val_loss_mean - want / 3) <= 1e-5 * abs(want)


def test_trainer_fused_loss_matches_unfused():
    """One Trainer.step with the fused loss head gives the same loss and the same parameter gradients as the
    float32-logits + torch cross-entropy path."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import harness
    img, lab = patterns.synth_batch(2, 14, 64, 101)
    grads, losses = [], []
    for fused in (False, True):
        model = UNet3p_SNws(14, 4, 10, 1, 1)
        patterns.det_init(model)
        model = model.to(DEV)
        tr = harness.Trainer(model, fused_loss=fused)
        tr.opt.step = lambda *a, **k: None        # keep the gradients of this one step
        losses.append(float(tr.step(img.to(DEV), lab.to(DEV))))
        grads.append({n: p.grad.clone() for n, p in model.named_parameters()})
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[0])
    for n in grads[0]:
        assert relerr(grads[1][n], grads[0][n]) < 1e-4 or float(grads[0][n].abs().max()) < 1e-7, n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_argmax_stitch_vs_torch(dtype):
    from snws import ops
    torch.manual_seed(6)
    H, W, b = 70, 90, 32
    pred = torch.full((H, W), -1, dtype=torch.int64, device=DEV)
    ref = pred.clone()
    for (y, x) in [(0, 0), (20, 50), (38, 58), (60, 80)]:          # overlapping, and one tile hanging over the border
        lg = torch.randn(1, 16, b, b).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
        lg[0, 3, 5, 7] = float("nan")
        lg[0, :, 9, 9] = 1.0                                         # tie: first maximum
        ops.argmax_stitch(lg, 10, pred, y, x)
        am = torch.argmax(lg[0, :10].float(), dim=0)
        hh, ww = min(b, H - y), min(b, W - x)
        ref[y:y + hh, x:x + ww] = am[:hh, :ww]
    assert torch.equal(pred, ref)
    lg = torch.zeros(1, 11, 4, 4, device=DEV).contiguous(memory_format=torch.channels_last)
    lg[0, 10, 0, 0] = 1.0
    lg[0, 4, 1, 1] = 1.0
    out = torch.zeros((4, 4), dtype=torch.int64, device=DEV)
    ops.argmax_stitch(lg, 11, out, 0, 0, remap_background=True)
    assert int(out[0, 0]) == 255 and int(out[1, 1]) == 4 and int(out[2, 2]) == 255     # classes 10 and 0 -> 255


def test_predict_disk_matches_tile_by_tile():
    """predict_disk (batched tiles, on-device arg-max + stitch) == the reference's loop written with plain torch:
    tile offsets from the oracle's gen_tiles_offs, one eval forward per tile, argmax, later tile overwrites."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import tiles
    model = UNet3p_SNws(14, 4, 10, 1, 1)
    patterns.det_init(model)
    model = model.to(DEV)
    img, _ = patterns.synth_batch(1, 14, 160, 7)
    disk = img[0, :, :150, :].to(DEV)                                # (14, 150, 160): H = 150, W = 160
    for block, overlap in ((64, 0), (64, 24)):
        offs = O.gen_tiles_offs(160, 150, block, overlap)
        assert tiles.gen_tiles_offs(160, 150, block, overlap) == offs
        ref = torch.zeros((150, 160), dtype=torch.int64, device=DEV)
        model.eval()
        with torch.no_grad():
            for (x, y) in offs:     # each tile scaled by its own min / max first (cloud_dataset.py:127-131), on the CPU oracle
                tile = torch.from_numpy(O.normalize_tile(disk[:, y:y + block, x:x + block].cpu().numpy())).to(DEV)
                lg = model(tile[None])
                ref[y:y + block, x:x + block] = torch.argmax(lg[0], dim=0)
        got = tiles.predict_disk(model, disk, block=block, overlap=overlap, tiles_per_forward=3)
        assert got.shape == ref.shape
        # eval-mode SwitchNorm has no batch coupling, so batching tiles changes nothing: exact equality
        assert torch.equal(got, ref)
        # without the per-tile scaling the raw values go in
        raw = tiles.predict_disk(model, disk, block=block, overlap=overlap, tiles_per_forward=3, normalize=None)
        ref2 = torch.zeros((150, 160), dtype=torch.int64, device=DEV)
        with torch.no_grad():
            for (x, y) in offs:
                ref2[y:y + block, x:x + block] = torch.argmax(model(disk[None, :, y:y + block, x:x + block])[0], dim=0)
        assert torch.equal(raw, ref2)
    # whole-disk scaling before tiling (utils/preddataset.py:42-56) == scaling the disk with the oracle, then raw tiles
    whole = torch.from_numpy(O.normalize_tile(disk.cpu().numpy())).to(DEV)
    assert torch.equal(tiles.predict_disk(model, disk, block=64, overlap=0, tiles_per_forward=3, normalize="minmax_disk"),
                       tiles.predict_disk(model, whole, block=64, overlap=0, tiles_per_forward=3, normalize=None))
    # a constant tile (0/0 under per-tile scaling) is classified like an all-zero tile, not by an arg-max over NaN
    flat = disk.clone()
    flat[:, :64, :64] = 3.0
    got = tiles.predict_disk(model, flat, block=64, overlap=0, tiles_per_forward=3)
    with torch.no_grad():
        want = torch.argmax(model(torch.zeros(1, 14, 64, 64, device=DEV))[0], dim=0)
    assert torch.equal(got[:64, :64], want)


def test_tile_minmax_scale_and_label_remap_bit_exact():
    """The HIP tile preparation against the numpy lines of the reference's dataset (cloud_dataset.py:127-134, restated in
    the oracle): per-tile min-max x 255 with NaN -> 0 and the label remap, bit for bit - batches of tiles, a tile whose
    size is not a multiple of the vector width, an unaligned view, NaNs, a constant tile (0/0 -> NaN like numpy), the
    bf16 output rounded from the float32 result, every label dtype."""
    from snws import data
    rng = np.random.default_rng(17)
    for shape in ((3, 14, 64, 64), (1, 14, 37, 41), (2, 3, 5, 7), (14, 128, 128)):
        a = (rng.standard_normal(shape) * 40.0 + 300.0).astype(np.float32)
        a.reshape(-1)[::97] = np.nan
        if len(shape) == 4 and shape[0] > 1:
            a[1] = 7.5                                      # a constant tile
        got = data.normalize_tiles(torch.from_numpy(a).to(DEV))
        tiles_np = a if a.ndim == 4 else a[None]
        want = np.stack([O.normalize_tile(t) for t in tiles_np]).reshape(shape)
        # (value equality with NaN == NaN: the constant tile's 0/0 carries the host's / the device's NaN sign bit)
        assert np.array_equal(got.cpu().numpy(), want, equal_nan=True), shape
        got16, mm = data.normalize_tiles(torch.from_numpy(a).to(DEV), torch.bfloat16, return_minmax=True)
        assert np.array_equal(got16.float().cpu().numpy(), torch.from_numpy(want).to(torch.bfloat16).float().numpy(),
                              equal_nan=True), shape
        clean = np.where(np.isnan(tiles_np), 0, tiles_np).reshape(tiles_np.shape[0], -1)
        assert np.array_equal(mm.cpu().numpy(), np.stack([clean.min(1), clean.max(1)], 1))
    odd = torch.from_numpy((rng.standard_normal(14 * 33 * 35 + 1) * 3).astype(np.float32)).to(DEV)[1:].view(14, 33, 35)
    assert np.array_equal(data.normalize_tiles(odd).cpu().numpy(), O.normalize_tile(odd.cpu().numpy()))
    lab = rng.integers(0, 12, size=(2, 33, 47)).astype(np.float32)
    lab[0, ::5, ::3] = 255.0
    lab[1, 1::7, ::2] = np.nan
    for dt in (np.float32, np.float64, np.int64, np.int32, np.uint8):
        src = np.nan_to_num(lab, nan=255.0).astype(dt) if np.dtype(dt).kind != "f" else lab.astype(dt)
        got = data.remap_labels(torch.from_numpy(src).to(DEV))
        assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), O.remap_label(src)), dt
    with pytest.raises(RuntimeError):
        data.normalize_tiles(torch.zeros(3, 4, 4))           # CPU tensor: no fallback


def test_comm_single_rank_roundtrip():
    """The C ABI's own RCCL collectives on a one-rank communicator: all-reduce(sum / max) leaves the buffers as they
    are.  (More ranks need more GPUs than the test box has; the multi-rank path is the same call.)"""
    from snws.comm import Comm
    c = Comm(Comm.unique_id(), 0, 1)
    a = torch.arange(1000, dtype=torch.float32, device=DEV) * 0.5
    b = torch.arange(100, dtype=torch.int64, device=DEV) - 50
    a0, b0 = a.clone(), b.clone()
    c.all_reduce(a)
    c.all_reduce(b, "sum")
    c.all_reduce(b, "max")
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)
    with pytest.raises(TypeError):
        c.all_reduce(a, "max")
    c.close()


def test_upconv_separable_kernel_all_factors():
    """The separable up-sampled-conv forward kernel is the default for k >= 8 only; with SNWS_UPCONV_SEP_ALL=1 (read
    once per process, hence the child process) it also takes k = 2 and 4.  Its output must equal the per-pixel
    gather kernel's (float32: 1e-6) for every factor, including heights/widths that are not multiples of its 8x32
    strip."""
    import subprocess
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(%r, "semantic-segmentation_amd"))
from snws import _lib as L
out = {}
torch.manual_seed(0)
for k, h, w in ((2, 3, 5), (2, 9, 20), (4, 3, 2), (4, 5, 9), (8, 2, 5), (16, 1, 3)):
    Co = 16
    Z = torch.randn(2, h, w, 9 * Co, device="cuda")
    b = torch.randn(Co, device="cuda")
    y = torch.empty(2, h * k, w * k, Co, device="cuda")
    L.call("snws_upconv_fwd", Z.data_ptr(), 9 * Co, y.data_ptr(), Co, b.data_ptr(), 0, 2, h, w, Co, k, 3, 0)
    torch.cuda.synchronize()
    out["k%%d_%%d_%%d" %% (k, h, w)] = y.cpu().numpy()
np.savez(sys.argv[1], **out)
""" % ROOT
    import tempfile
    res = []
    with tempfile.TemporaryDirectory() as d:
        for i, env_extra in enumerate(({"SNWS_UPCONV_GATHER": "1"}, {"SNWS_UPCONV_SEP_ALL": "1"})):
            env = dict(os.environ)
            env.update(env_extra)
            path = os.path.join(d, "o%d.npz" % i)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=200)
            res.append(dict(np.load(path)))
    assert set(res[0]) == set(res[1]) and len(res[0]) == 6
    for key in res[0]:
        a, b = res[0][key], res[1][key]
        assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max(), key


def test_flat_adamw_matches_torch_adamw():
    """snws.optim.FlatAdamW == torch.optim.AdamW step for step (decoupled decay, bias correction), including a
    parameter that never receives a gradient (left untouched, as torch does), and the two exchange state_dicts."""
    from snws.optim import FlatAdamW
    torch.manual_seed(8)
    shapes = [(7, 3, 3, 3), (7,), (1, 5, 1, 1), (3,), (10, 7, 1, 1), (2,)]
    pa = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = torch.optim.AdamW(pa, lr=2e-3, weight_decay=5e-4)
    ob = FlatAdamW(pb, lr=2e-3, weight_decay=5e-4)
    for it in range(6):
        lr = 2e-3 * (1 - it / 10.0) ** 0.9
        oa.param_groups[0]["lr"] = lr
        ob.param_groups[0]["lr"] = lr
        oa.zero_grad(set_to_none=True)
        ob.zero_grad()
        ws_ = [torch.randn(s, device=DEV) for s in shapes]
        la = sum((p * w).sum() + (p * p).sum() for p, w in zip(pa[:-1], ws_))     # the last parameter is unused
        lb = sum((p * w).sum() + (p * p).sum() for p, w in zip(pb[:-1], ws_))
        la.backward()
        lb.backward()
        oa.step()
        ob.step()
        if it == 2:      # hand the state over in both directions mid-run
            sa, sb = oa.state_dict(), ob.state_dict()
            assert set(sb["state"].keys()) == set(sa["state"].keys())
            ob.load_state_dict(sa)
            oa.load_state_dict(sb)
    for a, b in zip(pa, pb):
        assert relerr(b, a) < 2e-6
    assert torch.equal(pb[-1], pa[-1])
    sd = ob.state_dict()
    assert 5 not in sd["state"] and float(sd["state"][0]["step"]) == 6.0


def test_trainer_flat_optimizer_matches_torch_optimizer(tmp_path):
    """Three Trainer steps with the flat fused AdamW equal three steps with torch's AdamW; the checkpoint written after
    them (reference format) restores model, optimizer and counters into a fresh Trainer that then continues
    identically."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import harness, checkpoint
    img, lab = patterns.synth_batch(2, 14, 64, 101)
    img, lab = img.to(DEV), lab.to(DEV)
    models = []
    for kind in ("torch", "flat"):
        model = UNet3p_SNws(14, 4, 10, 1, 1)
        patterns.det_init(model)
        model = model.to(DEV)
        tr = harness.Trainer(model, optimizer=kind, total_iters=50)
        for _ in range(3):
            tr.step(img, lab)
        models.append((model, tr))
    for (n, a), (_, b) in zip(models[0][0].named_parameters(), models[1][0].named_parameters()):
        # Adam moves every element by ~lr per step whatever the gradient's size, so float32 noise in a near-zero
        # gradient shows up as a difference of the order of lr * steps = 6e-3 (absolute; weights are O(0.5))
        assert float((a - b).abs().max()) < 8e-3, n
    model, tr = models[1]
    path = str(tmp_path / "checkpoint.pth.tar")
    checkpoint.save_checkpoint(path, model, tr.opt, epoch=7, best_pred=0.625)
    ck = torch.load(path, weights_only=True)
    assert set(ck.keys()) == {"epoch", "state_dict", "optimizer", "best_pred"}
    assert list(ck["state_dict"].keys()) == list(model.state_dict().keys())
    fresh = UNet3p_SNws(14, 4, 10, 1, 1).to(DEV)
    tr2 = harness.Trainer(fresh, optimizer="flat", total_iters=50)
    epoch, best = checkpoint.load_checkpoint(path, fresh, tr2.opt)
    tr2.it = tr.it
    assert (epoch, best) == (7, 0.625)
    l1 = float(tr.step(img, lab))
    l2 = float(tr2.step(img, lab))
    assert abs(l1 - l2) <= 1e-5 * abs(l1)
    for (n, a), (_, b) in zip(model.named_parameters(), fresh.named_parameters()):
        assert relerr(b, a) < 1e-5, n


@pytest.mark.parametrize("shape,dtype,relu,tol", [((3, 8, 9, 7), torch.float32, False, 1e-4), ((2, 20, 16, 16), torch.float32, True, 1e-4),
                                                  ((2, 64, 32, 32), torch.bfloat16, True, 2e-2), ((1, 5, 6, 6), torch.float32, True, 1e-4)])
def test_hip_batchnorm_vs_torch(shape, dtype, relu, tol):
    """blocks.HipBatchNorm2d (+ fused ReLU) == nn.BatchNorm2d (+ ReLU): output, input / gamma / beta gradients, running
    buffers after two training calls, num_batches_tracked, and the eval-mode output."""
    from modeling.blocks import HipBatchNorm2d
    torch.manual_seed(9)
    N, C, H, W = shape
    ref = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        ref.weight.copy_(torch.rand(C) + 0.5)
        ref.bias.copy_(torch.randn(C) * 0.3)
    m = HipBatchNorm2d(C, fuse_relu=relu).to(DEV)
    m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    x = (torch.randn(shape) * 1.7 + 0.4).to(dtype)
    dy = torch.randn(shape).to(dtype)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy.double())
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = m(xg)
    y.backward(dy.to(DEV))
    assert y.dtype == dtype
    assert relerr(y.float(), yr) < tol and relerr(xg.grad.float(), xr.grad) < tol
    assert relerr(m.weight.grad, ref.weight.grad) < tol and relerr(m.bias.grad, ref.bias.grad) < tol
    with torch.no_grad():
        ref(xr.detach() * 1.5 - 0.2)
        m(xg.detach() * 1.5 - 0.2)
    assert relerr(m.running_mean, ref.running_mean) < max(tol, 1e-3) and relerr(m.running_var, ref.running_var) < max(tol, 1e-3)
    assert int(m.num_batches_tracked) == int(ref.num_batches_tracked) == 2
    ref.eval(); m.eval()
    with torch.no_grad():
        ye = ref(xr.detach())
        ye = torch.relu(ye) if relu else ye
        assert relerr(m(xg.detach()).float(), ye) < tol
    m.train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, C, 4, 4))


# ================================================================== SwitchNorm backward pass 1 in the dgrad epilogue
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, 64, 40, 50, 3), (1, 128, 320, 16, 32, 3), (2, 48, 64, 33, 20, 3),
                                   (2, 128, 64, 32, 32, 1)])
def test_dgrad_epilogue_sn_bwd_partials_match_reduce_kernel(shape):
    """snws_conv_mfma_dgrad_sn == snws_conv_mfma_dgrad for dx (bit-exact), and its per-tile partial sums, added up,
    equal those of snws_sn_bwd_reduce on that dx (f32 summation-order tolerance).  Two segments when C_in allows."""
    import ctypes
    from snws import _lib as L
    from snws import ops
    N, Cin, Cout, H, W, k = shape
    g = torch.Generator().manual_seed(5)
    lib = L.lib()
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    cfg = ops.ConvCfg(w.shape, 1, (k - 1) // 2, 1, 1)
    prep = ops.PreparedWeight(w.to(DEV), cfg, True, standardize=False)
    dy = torch.randn(N, H, W, Cout, generator=g).to(DEV).to(torch.bfloat16)
    # the SwitchNorm layers that produced the conv input: their inputs xin and forward coefficients
    segs = [(0, Cin)] if Cin < 128 else [(0, 64), (64, Cin - 64)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    T = lib.snws_conv_mfma_tiles(H, W)
    d = cfg.desc(N, H, W, Cin, Cin, Cout)
    dx0 = torch.empty(N, H, W, Cin, dtype=torch.bfloat16, device=DEV)
    L.call("snws_conv_mfma_dgrad", ctypes.byref(d), dy.data_ptr(), prep.pack_dgrad.data_ptr(), dx0.data_ptr(), st)
    dx1 = torch.empty_like(dx0)
    arr = (L.SnBwdSeg * len(segs))()
    keep = []
    for i, (c0, c) in enumerate(segs):
        xin = (torch.randn(N, H, W, c + 8, generator=g) * 2 + 0.5).to(DEV).to(torch.bfloat16)   # pixel stride c + 8
        ctx = torch.zeros(lib.snws_sn_ctx_floats(N, c), dtype=torch.float32, device=DEV)
        nc = N * c
        ctx[2 * nc:3 * nc] = torch.randn(nc, generator=g).to(DEV) * 0.3            # mean
        ctx[3 * nc:4 * nc] = torch.rand(nc, generator=g).to(DEV) + 0.5             # rstd
        ctx[4 * nc:5 * nc] = torch.randn(nc, generator=g).to(DEV)                  # scale (either sign)
        beta = torch.randn(c, generator=g).to(DEV)
        gamma = torch.ones(c, device=DEV)
        part = torch.full((N, T, c, 2), float("nan"), dtype=torch.float32, device=DEV)
        relu = 1 - (i & 1)
        arr[i] = L.SnBwdSeg(c0, c, xin.data_ptr(), c + 8, ctx.data_ptr(), beta.data_ptr(), part.data_ptr(), relu)
        keep.append((xin, ctx, beta, gamma, part, relu))
    L.call("snws_conv_mfma_dgrad_sn", ctypes.byref(d), dy.data_ptr(), prep.pack_dgrad.data_ptr(), dx1.data_ptr(), arr,
           len(segs), st)
    assert torch.equal(dx0, dx1)
    for (c0, c), (xin, ctx, beta, gamma, part, relu) in zip(segs, keep):
        Tr = lib.snws_sn_stats_tiles(N, H * W, c)
        ref = torch.empty((N, Tr, c, 2), dtype=torch.float32, device=DEV)
        L.call("snws_sn_bwd_reduce", dx0.data_ptr() + 2 * c0, Cin, xin.data_ptr(), c + 8, None, 0, L.SNWS_BF16,
               ctx.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, H * W, c, relu, ref.data_ptr(), st)
        a, b = part.double().sum(1), ref.double().sum(1)
        assert torch.isfinite(part).all()
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) < 1e-4 * scale, (c0, c, float((a - b).abs().max()), scale)


@pytest.mark.gpu
def test_fused_sn_bwd_model_gradients_match_unfused():
    """UNet3p_SNws(n_filters=64) backward with the dgrad-epilogue SwitchNorm reduction on and off.  The fused path is
    taken (hits counted); for EVERY layer that takes it, the epilogue's partial sums equal the stand-alone reduce
    kernel's on the very same gradient to f32 summation-order noise (the amplification-free check); and the
    parameter gradients of the two runs agree.  (This network's backward amplifies a 1e-6 perturbation of one
    layer's gradient ~5x per decoder stage - measured layer by layer with tools/fuse_probe.py: 8e-7 after the first
    SwitchNorm, 1e-2 at the first encoder conv - so the end-to-end bound is loose and the per-layer one tight.)"""
    import ctypes
    import statistics
    from modeling.UNet3p_SNws import UNet3p_SNws
    from snws import _lib as L
    from snws import ops
    torch.manual_seed(11)
    model = UNet3p_SNws(14, 64, 10, 1, 1).to(DEV).train()
    model.compute_dtype = torch.bfloat16
    img = (torch.rand(2, 14, 64, 64) * 255.0).to(DEV)
    lab = torch.randint(0, 10, (2, 64, 64)).to(DEV)
    orig = ops.fused_bwd_partials
    layer_err = []

    def checked(s, dys, lddy):
        r = orig(s, dys, lddy)
        if r is not None:
            part, T = r
            N, C, H, W = s["shape"]
            Tr = L.lib().snws_sn_stats_tiles(N, H * W, C)
            ref = torch.empty((N, Tr, C, 2), dtype=torch.float32, device=dys.device)
            L.call("snws_sn_bwd_reduce", dys.data_ptr(), lddy, s["xs"].data_ptr(), s["ldx"], None, 0, L.SNWS_BF16,
                   s["ctx"].data_ptr(), s["gamma"].data_ptr(), s["beta"].data_ptr(), N, H * W, C, int(s["relu"]),
                   ref.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            a, b = part.double().sum(1), ref.double().sum(1)
            layer_err.append(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30))
        return r

    grads, hits = {}, {}
    for fuse in (False, True):
        ops.FUSE_SN_BWD = fuse
        ops.FUSED_SN_BWD_HITS = 0
        ops.fused_bwd_partials = checked
        try:
            model.zero_grad(set_to_none=True)
            loss = torch.nn.functional.cross_entropy(model(img), lab) / 2
            loss.backward()
        finally:
            ops.FUSE_SN_BWD = True
            ops.fused_bwd_partials = orig
        hits[fuse] = ops.FUSED_SN_BWD_HITS
        grads[fuse] = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    assert hits[False] == 0
    # 20 decoder branches + 4 fusion convs + the first conv of 5 encoder stages + the centre's second conv; the last
    # fusion conv's reduction belongs to the fused classification head instead when that op is on
    want = 30 - int(ops.HEAD_FUSION)
    assert hits[True] == want and len(layer_err) == want, (hits, len(layer_err))
    assert max(layer_err) < 1e-5, layer_err
    errs = []
    for n in grads[True]:
        a, b = grads[True][n].double(), grads[False][n].double()
        errs.append(float((a - b).norm() / max(float(b.norm()), 1e-20)))
    print("fused vs unfused SwitchNorm backward: per-layer partial sums max rel diff %.2e; parameter gradients "
          "median %.2e worst %.2e" % (max(layer_err), statistics.median(errs), max(errs)))
    assert statistics.median(errs) < 2e-2 and max(errs) < 0.15


# ================================================================== LDS-staged up-sampled-conv forward (+ statistics)
@pytest.mark.gpu
def test_upconv_lds_kernel_matches_unstaged_and_emits_stats():
    """The LDS-staged forward (bf16, 3x3, 64-channel chunks; the default where it applies) against the kernels
    without LDS staging (SNWS_UPCONV_NOLDS=1, read once per process, hence the child process): agreement to one bf16
    rounding of the output.
    Sizes include strips cut by the image border, one-pixel sources and 128 channels.  snws_upconv_fwd_stats must
    store the same y and per-strip sums that add up to the moments of (y - bias) recomputed from y."""
    import subprocess
    import tempfile
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(%r, "semantic-segmentation_amd"))
from snws import _lib as L
out = {}
torch.manual_seed(0)
for k, h, w, Co in ((2, 3, 5, 64), (2, 20, 37, 64), (4, 3, 2, 64), (4, 9, 17, 128), (8, 2, 5, 64), (8, 5, 9, 64),
                    (16, 1, 3, 64), (16, 3, 4, 128), (2, 1, 1, 64), (16, 1, 1, 64)):
    Z = torch.randn(2, h, w, 9 * Co, device="cuda").to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    y = torch.empty(2, h * k, w * k, Co, device="cuda", dtype=torch.bfloat16)
    L.call("snws_upconv_fwd", Z.data_ptr(), 9 * Co, y.data_ptr(), Co, b.data_ptr(), 1, 2, h, w, Co, k, 3, 0)
    torch.cuda.synchronize()
    out["k%%d_%%d_%%d_%%d" %% (k, h, w, Co)] = y.float().cpu().numpy()
np.savez(sys.argv[1], **out)
""" % ROOT
    res = []
    with tempfile.TemporaryDirectory() as d:
        for i, env_extra in enumerate(({"SNWS_UPCONV_NOLDS": "1"}, {})):
            env = dict(os.environ)
            env.pop("SNWS_UPCONV_NOLDS", None)
            env.update(env_extra)
            path = os.path.join(d, "o%d.npz" % i)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=200)
            res.append(dict(np.load(path)))
    assert set(res[0]) == set(res[1]) and len(res[0]) == 10
    for key in res[0]:
        a, b = res[0][key], res[1][key]
        # other summation order (and packed-pair dot products for the column stage), then one bf16 rounding
        assert np.abs(a - b).max() <= 2 ** -7 * np.abs(a).max(), key
    # statistics entry point
    from snws import _lib as L
    lib = L.lib()
    torch.manual_seed(1)
    for k, h, w, Co in ((2, 20, 37, 64), (4, 9, 17, 128), (16, 3, 4, 64)):
        N, H, W = 2, h * k, w * k
        Z = torch.randn(N, h, w, 9 * Co, device=DEV).to(torch.bfloat16)
        b = torch.randn(Co, device=DEV)
        y0 = torch.empty(N, H, W, Co, device=DEV, dtype=torch.bfloat16)
        y1 = torch.empty_like(y0)
        T = lib.snws_upconv_stat_tiles(H, W)
        part = torch.full((N, T, Co, 2), float("nan"), device=DEV)
        L.call("snws_upconv_fwd", Z.data_ptr(), 9 * Co, y0.data_ptr(), Co, b.data_ptr(), 1, N, h, w, Co, k, 3, 0)
        L.call("snws_upconv_fwd_stats", Z.data_ptr(), 9 * Co, y1.data_ptr(), Co, b.data_ptr(), 1, N, h, w, Co, k, 3,
               part.data_ptr(), 0)
        assert torch.equal(y0, y1)
        d = y1.double() - b.double()
        s_ref = torch.stack([d.sum((1, 2)), (d * d).sum((1, 2))], -1)
        got = part.double().sum(1)
        assert torch.isfinite(part).all()
        assert float((got - s_ref).abs().max()) < 1e-5 * float(s_ref.abs().max())
    with pytest.raises(L.SnwsUnsupported):
        Z = torch.randn(1, 2, 2, 9 * 16, device=DEV).to(torch.bfloat16)
        y = torch.empty(1, 4, 4, 16, device=DEV, dtype=torch.bfloat16)
        part = torch.empty(1, 1, 16, 2, device=DEV)
        L.call("snws_upconv_fwd_stats", Z.data_ptr(), 144, y.data_ptr(), 16, None, 1, 1, 2, 2, 16, 2, 3, part.data_ptr(), 0)


@pytest.mark.gpu
def test_upconv_matrix_form_matches_vector_form():
    """The staged forward as small matrix products (snws_upconv_fwd_mode 2) against the same forward on the vector ALUs
    (mode 0), every k, strips cut by all four image borders, one-pixel sources, 128 channels, with and without bias:
    both forms take exact products of the same bf16 values and exactly represented interpolation weights, so they differ
    by the f32 summation order only - at most one bf16 unit in the last place of the output; the per-strip statistics
    must match the moments of the y each form stored.  A constant source map must come out exactly (weights sum to 1)."""
    from snws import _lib as L
    lib = L.lib()
    prev = lib.snws_upconv_fwd_mode(-1)
    try:
        torch.manual_seed(4)
        for k, h, w, Co, with_bias in ((2, 3, 5, 64, True), (2, 20, 37, 64, False), (4, 3, 2, 64, True), (4, 9, 17, 128, True),
                                       (8, 2, 5, 64, True), (8, 5, 9, 64, False), (16, 1, 3, 64, True), (16, 3, 4, 128, True),
                                       (2, 1, 1, 64, True), (16, 1, 1, 64, True), (4, 16, 24, 64, True), (8, 8, 12, 64, True)):
            N, H, W = 2, h * k, w * k
            Z = torch.randn(N, h, w, 9 * Co, device=DEV).to(torch.bfloat16)
            b = torch.randn(Co, device=DEV) if with_bias else None
            T = lib.snws_upconv_stat_tiles(H, W)
            ys, parts = [], []
            for mode in (0, 2):
                lib.snws_upconv_fwd_mode(mode)
                y = torch.full((N, H, W, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
                part = torch.full((N, T, Co, 2), float("nan"), device=DEV)
                L.call("snws_upconv_fwd_stats", Z.data_ptr(), 9 * Co, y.data_ptr(), Co, None if b is None else b.data_ptr(), 1,
                       N, h, w, Co, k, 3, part.data_ptr(), 0)
                y2 = torch.empty_like(y)
                L.call("snws_upconv_fwd", Z.data_ptr(), 9 * Co, y2.data_ptr(), Co, None if b is None else b.data_ptr(), 1, N,
                       h, w, Co, k, 3, 0)
                assert torch.equal(y, y2), (k, h, w, mode)
                d = y.double() - (0 if b is None else b.double())
                s_ref = torch.stack([d.sum((1, 2)), (d * d).sum((1, 2))], -1)
                assert torch.isfinite(part).all() and torch.isfinite(y.float()).all()
                assert float((part.double().sum(1) - s_ref).abs().max()) < 1e-5 * float(s_ref.abs().max()), (k, h, w, mode)
                ys.append(y.float())
            diff = (ys[0] - ys[1]).abs()
            assert float((diff / ys[0].abs().clamp_min(1e-3)).max()) <= 2 ** -7, (k, h, w, float(diff.max()))
            assert float((diff > 0).float().mean()) < 0.05, (k, h, w)          # and almost every value is identical
        # constant maps: every tap map constant c_t -> interior outputs are exactly sum_t c_t (+ bias)
        for k in (2, 4, 8, 16):
            h = w = 6
            Co = 64
            cvals = torch.tensor([0.5, -1.25, 2.0, 0.75, 3.0, -0.5, 1.5, 0.25, -2.0], device=DEV)
            Z = cvals.repeat_interleave(Co).to(torch.bfloat16).expand(1, h, w, 9 * Co).contiguous()
            for mode in (0, 2):
                lib.snws_upconv_fwd_mode(mode)
                y = torch.empty(1, h * k, w * k, Co, device=DEV, dtype=torch.bfloat16)
                L.call("snws_upconv_fwd", Z.data_ptr(), 9 * Co, y.data_ptr(), Co, None, 1, 1, h, w, Co, k, 3, 0)
                inner = y[:, 1:-1, 1:-1].float()
                assert float((inner - float(cvals.sum())).abs().max()) == 0.0, (k, mode)
    finally:
        lib.snws_upconv_fwd_mode(prev)


@pytest.mark.gpu
def test_upconv_backward_matrix_form_matches_vector_form():
    """snws_upconv_bwd for k = 2 and 4 as matrix products (snws_upconv_bwd_mode 1) against the separable vector-ALU kernel
    (mode 0): the same exact products (bf16 gradient x weights that are multiples of 1/16 or 1/64), another f32 summation order,
    one bf16 rounding.  Sources that are not multiples of the 4x16 tile, one-pixel and one-row sources, 128 channels, a
    strided dZ whose padding must stay untouched; and a constant gradient, for which interior entries are exact."""
    from snws import _lib as L
    lib = L.lib()
    prev = lib.snws_upconv_bwd_mode(-1)
    try:
        torch.manual_seed(8)
        for k, h, w, Co, pad in ((2, 20, 37, 64, 8), (2, 1, 1, 64, 0), (2, 3, 5, 64, 0), (2, 1, 40, 128, 0), (2, 9, 16, 64, 16),
                                 (2, 4, 17, 64, 0), (4, 9, 17, 128, 8), (4, 1, 1, 64, 0), (4, 3, 2, 64, 0), (4, 1, 7, 64, 0),
                                 (4, 8, 8, 64, 0), (4, 5, 13, 64, 16)):
            N = 2
            dy = torch.randn(N, h * k, w * k, Co, device=DEV).to(torch.bfloat16)
            outs = []
            for mode in (0, 1):
                lib.snws_upconv_bwd_mode(mode)
                dZ = torch.full((N, h, w, 9 * Co + pad), 7.0, device=DEV, dtype=torch.bfloat16)
                L.call("snws_upconv_bwd", dy.data_ptr(), Co, dZ.data_ptr(), 9 * Co + pad, 1, N, h, w, Co, k, 3, 0)
                if pad:
                    assert float((dZ[..., 9 * Co:].float() - 7.0).abs().max()) == 0.0
                outs.append(dZ[..., :9 * Co].float())
            assert torch.isfinite(outs[1]).all()
            diff = (outs[0] - outs[1]).abs()
            assert float(diff.max()) <= 2 ** -7 * float(outs[0].abs().max()), (h, w, Co, float(diff.max()))
            assert float((diff > 0).float().mean()) < 0.05, (h, w, Co)
        # constant gradient: every interior source pixel receives k*k output pixels' worth per tap
        h = w = 8
        for k in (2, 4):
            dy = torch.ones(1, k * h, k * w, 64, device=DEV, dtype=torch.bfloat16)
            for mode in (0, 1):
                lib.snws_upconv_bwd_mode(mode)
                dZ = torch.empty(1, h, w, 9 * 64, device=DEV, dtype=torch.bfloat16)
                L.call("snws_upconv_bwd", dy.data_ptr(), 64, dZ.data_ptr(), 9 * 64, 1, 1, h, w, 64, k, 3, 0)
                assert float((dZ[:, 2:-2, 2:-2].float() - float(k * k)).abs().max()) == 0.0, (k, mode)
    finally:
        lib.snws_upconv_bwd_mode(prev)


@pytest.mark.gpu
def test_upconv_separable_backward_matches_gather_backward():
    """The separable backward (bf16, 3x3; window columns split over 1/2/4/8 lanes for k = 2/4/8/16) against the
    per-source-pixel gather kernel (SNWS_UPCONV_BWD_GATHER=1, read once per process, hence the child process): the
    same f32 sums in another order, then one bf16 rounding.  Odd sizes, one-pixel sources, 8..128 channels, a
    strided dZ; plus the adjoint identity <up(Z), dy> = <Z, up^T(dy)> against the forward kernel."""
    import subprocess
    import tempfile
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(%r, "semantic-segmentation_amd"))
from snws import _lib as L
out = {}
torch.manual_seed(0)
for k, h, w, Co in ((2, 3, 5, 8), (2, 20, 37, 64), (4, 3, 2, 16), (4, 9, 17, 128), (8, 2, 5, 64), (8, 5, 9, 24),
                    (16, 1, 3, 64), (16, 3, 4, 16), (2, 1, 1, 64), (16, 1, 1, 8), (4, 1, 7, 64)):
    dy = torch.randn(2, h * k, w * k, Co, device="cuda").to(torch.bfloat16)
    dZ = torch.zeros(2, h, w, 9 * Co + 8, device="cuda", dtype=torch.bfloat16)
    L.call("snws_upconv_bwd", dy.data_ptr(), Co, dZ.data_ptr(), 9 * Co + 8, 1, 2, h, w, Co, k, 3, 0)
    torch.cuda.synchronize()
    out["k%%d_%%d_%%d_%%d" %% (k, h, w, Co)] = dZ.float().cpu().numpy()
np.savez(sys.argv[1], **out)
""" % ROOT
    res = []
    with tempfile.TemporaryDirectory() as d:
        for i, env_extra in enumerate(({"SNWS_UPCONV_BWD_GATHER": "1"}, {})):
            env = dict(os.environ)
            env.pop("SNWS_UPCONV_BWD_GATHER", None)
            env.update(env_extra)
            path = os.path.join(d, "o%d.npz" % i)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=200)
            res.append(dict(np.load(path)))
    assert set(res[0]) == set(res[1]) and len(res[0]) == 11
    for key in res[0]:
        a, b = res[0][key], res[1][key]
        assert np.all(b[..., -8:] == 0), key                      # the stride padding is left alone
        assert np.abs(a - b).max() <= 2 ** -7 * np.abs(a).max(), (key, np.abs(a - b).max(), np.abs(a).max())
    # adjoint identity in this process (default kernels), f64 inner products of the bf16 tensors
    from snws import _lib as L
    torch.manual_seed(2)
    for k, h, w, Co in ((2, 6, 9, 64), (4, 5, 3, 64), (8, 3, 2, 64), (16, 2, 2, 64)):
        Z = torch.randn(1, h, w, 9 * Co, device=DEV).to(torch.bfloat16)
        dy = torch.randn(1, h * k, w * k, Co, device=DEV).to(torch.bfloat16)
        y = torch.empty_like(dy)
        dZ = torch.empty_like(Z)
        L.call("snws_upconv_fwd", Z.data_ptr(), 9 * Co, y.data_ptr(), Co, None, 1, 1, h, w, Co, k, 3, 0)
        L.call("snws_upconv_bwd", dy.data_ptr(), Co, dZ.data_ptr(), 9 * Co, 1, 1, h, w, Co, k, 3, 0)
        lhs = float((y.double() * dy.double()).sum())
        rhs = float((Z.double() * dZ.double()).sum())
        scale = float(y.double().norm() * dy.double().norm())
        assert abs(lhs - rhs) < 5e-3 * scale, (k, lhs, rhs, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,with_bias,sliced", [(256, 128, True, False), (512, 256, False, False),
                                                       (128, 320, True, True)])
def test_wide_plain_1x1_conv_gemm_route_matches_conv_kernel(cin, cout, with_bias, sliced):
    """Plain 1x1 convolutions with >= 128 channels on both sides run forward and dgrad on the wide-tile 1x1 kernel
    (snws_gemm1x1) over the NHWC pixels; the result must equal the library's 64-channel-tile MFMA conv kernel route (same bf16 operands, f32 accumulation)
    within the bf16 bar, and the float64 product within 2e-2.  `sliced`: the input is a channel slice of a wider
    NHWC buffer (pixel stride > C_in)."""
    from snws import ops
    torch.manual_seed(3)
    N, H, W = 2, 24, 40
    wide = torch.randn(N, H, W, cin + 64, device=DEV).to(torch.bfloat16)
    x0 = (wide[..., 64:] if sliced else wide[..., :cin].contiguous()).permute(0, 3, 1, 2)
    w0 = (torch.randn(cout, cin, 1, 1, device=DEV) / cin ** 0.5)
    b0 = torch.randn(cout, device=DEV) if with_bias else None
    dy = torch.randn(N, H, W, cout, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
    out = {}
    prev = ops.GEMM_1X1
    try:
        for gemm in (True, False):
            ops.GEMM_1X1 = gemm
            x = x0.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) if not sliced else \
                x0.detach().requires_grad_(True)
            w = w0.clone().requires_grad_(True)
            b = None if b0 is None else b0.clone().requires_grad_(True)
            ops.PROFILE, tags = [], None
            y = ops.PlainConvFn.apply(x, w, b)
            y.backward(dy)
            tags = {t[0] for t in ops.PROFILE}
            ops.PROFILE = None
            assert ("gemm1x1_fwd" in tags and "gemm1x1_dgrad" in tags) == gemm, tags
            out[gemm] = (y.detach().float(), x.grad.float(), w.grad.float(), None if b is None else b.grad.float())
    finally:
        ops.GEMM_1X1 = prev
        ops.PROFILE = None
    xd, wd = x0.double(), w0.to(torch.bfloat16).double().view(cout, cin)
    ref = torch.einsum("nchw,oc->nohw", xd, wd) + (0 if b0 is None else b0.double().view(1, -1, 1, 1))
    refdx = torch.einsum("nohw,oc->nchw", dy.double(), wd)
    refdw = torch.einsum("nohw,nchw->oc", dy.double(), xd).view(cout, cin, 1, 1)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    for gemm in (True, False):
        assert rel(out[gemm][0], ref) < 2e-2 and rel(out[gemm][1], refdx) < 2e-2 and rel(out[gemm][2], refdw) < 1e-3
    assert rel(out[True][0], out[False][0]) < 5e-3 and rel(out[True][1], out[False][1]) < 5e-3
    assert rel(out[True][2], out[False][2]) < 1e-5
    if with_bias:
        assert rel(out[True][3], dy.double().sum((0, 2, 3))) < 1e-4 and rel(out[True][3], out[False][3]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,ldx_extra,bias", [
    (256, 256, 64, 0, False),          # one tile, one step
    (1000, 320, 320, 0, True),         # ragged pixel tail, 160-wide channel tiles
    (70000, 576, 128, 64, False),      # more tiles than workgroups (persistent loop), 192-wide tiles, strided input
    (5000, 1024, 512, 0, True),        # four channel tiles per pixel tile
    (513, 64, 1024, 0, True),          # narrowest output, deepest contraction
    (3000, 72, 256, 0, True),          # channel count that is not a multiple of 16
    (4096, 512, 256, 0, False),
])
def test_gemm1x1_kernel_vs_float64(M, N, K, ldx_extra, bias):
    """snws_gemm1x1 through the C ABI against the float64 product of the same bf16 operands: every output within
    half a bf16 step of the exact value plus f32 accumulation noise (relative L2 < 3e-3, worst element < 1e-2 of the
    largest); rows past M and columns past N of the padded tiles are never written."""
    import ctypes
    from snws import _lib as L
    torch.manual_seed(M + N + K)
    ldx = K + ldx_extra
    xbuf = torch.randn(M, ldx, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=DEV) if bias else None
    ldy = N + 8
    y = torch.full((M + 3, ldy), 7.0, device=DEV, dtype=torch.bfloat16)
    L.call("snws_gemm1x1", xbuf.data_ptr(), ldx, w.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), ldy, M, N, K,
           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = xbuf[:, :K].double() @ w.double().t() + (0 if b is None else b.double())
    got = y[:M, :N].double()
    assert float((got - ref).norm() / ref.norm()) < 3e-3
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-2
    assert bool((y[M:] == 7.0).all()) and bool((y[:, N:] == 7.0).all())


def test_gemm1x1_rejects_unsupported_shapes():
    from snws import _lib as L
    lib = L.lib()
    assert lib.snws_gemm1x1_supported(320, 320) == 1 and lib.snws_gemm1x1_supported(576, 1024) == 1
    assert lib.snws_gemm1x1_supported(16, 320) == 0 and lib.snws_gemm1x1_supported(320, 14) == 0
    assert lib.snws_gemm1x1_supported(100, 64) == 0
    x = torch.zeros(64, 96, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(L.SnwsUnsupported):
        L.call("snws_gemm1x1", x.data_ptr(), 96, x.data_ptr(), None, x.data_ptr(), 96, 64, 64, 96, None)


@pytest.mark.parametrize("arch", ["unet3p", "unet"])
def test_models_non_square_ragged_input_vs_oracle(arch):
    """Whole models on a non-square 48 x 80 input (deepest map 3 x 5: every MFMA / up-sampling / pooling kernel meets
    partial tiles in both directions), batch 3, n_filters 16.  float32 tier: logits, loss within 1e-3 of the float32
    oracle.  bf16 tier: logits within 2e-2 (L2) of the oracle with the same bf16 storage points; every gradient finite."""
    from modeling.UNet3p_SNws import UNet3p_SNws
    from modeling.UNet_SNws import UNet_SNws
    torch.manual_seed(21)
    cls = UNet3p_SNws if arch == "unet3p" else UNet_SNws
    model = cls(14, 16, 10, 1, 1)
    for p in model.parameters():
        if p.dim() == 1 and p.numel() == 3:
            p.data.copy_(torch.randn(3) * 0.5)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img = bf16_round(torch.rand(3, 14, 48, 80) * 255.0)
    lab = torch.randint(0, 10, (3, 48, 80)).float()
    lab[2, 40:] = 255.0
    logits_ref, loss_ref, _ = O.model_step(arch, {k: v.clone() for k, v in state.items()}, img, lab)
    logits_r16, _, _ = O.model_step(arch, {k: v.clone() for k, v in state.items()}, img, lab, act_round=torch.bfloat16)

    def l2(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / max(float(b.norm()), 1e-12))

    model = model.to(DEV).train()
    for dtype in (torch.float32, torch.bfloat16):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        model.compute_dtype = dtype
        logits = model(img.to(DEV))
        assert tuple(logits.shape) == (3, 10, 48, 80)
        loss = torch.nn.functional.cross_entropy(logits.float(), lab.to(DEV).long(), ignore_index=255) / 3
        loss.backward()
        for name, p in model.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        if dtype == torch.float32:
            assert relerr(logits, logits_ref) < 1e-3
            assert abs(float(loss.detach()) - loss_ref) < 1e-3 * abs(loss_ref)
        else:
            noise = l2(logits_r16, logits_ref)
            print("%s 48x80 bf16: GPU vs bf16-storage oracle %.3g; storage noise %.3g" % (arch, l2(logits, logits_r16), noise))
            assert l2(logits, logits_r16) < max(2e-2, 0.6 * noise)
            assert abs(float(loss.detach()) - loss_ref) < 2e-2 * abs(loss_ref)


def test_unet_snws_odd_input_zero_padded_skips_golden():
    """UNet_SNws on a 75 x 75 input (level sizes 75 / 38 / 19 / 10 / 5): every up-sampled decoder map is larger than
    its skip and the reference's CropConcat (modeling/UNet_SNws.py:35-39) zero-pads the skip; output 80 x 80.
    float32 tier against the reference's float64 run (fixture g4_unet_odd): logits 1e-3; each gradient 1e-3 + 3x the
    distance of the reference's own float32 run recorded in the fixture.  bf16 tier (n_filters 16, MFMA kernels on
    odd-sized maps): logits within 2e-2 (L2) of the oracle with the same storage points."""
    from modeling.UNet_SNws import UNet_SNws
    g = load("g4_unet_odd")
    model = UNet_SNws(14, 4, 10, 1, 1)
    patterns.det_init(model)
    model = model.to(DEV).train()
    img, _ = patterns.synth_batch(2, 14, int(g["in_size"]), 131)
    logits = model(img.to(DEV))
    assert tuple(logits.shape) == (2, 10, 80, 80)
    dy = patterns.t32(patterns.fill(tuple(logits.shape), 133, 1.0, 0.1))
    (logits * dy.to(DEV)).sum().backward()
    assert relerr(logits, g["logits"]) < 1e-3 + 3 * float(g["gap:logits"])
    for name, p in model.named_parameters():
        tol = 1e-3 + 3 * float(g["gap:grad:" + name])
        assert relerr(p.grad, g["grad:" + name]) < tol, (name, relerr(p.grad, g["grad:" + name]), tol)
    for name, b in model.named_buffers():
        assert relerr(b, g["buf:" + name]) < 1e-3, name
    # bf16 / MFMA tier on the same odd sizes
    torch.manual_seed(23)
    model = UNet_SNws(14, 16, 10, 1, 1)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img = bf16_round(torch.rand(2, 14, 75, 75) * 255.0)
    dy = bf16_round(torch.randn(2, 10, 80, 80))
    logits_ref, _, _ = O.model_step("unet", {k: v.clone() for k, v in state.items()}, img, None, dlogits=dy)
    logits_r16, _, _ = O.model_step("unet", {k: v.clone() for k, v in state.items()}, img, None, dlogits=dy,
                                    act_round=torch.bfloat16)
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16
    logits = model(img.to(DEV))
    (logits * dy.to(DEV)).sum().backward()

    def l2(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / max(float(b.norm()), 1e-12))
    noise = l2(logits_r16, logits_ref)
    print("75x75 bf16: GPU vs bf16-storage oracle %.3g; storage noise %.3g" % (l2(logits, logits_r16), noise))
    assert l2(logits, logits_r16) < max(2e-2, 0.6 * noise)
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name


@pytest.mark.parametrize("dims,shape", [(1, (6, 10)), (1, (4, 10, 7)), (2, (3, 10, 5, 6)), (3, (2, 10, 3, 4, 5))])
def test_synchronized_batchnorm_all_ranks_vs_torch(dims, shape):
    """SynchronizedBatchNorm{1,2,3}d outside a process group == nn.BatchNorm{1,2,3}d (the reference's fallback,
    modeling/sync_batchnorm/batchnorm.py:40-43) on the HIP kernels for every input rank: output, gradients, running
    buffers after two training calls, eval output."""
    import modeling.sync_batchnorm as sb
    torch.manual_seed(13)
    C = shape[1]
    ref = {1: torch.nn.BatchNorm1d, 2: torch.nn.BatchNorm2d, 3: torch.nn.BatchNorm3d}[dims](C).double()
    with torch.no_grad():
        ref.weight.copy_(torch.rand(C) + 0.5)
        ref.bias.copy_(torch.randn(C) * 0.3)
    m = {1: sb.SynchronizedBatchNorm1d, 2: sb.SynchronizedBatchNorm2d, 3: sb.SynchronizedBatchNorm3d}[dims](C).to(DEV)
    m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    x = torch.randn(shape) * 1.7 + 0.4
    dy = torch.randn(shape)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double())
    xg = x.to(DEV).requires_grad_(True)
    y = m(xg)
    assert tuple(y.shape) == tuple(shape)
    y.backward(dy.to(DEV))
    assert relerr(y, yr) < 1e-4 and relerr(xg.grad, xr.grad) < 1e-4
    assert relerr(m.weight.grad, ref.weight.grad) < 1e-4 and relerr(m.bias.grad, ref.bias.grad) < 1e-4
    with torch.no_grad():
        ref(xr.detach() * 1.5 - 0.2)
        m(xg.detach() * 1.5 - 0.2)
    assert relerr(m.running_mean, ref.running_mean) < 1e-4 and relerr(m.running_var, ref.running_var) < 1e-4
    assert int(m.num_batches_tracked) == 2
    ref.eval(); m.eval()
    with torch.no_grad():
        assert relerr(m(xg.detach()), ref(xr.detach())) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 14, 40, 72), (1, 3, 7, 5), (3, 20, 33, 64), (1, 44, 9, 11), (1, 50, 9, 11)])
def test_input_layout_kernel_matches_torch_route(shape, src):
    """snws_nchw_to_nhwc_pad (cast + NCHW -> NHWC + zero padding to 16 channels in one pass) against the stock route,
    bit for bit, and the first block of a model fed through it against the same block fed a channels-last tensor."""
    from modeling.blocks import ConvSnRelu, to_device_layout
    torch.manual_seed(3)
    x = (torch.randn(*shape, device=DEV) * 50).to(src)
    got = to_device_layout(x, torch.bfloat16)
    ref = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    assert torch.equal(got, ref)
    if shape[1] > 48:       # beyond the one-pass kernel's 48 padded channels: the stock cast + layout copy
        assert not hasattr(got, "_snws_padded")
        return
    full = got._snws_padded
    assert full.shape[1] % 16 == 0 and full.data_ptr() == got.data_ptr()
    assert float(full[:, shape[1]:].abs().max()) == 0.0
    blk = ConvSnRelu(shape[1], 64).to(DEV).train()
    y_fast = blk(got)
    y_ref = blk(ref)
    assert torch.equal(y_fast, y_ref)


@pytest.mark.gpu
def test_switchnorm_batched_launches_equal_single_calls():
    """ops.stat_batch + ops.assemble: five independent ConvSnRelu-style SwitchNorm calls, two of them through ONE module
    (so the batch needs two rounds), against the same five calls made one after the other: outputs, running
    statistics after both uses, input gradients and the parameter gradients of the modules used once are identical bit
    for bit; the twice-used module's parameter gradients are the same two terms summed in the other order."""
    import modeling.snorm as sn
    from snws import ops
    torch.manual_seed(12)
    N, C, H, W = 2, 16, 24, 40

    def run(batched):
        torch.manual_seed(13)
        mods = [sn.SwitchNorm2d(C).to(DEV).train() for _ in range(4)]
        for m in mods:
            m.fuse_relu = True
            with torch.no_grad():
                m.weight.copy_(torch.rand_like(m.weight) + 0.5)
                m.bias.copy_(torch.randn_like(m.bias) * 0.2)
                m.mean_weight.copy_(torch.randn(3, device=DEV))
                m.var_weight.copy_(torch.randn(3, device=DEV))
        use = [mods[0], mods[1], mods[1], mods[2], mods[3]]      # module 1 normalises two different inputs
        xs = [(torch.randn(N, C, H, W, device=DEV) * (1 + i) + 0.3 * i).to(torch.bfloat16)
              .contiguous(memory_format=torch.channels_last).requires_grad_(True) for i in range(5)]
        buf = ops.new_nhwc(N, 5 * C, H, W, torch.bfloat16, torch.device(DEV))
        old = ops.SN_BATCHING
        ops.SN_BATCHING = batched
        try:
            parts = []
            with ops.stat_batch(None):
                for i, (m, x) in enumerate(zip(use, xs)):
                    y, saved = ops.sn_forward(x.detach(), ops.SNParams(m), True, out=buf[:, i * C:(i + 1) * C])
                    parts.append((y, saved))
            torch.cuda.synchronize()
            outs = [y.detach().float().cpu().clone() for y, _ in parts]
            g = torch.randn(N, 5 * C, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            grads = []
            if batched:
                pairs = [(s, ops.sn_backward_pass1(g[:, i * C:(i + 1) * C], s, launch=False)) for i, (_, s) in enumerate(parts)]
                launches = [grp for rnd in ops._rounds(pairs, lambda sp: sp[0]["gamma"].data_ptr())
                            for grp in ops._groups(rnd, ops._sn_bwd_sig)]
                assert [len(grp) for grp in launches] == [4, 1]
                for grp in launches:
                    ops._sn_bwd_a(grp)
                for grp in launches:
                    ops._sn_bwd_b(grp)
                for i, (_, s) in enumerate(parts):
                    grads.append(ops.sn_backward(g[:, i * C:(i + 1) * C], s))
            else:
                for i, (_, s) in enumerate(parts):
                    grads.append(ops.sn_backward(g[:, i * C:(i + 1) * C], s))
            torch.cuda.synchronize()
        finally:
            ops.SN_BATCHING = old
        res = [[None if t is None else t.detach().float().cpu().clone() for t in gr] for gr in grads]
        stats = [(m.running_mean.detach().cpu().clone(), m.running_var.detach().cpu().clone()) for m in mods]
        return outs, res, stats

    o1, g1, s1 = run(True)
    o0, g0, s0 = run(False)
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    for (am, av), (bm, bv) in zip(s1, s0):
        assert torch.equal(am, bm) and torch.equal(av, bv)
    for i, (ga, gb) in enumerate(zip(g1, g0)):
        assert torch.equal(ga[0], gb[0]), i                      # dx
        for a, b in zip(ga[1:], gb[1:]):
            assert torch.equal(a, b), i                          # per-call parameter gradients (no flat buffer here)


@pytest.mark.gpu
def test_switchnorm_batched_launch_rejects_shared_buffers():
    import ctypes
    from snws import _lib as L
    N, C, HW = 2, 16, 64
    f = lambda n: torch.zeros(n, dtype=torch.float32, device=DEV)      # noqa: E731
    ctxs = [f(L.lib().snws_sn_ctx_floats(N, C)) for _ in range(2)]
    packs, gam, mw, vw, rm, rv = [f(2 * C + 1) for _ in range(2)], f(C), f(3), f(3), f(C), f(C)
    items = (L.SnFwdItem * 2)()
    for it, c, p in zip(items, ctxs, packs):
        it.ctx, it.pack, it.gamma, it.beta = c.data_ptr(), p.data_ptr(), gam.data_ptr(), gam.data_ptr()
        it.mean_weight, it.var_weight, it.running_mean, it.running_var = mw.data_ptr(), vw.data_ptr(), rm.data_ptr(), rv.data_ptr()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with pytest.raises(ValueError, match="share their running buffers"):
        L.call("snws_sn_mix_b_batched", items, 2, N, C, 1e-5, 0.997, 1, 1, 1, st)
    L.call("snws_sn_mix_b_batched", items, 2, N, C, 1e-5, 0.997, 0, 1, 1, st)      # evaluation mode writes no buffers
    with pytest.raises(ValueError, match="count"):
        L.call("snws_sn_mix_b_batched", items, 9, N, C, 1e-5, 0.997, 0, 1, 1, st)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_upconv_bwd_accumulating_form_equals_temporary_plus_add():
    """snws_upconv_bwd_acc (a later consumer of a shared tap map adds its share in the kernel's store epilogue) against
    snws_upconv_bwd into a temporary followed by snws_add_inplace: the same float32 sum of the same two bf16 values,
    rounded once - bit for bit; factors the in-place form does not take are refused, not silently overwritten."""
    import ctypes
    from snws import _lib as L
    torch.manual_seed(31)
    N, h, w, Co, k = 2, 13, 22, 64, 2
    dy = torch.randn(N, h * k, w * k, Co, device=DEV).to(torch.bfloat16)
    base = torch.randn(N, h, w, 9 * Co, device=DEV).to(torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    tmp = torch.empty_like(base)
    L.call("snws_upconv_bwd", P(dy), Co, P(tmp), 9 * Co, L.SNWS_BF16, N, h, w, Co, k, 3, st)
    want = base.clone()
    L.call("snws_add_inplace", P(want), 9 * Co, P(tmp), 9 * Co, L.SNWS_BF16, N * h * w, 9 * Co, st)
    got = base.clone()
    L.call("snws_upconv_bwd_acc", P(dy), Co, P(got), 9 * Co, L.SNWS_BF16, N, h, w, Co, k, 3, st)
    torch.cuda.synchronize()
    assert float(tmp.float().abs().max()) > 0
    assert torch.equal(got, want)
    dy4 = torch.randn(N, h * 4, w * 4, Co, device=DEV).to(torch.bfloat16)
    keep = got.clone()
    with pytest.raises(L.SnwsUnsupported):
        L.call("snws_upconv_bwd_acc", P(dy4), Co, P(got), 9 * Co, L.SNWS_BF16, N, h, w, Co, 4, 3, st)
    torch.cuda.synchronize()
    assert torch.equal(got, keep)


@pytest.mark.gpu
def test_switchnorm_batch_falls_back_for_unaligned_channel_counts():
    """Two SwitchNorm calls of 12 bf16 channels inside one stat_batch: the per-(image, channel) passes run batched, the
    normalise pass cannot (24-byte pixels are not 16-byte lanes) and falls back to the single launches - same result as
    without the batch."""
    import modeling.snorm as sn
    from snws import ops
    N, C, H, W = 2, 12, 9, 14
    torch.manual_seed(3)
    mods = [sn.SwitchNorm2d(C).to(DEV).train() for _ in range(2)]
    xs = [(torch.randn(N, C, H, W, device=DEV) * (i + 1)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
          for i in range(2)]
    want = [ops.sn_forward(x, ops.SNParams(m), True)[0].float().clone() for m, x in zip(mods, xs)]
    for m in mods:
        m.reset_parameters()
        m.running_mean.zero_()
        m.running_var.zero_()
    with ops.stat_batch(None):
        got = [ops.sn_forward(x, ops.SNParams(m), True)[0] for m, x in zip(mods, xs)]
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a.float(), b)
