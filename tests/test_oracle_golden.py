"""Pins oracle/snws_oracle.py against fixtures produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import snws_oracle as O
import patterns

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def close(a, b, rtol=1e-4, atol=1e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= atol * scale + rtol * scale, (err, scale)


# ---------------------------------------------------------------- Conv2dws
@pytest.mark.parametrize("ci", range(4))
def test_conv2dws_golden(ci):
    g = load("g1_conv2dws")
    pre = "c%d_" % ci
    Oc, I, k, s, p, d, has_b = [int(v) for v in g[pre + "cfg"]]
    w = torch.from_numpy(g[pre + "w"])
    x = torch.from_numpy(g[pre + "x"])
    b = torch.from_numpy(g[pre + "b"]) if has_b else None
    dy = torch.from_numpy(g[pre + "dy"])
    close(O.ws_standardize(w), g[pre + "what"])
    close(O.conv2dws_fwd(x, w, b, s, p, d), g[pre + "y"])
    dx, dw, db = O.conv2dws_bwd(x, w, dy, bool(has_b), s, p, d)
    close(dx, g[pre + "dx"])
    close(dw, g[pre + "dw"], rtol=2e-4)
    if has_b:
        close(db, g[pre + "db"])


def test_ws_bwd_matches_autograd_f64():
    w = torch.from_numpy(patterns.fill((6, 5, 3, 3), 3, 0.8)).double().requires_grad_(True)
    g = torch.from_numpy(patterns.fill((6, 5, 3, 3), 9, 1.0)).double()
    O.ws_standardize(w).backward(g)
    close(O.ws_standardize_bwd(w.detach(), g), w.grad, rtol=1e-11, atol=1e-12)


# ---------------------------------------------------------------- SwitchNorm
def _sn_cases():
    g = load("g2_switchnorm")
    return sorted({k[:-4] for k in g.files if k.endswith("_cfg")})


def _sn_state(g, pre):
    N, C, H, W, mov, ubn, lg = [int(v) for v in g[pre + "cfg"]]
    st = O.SNState(C, using_movavg=mov, using_bn=ubn, last_gamma=lg)
    st.gamma = torch.from_numpy(g[pre + "gamma"]).reshape(-1).clone()
    st.beta = torch.from_numpy(g[pre + "beta"]).reshape(-1).clone()
    st.mean_weight = torch.from_numpy(g[pre + "mw"]).clone()
    st.var_weight = torch.from_numpy(g[pre + "vw"]).clone()
    return st, ubn


@pytest.mark.parametrize("pre", _sn_cases())
def test_switchnorm_golden(pre):
    g = load("g2_switchnorm")
    pre = pre + "_"
    st, ubn = _sn_state(g, pre)
    x = torch.from_numpy(g[pre + "x"])
    dy = torch.from_numpy(g[pre + "dy"])
    y, ctx = O.switchnorm_fwd(x, st, training=True)
    close(y, g[pre + "y"])
    dx, dga, dbe, dmw, dvw = O.switchnorm_bwd(dy, st, ctx)
    close(dx, g[pre + "dx"], rtol=2e-4)
    close(dga, g[pre + "dgamma"].reshape(-1), rtol=2e-4)
    close(dbe, g[pre + "dbeta"].reshape(-1), rtol=2e-4)
    close(dmw, g[pre + "dmw"], rtol=5e-4, atol=1e-4)
    close(dvw, g[pre + "dvw"], rtol=5e-4, atol=1e-4)
    if ubn:
        close(st.running_mean, g[pre + "rm1"].reshape(-1))
        close(st.running_var, g[pre + "rv1"].reshape(-1))
        for j in range(3):
            O.switchnorm_fwd(x * (1.0 + 0.5 * (j + 1)) - 0.3 * j, st, training=True)
        close(st.running_mean, g[pre + "rm4"].reshape(-1))
        close(st.running_var, g[pre + "rv4"].reshape(-1))
    y_eval, _ = O.switchnorm_fwd(x, st, training=False)
    close(y_eval, g[pre + "y_eval"])


def test_switchnorm_bwd_matches_autograd_f64_with_relu():
    torch.manual_seed(0)
    N, C, H, W = 3, 5, 4, 6
    st = O.SNState(C, dtype=torch.float64)
    st.gamma = (torch.rand(C, dtype=torch.float64) + 0.5).requires_grad_(True)
    st.beta = torch.randn(C, dtype=torch.float64).requires_grad_(True)
    st.mean_weight = torch.randn(3, dtype=torch.float64).requires_grad_(True)
    st.var_weight = torch.randn(3, dtype=torch.float64).requires_grad_(True)
    x = torch.randn(N, C, H, W, dtype=torch.float64).requires_grad_(True)
    dy = torch.randn(N, C, H, W, dtype=torch.float64)
    y, ctx = O.switchnorm_fwd(x, st, training=True, relu=True)
    y.backward(dy)
    with torch.no_grad():
        dx, dga, dbe, dmw, dvw = O.switchnorm_bwd(dy, st, ctx)
    for a, b in ((dx, x.grad), (dga, st.gamma.grad), (dbe, st.beta.grad), (dmw, st.mean_weight.grad),
                 (dvw, st.var_weight.grad)):
        close(a, b, rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("shards", [2, 4])
def test_switchnorm_sharded_equals_global_batch(shards):
    """Data-parallel identity: shard-local SwitchNorm with all-reduced BN sums == the reference run on
    the whole batch (fixture s3 has N=4)."""
    g = load("g2_switchnorm")
    pre = "s3_bn_mov_"
    x = torch.from_numpy(g[pre + "x"]).double()
    dy = torch.from_numpy(g[pre + "dy"]).double()
    N = x.shape[0]
    per = N // shards
    sts = []
    for r in range(shards):
        st, _ = _sn_state(g, pre)
        for a in ("gamma", "beta", "mean_weight", "var_weight", "running_mean", "running_var"):
            setattr(st, a, getattr(st, a).double())
        sts.append(st)
    xs = [x[r * per:(r + 1) * per] for r in range(shards)]
    # forward: emulate the all-reduce by summing every shard's pack
    packs = []
    for r in range(shards):
        O.switchnorm_fwd(xs[r], sts[r], True, allreduce=lambda p: (packs.append(p.clone()), p)[1])
    total = torch.stack(packs).sum(0)
    for st, _ in [(_sn_state(g, pre))]:
        pass
    ys, ctxs = [], []
    sts = []
    for r in range(shards):
        st, _ = _sn_state(g, pre)
        for a in ("gamma", "beta", "mean_weight", "var_weight", "running_mean", "running_var"):
            setattr(st, a, getattr(st, a).double())
        sts.append(st)
        y, ctx = O.switchnorm_fwd(xs[r], st, True, allreduce=lambda p: total)
        ys.append(y)
        ctxs.append(ctx)
    close(torch.cat(ys), g[pre + "y"], rtol=1e-5)
    close(sts[0].running_mean, g[pre + "rm1"].reshape(-1), rtol=1e-5)
    close(sts[1].running_var, g[pre + "rv1"].reshape(-1), rtol=1e-5)
    # backward: first collect packs, then rerun with the total
    bpacks = []
    for r in range(shards):
        O.switchnorm_bwd(dy[r * per:(r + 1) * per], sts[r], ctxs[r], allreduce=lambda p: (bpacks.append(p.clone()), p)[1])
    btotal = torch.stack(bpacks).sum(0)
    outs = [O.switchnorm_bwd(dy[r * per:(r + 1) * per], sts[r], ctxs[r], allreduce=lambda p: btotal) for r in range(shards)]
    close(torch.cat([o[0] for o in outs]), g[pre + "dx"], rtol=2e-4)
    close(sum(o[1] for o in outs), g[pre + "dgamma"].reshape(-1), rtol=2e-4)
    close(sum(o[3] for o in outs), g[pre + "dmw"], rtol=5e-4, atol=1e-4)
    close(sum(o[4] for o in outs), g[pre + "dvw"], rtol=5e-4, atol=1e-4)


# ---------------------------------------------------------------- resampling vs stock torch
@pytest.mark.parametrize("k,hw", [(2, (8, 8)), (4, (8, 16)), (8, (16, 8)), (2, (5, 7)), (4, (9, 6))])
def test_maxpool_ceil(k, hw):
    x = torch.from_numpy(patterns.fill((2, 3) + hw, k, 1.0)).float()
    ref = torch.nn.functional.max_pool2d(x, k, k, ceil_mode=True)
    assert torch.equal(O.maxpool_ceil(x, k), ref)


@pytest.mark.parametrize("s", [2, 4, 8, 16])
def test_upsample_bilinear(s):
    x = torch.from_numpy(patterns.fill((2, 3, 4, 5), s, 1.0)).float()
    ref = torch.nn.functional.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
    close(O.upsample_bilinear(x, s), ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k,ks", [(2, 3), (4, 3), (16, 3), (8, 1)])
def test_conv_of_upsampled_commutes(k, ks):
    """The identity behind the GPU path's conv-of-upsample: contracting channels at the source resolution and
    interpolating afterwards equals convolving the up-sampled map (borders included)."""
    x = torch.from_numpy(patterns.fill((2, 5, 3, 4), k, 1.0)).double()
    w = torch.from_numpy(patterns.fill((6, 5, ks, ks), k + 1, 0.8)).double()
    b = torch.from_numpy(patterns.fill((6,), k + 2, 0.3)).double()
    what = O.ws_standardize(w)
    direct = torch.nn.functional.conv2d(O.upsample_bilinear(x, k), what, b, 1, (ks - 1) // 2)
    close(O.conv_of_upsampled_commuted(x, what, b, k), direct, rtol=1e-12, atol=1e-12)


# ---------------------------------------------------------------- whole models
def _state_from_pattern(arch, nf=4, dtype=torch.float32):
    st = O.make_state(arch, 14, nf, 10, dtype)

    class _M:  # adapter so patterns.det_init can walk "named_parameters" in the reference's order
        def named_parameters(self_inner):
            return [(k, v) for k, v in st.items() if not O.is_buffer(k)]
    (patterns.det_init_ocr if arch == "unet3p_res_ocr" else patterns.det_init)(_M())
    return st


def test_res_ocr_model_golden():
    """UNet3p_res_ocr_SNws: the fixture holds the reference run in float64 (this toy size amplifies float32 rounding
    to 1e-3..1e-1 of some gradients, recorded in the fixture as "gap:*"), so the oracle is run in float64 too and
    must agree to the float32 precision the fixture is stored in."""
    arch, g = "unet3p_res_ocr", load("g4_unet3p_res_ocr")
    st = _state_from_pattern(arch, dtype=torch.float64)
    assert list(st.keys()) == [str(k) for k in g["state_keys"]]
    assert [",".join(map(str, v.shape)) for v in st.values()] == [str(s) for s in g["state_shapes"]]
    img, lab = patterns.synth_batch(2, 14, 64, 101)
    logits, loss, grads = O.model_step(arch, st, img.double(), lab, training=True)
    close(logits, g["logits"], rtol=1e-6, atol=1e-6)
    assert abs(loss - float(g["loss"])) <= 1e-9 * abs(float(g["loss"]))
    for name in g["param_names"]:
        name = str(name)
        ref = g["grad:" + name]
        assert np.abs(grads[name].numpy() - ref).max() <= 1e-6 * np.abs(ref).max() + 1e-9, name
    for k in st:
        if O.is_buffer(k):
            close(st[k].double(), g["buf:" + k].astype(np.float64), rtol=1e-6)
    logits_eval, _, _ = O.model_step(arch, st, img.double(), lab, training=False)
    close(logits_eval, g["logits_eval"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("arch,tag", [("unet", "g4_unet"), ("unet3p", "g4_unet3p")])
def test_model_golden(arch, tag):
    g = load(tag)
    st = _state_from_pattern(arch)
    assert list(st.keys()) == [str(k) for k in g["state_keys"]]
    assert [",".join(map(str, v.shape)) for v in st.values()] == [str(s) for s in g["state_shapes"]]
    img, lab = patterns.synth_batch(2, 14, 64, 101)
    logits, loss, grads = O.model_step(arch, st, img, lab, training=True)
    close(logits, g["logits"], rtol=1e-3, atol=1e-4)
    assert abs(loss - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    for name in g["param_names"]:
        name = str(name)
        ref = g["grad:" + name]
        got = grads[name].numpy()
        denom = max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() <= 2e-3 * denom + 1e-6, name
    for k in st:
        if O.is_buffer(k):
            close(st[k].double(), g["buf:" + k].astype(np.float64), rtol=1e-4)
    logits_eval, _, _ = O.model_step(arch, st, img, lab, training=False)
    close(logits_eval, g["logits_eval"], rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------- evaluator
@pytest.mark.parametrize("case,C", [("a", 10), ("b", 10), ("e", 3)])
def test_evaluator_golden(case, C):
    g = load("g6_evaluator")
    gt, pr = g[case + "_gt"], g[case + "_pred"]
    cm = O.confusion_matrix(gt, pr, C)
    if case == "a":
        cm = cm + O.confusion_matrix(gt[::-1].copy(), pr, C)
    assert np.array_equal(cm.astype(np.float64), g[case + "_cm"])
    pa, pac, miou, fw = O.evaluator_metrics(cm)
    for got, key in ((pa, "_pa"), (pac, "_pac"), (miou, "_miou"), (fw, "_fwiou")):
        assert np.array_equal(np.asarray(got, np.float64).view(np.uint64), g[case + key].view(np.uint64)), key


def test_evaluator_all_ignored_and_full_disk():
    g = load("g6_evaluator")
    cm = O.confusion_matrix(np.full((1, 8, 8), 255.0, np.float32), np.zeros((1, 8, 8), np.int64), 10)
    assert np.array_equal(cm, g["c_cm"])
    pa, pac, miou, fw = O.evaluator_metrics(cm)
    assert np.isnan(pa) and np.isnan(g["c_pa"]) and np.isnan(miou) and np.isnan(g["c_miou"])
    assert fw == g["c_fwiou"] == 0.0
    rng = np.random.default_rng(int(g["d_seed"]))
    gt = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.float32)
    pr = rng.integers(0, 10, size=(1, 2748, 2748)).astype(np.int64)
    cm = O.confusion_matrix(gt, pr, 10)
    assert np.array_equal(cm.astype(np.float64), g["d_cm"])
    pa, pac, miou, fw = O.evaluator_metrics(cm)
    assert pa == g["d_pa"] and miou == g["d_miou"] and fw == g["d_fwiou"]


def test_evaluator_out_of_range_prediction_raises():
    with pytest.raises(ValueError):
        O.confusion_matrix(np.array([9.0]), np.array([10]), 10)


# ---------------------------------------------------------------- misc
def test_tiles_lr_ce_golden():
    g = load("g78_misc")
    for i in range(int(g["n_tiles_cases"])):
        args = [int(v) for v in g["tiles%d_args" % i]]
        assert np.array_equal(np.array(O.gen_tiles_offs(*args), np.int64), g["tiles%d" % i])
    for (i, ep), lr in zip(g["lr_pts"], g["lr_vals"]):
        assert O.poly_lr(0.002, int(ep) * 746 + int(i), 200 * 746) == lr
    logit = torch.from_numpy(g["ce_logit"]).requires_grad_(True)
    loss = O.ce_loss_batch_avg(logit, torch.from_numpy(g["ce_target"]))
    loss.backward()
    assert abs(float(loss) - float(g["ce_loss"])) < 1e-6
    close(logit.grad, g["ce_dlogit"], rtol=1e-5, atol=1e-7)


def test_weighted_ce_golden():
    """Class-weighted criterion (utils/loss.py:148 with the weights of train.py:55-67) against the reference's output."""
    g = load("g8w_weighted_ce")
    logit = torch.from_numpy(g["logit"]).requires_grad_(True)
    loss = O.ce_loss_batch_avg(logit, torch.from_numpy(g["target"]), weight=torch.from_numpy(g["weight"]))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    close(logit.grad, g["dlogit"], rtol=1e-5, atol=1e-7)


def test_loss_curve_golden_prefix():
    """First 12 steps of the 200-step AdamW/poly curve (the full curve is the GPU test's target)."""
    g = load("g5_loss_curve")
    st = _state_from_pattern("unet3p")
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if not O.is_buffer(k)}
    full = dict(st)
    full.update(params)
    opt = torch.optim.AdamW(list(params.values()), lr=2e-3, weight_decay=5e-4)
    batches = [patterns.synth_batch_learnable(4, 14, 64, 200 + j) for j in range(4)]
    for i in range(12):
        for gi in opt.param_groups:
            gi["lr"] = O.poly_lr(2e-3, i, 200)
        assert abs(opt.param_groups[0]["lr"] - float(g["lrs"][i])) < 1e-12
        opt.zero_grad()
        net = O.OracleNet(full)
        img, lab = batches[i % 4]
        loss = O.ce_loss_batch_avg(net.unet3p(img), lab)
        loss.backward()
        opt.step()
        assert abs(float(loss) - float(g["losses"][i])) <= 2e-3 * float(g["losses"][i]), i


# ---------------------------------------------------------------- bf16 storage mode
def test_bf16_storage_mode_matches_reference_run_with_bf16_storage():
    """The oracle's act_round=bfloat16 mode (what the bf16 GPU tier is compared with) against the REFERENCE model run
    with the same storage roundings applied from outside (fixture g9, tests/golden/gen_golden.py::gen_bf16_storage),
    in a trained state.  Two float32 evaluations of the same rounded network differ by summation order only, but a
    last-bit difference flips some bf16 roundings, and the fixture records how far the reference's own result moves
    when its parameters move by a relative 1e-6 ("noise_*"): logits 0.65 %, all gradients together 1.6 % (L2).  The
    oracle must be within that self-noise level of the reference: 1.5x for the logits / loss, 2x for the gradients;
    the plain float32 oracle must reproduce the fixture's float32 logits to 1e-3."""
    g = load("g9_bf16_storage")
    state, img, lab, ref_grads = patterns.load_bf16_storage_fixture(g)
    st = O.make_state("unet3p", 14, int(g["n_filters"]), 10)
    assert set(st.keys()) == set(state.keys())
    st = {k: state[k].reshape(st[k].shape).clone() for k in st}

    def l2(a, b):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        return float((a - b).norm() / b.norm())

    lo32, loss32, _ = O.model_step("unet3p", {k: v.clone() for k, v in st.items()}, img, lab)
    assert l2(lo32, g["logits_f32"]) < 1e-3 and abs(loss32 - float(g["loss_f32"])) < 1e-3 * float(g["loss_f32"])
    sb = {k: v.clone() for k, v in st.items()}
    lob, lossb, gb = O.model_step("unet3p", sb, img, lab, act_round=torch.bfloat16)
    noise_l, noise_g = float(g["noise_logits"]), float(g["noise_grads_all"])
    assert 1e-3 < noise_l < 2e-2 and 1e-3 < noise_g < 5e-2
    assert l2(lob, g["logits_bf16"]) < 1.5 * noise_l
    assert abs(lossb - float(g["loss_bf16"])) < 1.5 * noise_l * float(g["loss_bf16"])
    names = [str(n) for n in g["param_names"]]
    got = torch.cat([gb[n].flatten().double() for n in names])
    ref = torch.cat([ref_grads[n].flatten() for n in names])
    assert l2(got, ref) < 2.0 * noise_g
    for k in sb:
        if O.is_buffer(k):
            close(sb[k].double(), g["buf1:" + k].astype(np.float64), rtol=2e-2, atol=1e-4)


def test_bf16_storage_mode_res_ocr_matches_reference_run_with_bf16_storage():
    """The same pinning for UNet3p_res_ocr_SNws (fixture g9b, gen_golden.py::gen_bf16_storage_ocr: the reference model
    in a trained state, run with bf16 storage roundings applied from outside on every filter, convolution output,
    ReLU output, up-sampled map, both soft-max results, the region features and the attention context).  The
    expected values are the reference with FLOAT64 arithmetic between the roundings; the fixture also records how far
    that result moves after a relative 1e-6 change of every parameter ("noise_*": logits 0.7 %, the gradient vector
    26 %; per tensor 2 % at a few SwitchNorm mixing vectors, 10-80 % in the encoder) and how far the reference's own
    float32 arithmetic lands from it ("f32gap_*": 0.55 % / 16 %).  With bf16 storage this model's whole-network
    gradients are not reproducible by ANY two implementations, the reference's float32 and float64 runs included, so
    the oracle is gated on the logits / loss (1.5x self-noise, and the 2e-2 bar) and on the gradients only as far as
    the recorded self-noise allows (the whole vector, and the median over the tensors)."""
    g = load("g9b_bf16_storage_res_ocr")
    state, img, lab, ref_grads = patterns.load_bf16_storage_fixture(g)
    st = O.make_state("unet3p_res_ocr", 14, int(g["n_filters"]), 10)
    assert set(st.keys()) == set(state.keys())
    st = {k: state[k].reshape(st[k].shape).clone() for k in st}

    def l2(a, b):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        return float((a - b).norm() / max(float(b.norm()), 1e-30))

    lo32, loss32, _ = O.model_step("unet3p_res_ocr", {k: v.clone() for k, v in st.items()}, img, lab)
    assert l2(lo32, g["logits_f32"]) < 1e-3 and abs(loss32 - float(g["loss_f32"])) < 1e-3 * float(g["loss_f32"])
    sb = {k: v.clone() for k, v in st.items()}
    lob, lossb, gb = O.model_step("unet3p_res_ocr", sb, img, lab, act_round=torch.bfloat16)
    noise_l, noise_g = float(g["noise_logits"]), float(g["noise_grads_all"])
    e_l = l2(lob, g["logits_bf16"])
    names = [str(n) for n in g["param_names"]]
    e_g = l2(torch.cat([gb[n].flatten().double() for n in names]), torch.cat([ref_grads[n].flatten() for n in names]))
    print("res-OCR bf16-storage oracle vs reference run: logits %.4g (self-noise %.4g), loss %.6f vs %.6f, gradients %.4g "
          "(self-noise %.4g)" % (e_l, noise_l, lossb, float(g["loss_bf16"]), e_g, noise_g))
    assert e_l < 1.5 * noise_l and e_l < 2e-2
    assert abs(lossb - float(g["loss_bf16"])) < 2e-2 * float(g["loss_bf16"])
    assert e_g < 2.0 * noise_g
    import statistics
    per = [l2(gb[n].reshape(-1), ref_grads[n].reshape(-1)) for n in names if float(ref_grads[n].abs().max()) > 0]
    per_noise = [max(float(g["noise:" + n]), float(g["f32gap:" + n])) for n in names if float(ref_grads[n].abs().max()) > 0]
    assert statistics.median(per) < 2e-2 + 1.5 * statistics.median(per_noise)


def test_unet_odd_input_golden():
    """UNet_SNws on a 75 x 75 input: ceil-mode pooling + x2 up-sampling leave decoder maps larger than the skips and
    the reference's CropConcat zero-pads the skips (UNet_SNws.py:35-39); output 80 x 80."""
    g = load("g4_unet_odd")
    st = _state_from_pattern("unet", dtype=torch.float64)      # the fixture holds the reference's float64 run
    img, _ = patterns.synth_batch(2, 14, int(g["in_size"]), 131)
    dy = patterns.t32(patterns.fill(tuple(g["logits"].shape), 133, 1.0, 0.1))
    logits, _, grads = O.model_step("unet", st, img.double(), None, training=True, dlogits=dy.double())
    assert tuple(logits.shape) == (2, 10, 80, 80)
    close(logits, g["logits"], rtol=1e-6, atol=1e-6)
    for name in g["param_names"]:
        name = str(name)
        ref = g["grad:" + name]
        assert np.abs(grads[name].numpy() - ref).max() <= 1e-6 * np.abs(ref).max() + 1e-9, name
    for k in st:
        if O.is_buffer(k):
            close(st[k].double(), g["buf:" + k].astype(np.float64), rtol=1e-6)


def test_tile_normalisation_and_label_remap_known_answers():
    """cloud_dataset.py:127-137 restated (the dataset module itself needs gdal and cannot be imported): hand-checked
    values - NaN counts as 0 in the minimum / maximum and in the result, the extremes land on 0 and 255."""
    t = np.array([[[0.5, 1.0], [np.nan, 3.0]], [[-1.0, 2.0], [0.0, 1.0]]], np.float32)       # min -1, max 3
    want = (np.array([[[1.5, 2.0], [1.0, 4.0]], [[0.0, 3.0], [1.0, 2.0]]], np.float32) / np.float32(4.0)) * np.float32(255)
    assert np.array_equal(O.normalize_tile(t), want)
    assert O.normalize_tile(t).dtype == np.float32 and np.isnan(O.normalize_tile(np.full((1, 2, 2), 3.0, np.float32))).all()
    m = np.array([[0.0, 10.0, 255.0], [np.nan, 9.0, 3.0]], np.float32)
    assert np.array_equal(O.remap_label(m), np.array([[0, 0, 0], [0, 9, 3]], np.float32))
    assert np.array_equal(O.remap_label(np.array([10, 255, 4], np.uint8)), np.array([0, 0, 4], np.float32))
