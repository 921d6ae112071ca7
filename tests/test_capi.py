"""CPU-side checks of the C-ABI boundary and the host logic (no kernel is launched)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import snws_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from snws import _lib
    header = open(os.path.join(ROOT, "include", "snws.h")).read()
    declared = set(re.findall(r"\b(snws_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(handle, name) is not None
    assert _lib.lib().snws_version() >= 100


def test_host_side_size_queries_and_errors():
    from snws import _lib as L
    lib = L.lib()
    assert lib.snws_conv_out_size(512, 3, 1, 1, 1) == 512
    assert lib.snws_conv_out_size(12, 3, 2, 1, 1) == 6
    assert lib.snws_conv_out_size(12, 3, 1, 2, 2) == 12
    assert lib.snws_conv_pack_bytes(14, 64, 3, 3) == 16 * 64 * 9 * 2        # 14 -> 16 channels
    assert lib.snws_conv_pack_bytes(320, 10, 1, 1) == 320 * 64 * 2          # 10 -> 64 output channels
    assert lib.snws_conv_mfma_tiles(512, 512) == 32 * 16
    assert lib.snws_conv_mfma_tiles(17, 33) == 2 * 2
    assert lib.snws_sn_ctx_floats(16, 64) == 5 * 16 * 64 + 2 * 16 + 2 * 64
    assert lib.snws_sn_ctxb_floats(2, 8) == 6 * 16 + 4
    assert 1 <= lib.snws_sn_stats_tiles(16, 512 * 512, 64) <= 256
    # argument errors are reported before anything touches the GPU
    with pytest.raises(ValueError, match="num_class"):
        L.call("snws_confusion_hist", None, 0, None, 3, 10, 0, None, None)
    with pytest.raises(ValueError, match="fan-in"):
        L.call("snws_ws_fwd", ctypes.c_void_p(16), 4, 1, 1, 1, None, None, None, None, None)
    d = L.ConvDesc(1, 8, 8, 16, 16, 3, 3, 2, 1, 1, 1, 16, 16)
    with pytest.raises(L.SnwsUnsupported):
        L.call("snws_conv_mfma_fwd", ctypes.byref(d), ctypes.c_void_p(16), ctypes.c_void_p(16), None,
               ctypes.c_void_p(16), None, None)
    d = L.ConvDesc(1, 8, 8, 16, 16, 3, 3, 1, 1, 1, 3, 16, 16)
    with pytest.raises(ValueError, match="groups"):
        L.call("snws_conv_direct_fwd", ctypes.byref(d), 0, ctypes.c_void_p(16), ctypes.c_void_p(16), None,
               ctypes.c_void_p(16), None)


def test_missing_library_fails_loudly(monkeypatch):
    from snws import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsnws.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_modules_refuse_cpu_tensors():
    import modeling.snorm as sn
    import modeling.weightstd as ws
    from modeling.UNet3p_SNws import UNet3p_SNws
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ws.Conv2dws(4, 4, 3, 1, 1)(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sn.SwitchNorm2d(4)(torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError):
        sn.SwitchNorm2d(4)(torch.zeros(1, 4, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        UNet3p_SNws(14, 4, 10, 1, 1)(torch.zeros(1, 14, 32, 32))


@pytest.mark.parametrize("arch,tag", [("unet", "g4_unet"), ("unet3p", "g4_unet3p")])
def test_state_dict_matches_reference(arch, tag, golden_dir):
    from modeling.UNet3p_SNws import UNet3p_SNws
    from modeling.UNet_SNws import UNet_SNws
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    m = (UNet3p_SNws if arch == "unet3p" else UNet_SNws)(14, 4, 10, 1, 1)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["state_shapes"]]
    # module defaults of the reference: gamma 1, beta 0, softmax weights 1, buffers 0
    assert float(sd["conv_fusion.1.weight"].min() if arch == "unet3p" else sd["conv7_3.1.weight"].min()) == 1.0
    full = (UNet3p_SNws if arch == "unet3p" else UNet_SNws)(14, 64, 10, 1, 1)
    assert sum(p.numel() for p in full.parameters()) == (21111600 if arch == "unet3p" else 28963534)


def test_switchnorm_module_attributes():
    import modeling.snorm as sn
    m = sn.SwitchNorm2d(6, using_bn=False, last_gamma=True)
    assert tuple(m.weight.shape) == (1, 6, 1, 1) and float(m.weight.abs().sum()) == 0.0
    assert m.mean_weight.numel() == 2 and not hasattr(m, "running_mean")
    m = sn.SwitchNorm2d(6)
    assert tuple(m.running_var.shape) == (1, 6, 1) and m.momentum == 0.997 and m.eps == 1e-5


def test_evaluator_metrics_host_side(golden_dir):
    from utils.metrics import Evaluator
    g = np.load(os.path.join(golden_dir, "g6_evaluator.npz"))
    for case, C in (("a", 10), ("b", 10), ("d", 10), ("e", 3)):
        ev = Evaluator(C)
        ev.confusion_matrix = g[case + "_cm"].copy()
        with np.errstate(all="ignore"):
            got = (ev.Pixel_Accuracy(), ev.Pixel_Accuracy_Class(), ev.Mean_Intersection_over_Union(),
                   ev.Frequency_Weighted_Intersection_over_Union())
        for v, key in zip(got, ("_pa", "_pac", "_miou", "_fwiou")):
            assert np.array_equal(np.asarray(v, np.float64).view(np.uint64), g[case + key].view(np.uint64))
        assert all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
                   for a, b in zip(got, O.evaluator_metrics(g[case + "_cm"])))


def test_poly_lr_and_tiles(golden_dir):
    from snws import harness
    g = np.load(os.path.join(golden_dir, "g78_misc.npz"))
    for (i, ep), lr in zip(g["lr_pts"], g["lr_vals"]):
        assert harness.poly_lr(0.002, int(ep) * 746 + int(i), 200 * 746) == lr


def test_tile_offsets_match_golden():
    """snws.tiles.gen_tiles_offs (host side of the full-disk inference path) against the vectors produced by the
    reference's utils/gen_tiles_offs.py:3-27."""
    import numpy as np
    from snws import tiles
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g78_misc.npz"))
    for i in range(int(g["n_tiles_cases"])):
        args = [int(v) for v in g["tiles%d_args" % i]]
        assert np.array_equal(np.array(tiles.gen_tiles_offs(*args), np.int64), g["tiles%d" % i])
    assert len(tiles.gen_tiles_offs(2748, 2748, 512, 0)) == 36


def test_saver_layout_and_checkpoint_format(tmp_path):
    """snws.checkpoint.Saver reproduces the run-directory layout of utils/saver.py:11-47 and the checkpoint dict of
    train.py:272-277; a DataParallel-style 'module.' prefix is accepted on load."""
    from snws import checkpoint
    net = torch.nn.Sequential(torch.nn.Linear(3, 2))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    root = str(tmp_path / "run")
    s0 = checkpoint.Saver("cloud", "unet3p", root=root)
    f0 = s0.save_checkpoint(net, opt, epoch=1, best_pred=0.5, is_best=True)
    assert f0.endswith(os.path.join("cloud", "unet3p", "experiment_0", "checkpoint.pth.tar"))
    assert open(os.path.join(os.path.dirname(f0), "best_pred.txt")).read() == "0.5"
    assert os.path.exists(os.path.join(root, "cloud", "unet3p", "model_best.pth.tar"))
    s1 = checkpoint.Saver("cloud", "unet3p", root=root)
    assert s1.experiment_dir.endswith("experiment_1")
    s1.save_checkpoint(net, opt, epoch=2, best_pred=0.4, is_best=True)       # does not beat run 0: best file unchanged
    best = torch.load(os.path.join(root, "cloud", "unet3p", "model_best.pth.tar"), weights_only=True)
    assert best["epoch"] == 1 and best["best_pred"] == 0.5
    ck = torch.load(f0, weights_only=True)
    ck["state_dict"] = {"module." + k: v for k, v in ck["state_dict"].items()}
    p2 = str(tmp_path / "dp.pth.tar")
    torch.save(ck, p2)
    net2 = torch.nn.Sequential(torch.nn.Linear(3, 2))
    epoch, bp = checkpoint.load_checkpoint(p2, net2, None)
    assert (epoch, bp) == (1, 0.5) and torch.equal(net2[0].weight, net[0].weight)
    with pytest.raises(RuntimeError):
        checkpoint.load_checkpoint(str(tmp_path / "missing.pth.tar"), net2)


def test_share_grad_is_idempotent():
    """Marking an already shared tensor again must not stack a second identity node whose slot no consumer fills."""
    import torch
    from snws import ops
    x = torch.ones(1, 2, 3, 3, requires_grad=True) * 2.0
    y = ops.share_grad(x)
    if ops.GRAD_SHARE:
        assert ops.grad_slot(y) is not None
    assert ops.share_grad(y) is y
