"""The kernel variants kept behind SNWS_* switches for A/B measurements must compute what the default variants
compute.  The switches are read once per process, so every variant runs in a process of its own (tests/ab_probe.py)
and the results are compared here: element-wise kernels and the matrix kernels bit for bit (the variants keep the order
of every floating-point sum), anything downstream of a re-grouped partial-sum reduction to rounding noise."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, env_extra):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "ab_probe.py"), out], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    return _run(tmp_path_factory.mktemp("ab"), "default", {})


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(float(np.linalg.norm(b.astype(np.float64))), 1e-30))


def test_wave_priority_changes_no_bit(default_run, tmp_path):
    # SNWS_PRIO=0: no wave priorities in the 3x3 matrix kernels - scheduling only
    other = _run(tmp_path, "prio", {"SNWS_PRIO": "0"})
    for k in default_run.files:
        assert np.array_equal(other[k], default_run[k]), (k, _rel(other[k], default_run[k]))


def test_wgrad_launch_forms_agree(default_run, tmp_path):
    # SNWS_WGRAD9_PP=0: every 3x3 weight gradient on conv_wgrad9q_kernel (two independent workgroups per CU, ~32 pixel
    # tiles per work item) instead of the ping-pong kernel (one 8-wave workgroup per CU, one resident round of long
    # work items).  Same wave body and order inside a work item; the pixel tiles are cut into different partial sums,
    # so filter gradients may move by float32 rounding - and nothing else may move at all.
    other = _run(tmp_path, "pp", {"SNWS_WGRAD9_PP": "0"})
    # (at the probe's sizes both forms cut the pixel tiles alike and every bit agrees; the cuts differ from about 64 K
    # pixel-tile x channel-pair products upwards, where tests/test_gpu_fullsize.py checks both against exact checksums)
    for k in default_run.files:
        a, b = other[k], default_run[k]
        if np.array_equal(a, b):
            continue
        assert a.ndim == 4 and "_d_" in k and _rel(a, b) < 1e-5, (k, _rel(a, b))


def test_streaming_walk_variants_agree(default_run, tmp_path):
    # contiguous ranges instead of interleaved pieces: the apply passes are element-wise (the forward output of the
    # fused block - statistics from the conv epilogue - must be identical); stand-alone statistics and backward
    # reductions group their partial sums differently, so what depends on them may move by rounding noise
    other = _run(tmp_path, "walk", {"SNWS_SN_INTERLEAVE": "0"})
    for k in default_run.files:
        a, b = other[k], default_run[k]
        if k.startswith("m16"):
            # the whole bf16 model amplifies the re-grouped sums' rounding noise layer by layer: forward only, loosely
            if k.endswith("_logits"):
                assert _rel(a, b) < 3e-2, (k, _rel(a, b))
            continue
        if k.startswith("b") and k.endswith("_y"):
            assert np.array_equal(a, b), k
        else:
            assert _rel(a, b) < 3e-3, (k, _rel(a, b))


def test_batched_switchnorm_launches_agree(default_run, tmp_path):
    # SNWS_SN_BATCH=0: every SwitchNorm call of a decoder scale launches its own passes, in program order, instead of
    # sharing one batched launch per pass and round.  The arithmetic of a call is the same code either way: logits,
    # running statistics and every gradient are identical - except the SwitchNorm parameters and the conv bias of a
    # module used on several sources of one scale, whose per-use gradients are summed in forward order by the batched
    # rounds and in autograd's order otherwise (rounding noise).
    other = _run(tmp_path, "snbatch", {"SNWS_SN_BATCH": "0"})
    loose = 0
    for k in default_run.files:
        a, b = other[k], default_run[k]
        if np.array_equal(a, b):
            continue
        is_sn_or_bias = k.startswith("m16_d_") and (k.endswith(".1.weight") or k.endswith(".1.bias") or k.endswith("mean_weight")
                                                     or k.endswith("var_weight") or k.endswith(".0.bias"))
        assert is_sn_or_bias and _rel(a, b) < 1e-5, (k, _rel(a, b))
        loose += 1
    assert loose < 0.25 * len(default_run.files)
