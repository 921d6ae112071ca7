"""The entry point that allocates nothing (snws_sn_apply_noalloc) against the stand-alone SwitchNorm entry points it
is a variant of (snws_sn_stats -> finalize -> mix_a -> mix_b -> snws_sn_apply with caller-provided side buffers),
called through the C ABI.  Same kernels, same launch geometry: outputs and running buffers must be bit-identical."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _reference_chain(L, x, res, gamma, beta, mw, vw, rm, rv, relu, training, dt):
    N, H, W, C = x.shape
    HW = H * W
    lib = L.lib()
    T = lib.snws_sn_stats_tiles(N, HW, C)
    part = torch.empty((N, T, C, 2), dtype=torch.float32, device=DEV)
    shift = torch.empty((N, C), dtype=torch.float32, device=DEV)
    ctx = torch.empty(lib.snws_sn_ctx_floats(N, C), dtype=torch.float32, device=DEV)
    pack = torch.empty(2 * C + 1, dtype=torch.float32, device=DEV)
    y = torch.empty_like(x)
    L.call("snws_sn_stats", _ptr(x), dt, N, HW, C, C, _ptr(part), _ptr(shift), None)
    L.call("snws_sn_finalize", _ptr(part), T, _ptr(shift), 1, N, HW, C, _ptr(ctx), None)
    L.call("snws_sn_mix_a", _ptr(ctx), _ptr(pack), N, C, None)
    L.call("snws_sn_mix_b", _ptr(ctx), _ptr(pack), _ptr(gamma), _ptr(mw), _ptr(vw), _ptr(rm), _ptr(rv), N, C, 1e-5, 0.997,
           int(training), 1, 1, None)
    L.call("snws_sn_apply", _ptr(x), C, _ptr(y), C, _ptr(res), C if res is not None else 0, dt, _ptr(ctx), _ptr(beta), N,
           HW, C, int(relu), None)
    return y


def _case(L, seed, N, H, W, C, dtype, relu, with_res, training):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn((N, H, W, C), generator=g) * 3.0 + 1.5).to(device=DEV, dtype=dtype)
    res = torch.randn((N, H, W, C), generator=g).to(device=DEV, dtype=dtype) if with_res else None
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.rand(C, generator=g) - 0.5).to(DEV)
    mw = torch.tensor([0.3, -0.2, 0.9], device=DEV)
    vw = torch.tensor([-0.5, 0.4, 0.1], device=DEV)
    rm0 = torch.rand(C, generator=g).to(DEV)
    rv0 = (torch.rand(C, generator=g) + 0.5).to(DEV)
    dt = L.SNWS_BF16 if dtype == torch.bfloat16 else L.SNWS_F32
    rm_a, rv_a = rm0.clone(), rv0.clone()
    want = _reference_chain(L, x, res, gamma, beta, mw, vw, rm_a, rv_a, relu, training, dt)
    rm_b, rv_b = rm0.clone(), rv0.clone()
    got = torch.empty_like(x)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    free_before = torch.cuda.mem_get_info()[0]      # the driver's view: sees a hipMalloc inside the library too
    L.call("snws_sn_apply_noalloc", _ptr(x), C, _ptr(got), C, _ptr(res), C if with_res else 0, dt, _ptr(gamma), _ptr(beta),
           _ptr(mw), _ptr(vw), _ptr(rm_b), _ptr(rv_b), N, H * W, C, 1e-5, 0.997, int(training), 1, 1, int(relu), None)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] >= free_before
    assert torch.equal(got, want), (N, H, W, C, dtype)
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    if training:
        assert not torch.equal(rm_b, rm0)        # the running buffers did move
    else:
        assert torch.equal(rm_b, rm0)


def test_noalloc_switchnorm_matches_standalone_entry_points():
    from snws import _lib as L
    lib = L.lib()
    assert lib.snws_sn_noalloc_max_floats() == 4 << 20       # also resolves the array's address on this device
    # two calls in a row on one GPU, the second with a smaller input: the shared array still holds the first call's
    # partial sums and coefficients, none of which may leak into the second result
    _case(L, 1, 4, 96, 80, 64, torch.bfloat16, True, False, True)
    _case(L, 2, 2, 24, 20, 24, torch.bfloat16, True, False, True)
    # f32, residual add, eval mode (running buffers are read, not written), channel count off the vector width
    _case(L, 3, 3, 17, 19, 10, torch.float32, False, True, False)
    _case(L, 4, 1, 8, 8, 320, torch.float32, True, True, True)
    # the largest SwitchNorm call of the benchmark configuration fits the array
    assert lib.snws_sn_noalloc_floats(16, 512 * 512, 320) <= lib.snws_sn_noalloc_max_floats()


def test_noalloc_switchnorm_rejects_oversized_input():
    from snws import _lib as L
    lib = L.lib()
    N, HW, C = 64, 512 * 512, 1024
    assert lib.snws_sn_noalloc_floats(N, HW, C) > lib.snws_sn_noalloc_max_floats()
    x = torch.zeros(16, device=DEV)
    with pytest.raises(ValueError, match="static scratch array"):
        L.call("snws_sn_apply_noalloc", _ptr(x), C, _ptr(x), C, None, 0, L.SNWS_F32, _ptr(x), _ptr(x), _ptr(x), _ptr(x),
               _ptr(x), _ptr(x), N, HW, C, 1e-5, 0.997, 1, 1, 1, 1, None)
