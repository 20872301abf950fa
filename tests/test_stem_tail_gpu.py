"""Fused stem tail (bn1 -> relu -> maxpool 3x3/2) against the composition of the unfused kernels, through the C ABI: the pooled
activation, and in the backward dgamma / dbeta / dy, must be the SAME BITS — the fused kernels evaluate the same expressions in the
same order, they only skip materialising the 112x112x64 activation and its gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,C", [(3, 16, 16, 64), (2, 17, 23, 64), (4, 9, 9, 8), (1, 112, 112, 64)])
def test_bn_relu_maxpool_fused_equals_composition(cuda, B, H, W, C):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(B * 100 + H + W)
    M = B * H * W
    y = (torch.randn(B, H, W, C, generator=g) * 2 + 0.3).to(cuda)
    y[0, :4, :4] = -5.0                       # a window whose maximum is not positive: the ReLU gate must block its gradient
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    wsb = max(L.osi_bn_workspace(M, C), L.osi_bn_backward_workspace(M, C))
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    mean, invstd, scale, shift = (torch.empty(C, device=cuda) for _ in range(4))
    N.check(L.osi_bn_train_stats(N.ptr(y), M, C, N.ptr(gamma), N.ptr(beta), 1e-5, 0.1, None, None, N.ptr(mean), N.ptr(invstd), N.ptr(scale),
                                 N.ptr(shift), N.ptr(ws), wsb, T.S()))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    # unfused: apply (+mask) -> maxpool
    a = torch.empty_like(y)
    mask = torch.zeros(L.osi_bn_relu_mask_bytes(M, C), dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_apply_relu_mask(N.ptr(y), None, N.ptr(scale), N.ptr(shift), N.ptr(a), N.ptr(mask), M, C, T.S()))
    p_ref = torch.empty(B, Ho, Wo, C, device=cuda)
    idx_ref = torch.zeros(B * Ho * Wo * C, dtype=torch.uint8, device=cuda)
    N.check(L.osi_maxpool3x3s2_fwd(N.ptr(a), N.ptr(p_ref), N.ptr(idx_ref), B, H, W, C, T.S()))
    # fused
    p = torch.full((B, Ho, Wo, C), float("nan"), device=cuda)
    idx = torch.zeros(B * Ho * Wo * C, dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_relu_maxpool_fwd(N.ptr(y), N.ptr(scale), N.ptr(shift), N.ptr(p), N.ptr(idx), B, H, W, C, T.S()))
    assert torch.equal(p, p_ref)
    assert torch.equal(idx & 0x0F, idx_ref) and torch.equal((idx >> 7).bool(), (p_ref.reshape(-1) > 0))
    assert not bool((idx >> 7).bool().all()), "the test needs some gated windows"
    # backward: unfused = maxpool scatter -> BN backward with the ReLU bitmask
    gp = torch.randn(B, Ho, Wo, C, generator=g).to(cuda)
    dA = torch.empty_like(y)
    N.check(L.osi_maxpool3x3s2_bwd(N.ptr(gp), N.ptr(idx_ref), N.ptr(dA), B, H, W, C, T.S()))
    dy_ref, dg_ref, db_ref = torch.empty_like(y), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    N.check(L.osi_bn_backward_relu_mask(N.ptr(dA), N.ptr(mask), N.ptr(y), N.ptr(mean), N.ptr(invstd), N.ptr(gamma), N.ptr(dy_ref), None,
                                        N.ptr(dg_ref), N.ptr(db_ref), M, C, N.ptr(ws), wsb, T.S()))
    torch.cuda.synchronize()
    dy, dg, db = torch.full_like(y, float("nan")), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    N.check(L.osi_bn_relu_maxpool_bwd(N.ptr(gp), N.ptr(idx), N.ptr(y), N.ptr(mean), N.ptr(invstd), N.ptr(gamma), N.ptr(dy), N.ptr(dg),
                                      N.ptr(db), B, H, W, C, N.ptr(ws), wsb, T.S()))
    assert torch.equal(db, db_ref) and torch.equal(dg, dg_ref)
    assert torch.equal(dy, dy_ref)
