"""Balanced remainder of the forward / input-gradient convolutions (plan_tail_split in csrc/conv_igemm.hip): the tiles of a launch's
ragged last round are split along K into short workgroups, their raw accumulator tiles go through a slab, and a fix-up pass adds the
splits in fixed order and finishes the epilogue (output rows + BatchNorm partials). Oracle: torch conv2d in fp64 (the arithmetic the
reference runs under openset_imagenet/model.py:37); the un-split kernel is the second witness; results are bitwise reproducible."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _fwd(L, N, T, d, x, w, nb, act=None, res=None):
    ps = torch.full((nb // 4,), float("nan"), device=x.device)
    y = torch.full((d.B, d.Ho, d.Wo, d.Cout), float("nan"), device=x.device)
    P, rows = ctypes.c_int(), ctypes.c_int()
    if act is None:
        N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), T.S()))
    elif res is None:
        N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(act[0]), N.ptr(act[1]), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb,
                                   ctypes.byref(P), ctypes.byref(rows), T.S()))
    else:
        N.check(L.osi_conv_fwd_act2(ctypes.byref(d), N.ptr(x), N.ptr(act[0]), N.ptr(act[1]), N.ptr(res), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb,
                                    ctypes.byref(P), ctypes.byref(rows), T.S()))
    return y, ps, P.value, rows.value


# (Cin, Cout, k, stride, H, B, cus, fused): `cus` = CU count the plan balances for (osi_set_tuning("tail_cus")), chosen so that the
# case has full rounds + a split remainder (ragged last row tile included), only a remainder (uniform split-K), or 3x3 / strided taps
CASES = [(128, 64, 1, 1, 14, 3, 4, 0), (128, 128, 3, 1, 9, 3, 6, 1), (256, 128, 3, 2, 9, 5, 9, 1), (256, 64, 1, 1, 7, 5, 64, 2),
         (512, 512, 3, 1, 7, 8, 256, 1), (256, 256, 1, 1, 12, 2, 9, 0), (1024, 256, 1, 1, 14, 4, 12, 2)]


@pytest.mark.parametrize("Cin,Cout,k,stride,H,B,cus,fused", CASES)
def test_forward_tail_split_vs_fp64_and_unsplit(cuda, Cin, Cout, k, stride, H, B, cus, fused):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    pad = 1 if k == 3 else 0
    g = torch.Generator().manual_seed(Cin + Cout + H + cus)
    x = (torch.randn(B, H, H, Cin, generator=g) + 0.2).to(cuda)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(cuda)
    act = res = None
    a64 = x.double()
    if fused:
        act = ((torch.rand(Cin, generator=g) + 0.5).to(cuda), (torch.randn(Cin, generator=g) * 0.5).to(cuda))
        a64 = x.double() * act[0].double() + act[1].double()
        if fused == 2:
            res = torch.randn(B, H, H, Cin, generator=g).to(cuda)
            a64 = a64 + res.double()
        a64 = torch.relu(a64)
    ref = F.conv2d(T.nchw(a64), T.oihw(w.double()), None, stride, pad).permute(0, 2, 3, 1)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, stride, pad)
    M = B * d.Ho * d.Wo
    N.check(L.osi_set_tuning(b"tail_cus", cus))
    N.check(L.osi_set_tuning(b"tail_mint", 2)); N.check(L.osi_set_tuning(b"tail_smax", 32))    # small shapes: short splits allowed
    try:
        nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
        N.check(L.osi_set_tuning(b"tail_split", 0))
        nb0 = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
        y0, ps0, P0, rows0 = _fwd(L, N, T, d, x, w, nb0, act, res)
        N.check(L.osi_set_tuning(b"tail_split", 1))
        assert nb > nb0, "this case is meant to have a split remainder (the workspace grows by the slab)"
        y1, ps1, P1, rows1 = _fwd(L, N, T, d, x, w, nb, act, res)
        y2, ps2, _, _ = _fwd(L, N, T, d, x, w, nb, act, res)
        # a caller that only provides the old workspace size silently gets the un-split launch
        y3, _, _, _ = _fwd(L, N, T, d, x, w, nb0, act, res)
    finally:
        N.check(L.osi_set_tuning(b"tail_cus", 0))
        N.check(L.osi_set_tuning(b"tail_split", 1))
        N.check(L.osi_set_tuning(b"tail_mint", 16)); N.check(L.osi_set_tuning(b"tail_smax", 8))
    K = Cin * k * k
    tol = (2e-6 + 6e-8 * K ** 0.5) * float(ref.abs().max()) + 1e-6
    assert float((y1.double() - ref).abs().max()) <= tol and float((y0.double() - ref).abs().max()) <= tol
    assert not torch.isnan(y1).any() and (P1, rows1) == (P0, rows0) == ((M + 63) // 64, 64)
    assert torch.equal(y1, y2) and torch.equal(ps1[:2 * P1 * Cout], ps2[:2 * P1 * Cout]), "fixed-order fix-up: bitwise reproducible"
    assert torch.equal(y3, y0)
    assert not torch.equal(y1, y0), "the split really changed the summation order of some tile"
    # row tiles of the full rounds are untouched by the split: the same bits as the un-split launch
    if ((M + 63) // 64) * (Cout // 64) >= 2 * cus:
        assert torch.equal(y1.view(M, Cout)[:64], y0.view(M, Cout)[:64])
    # BatchNorm partials: per row tile (mean, M2) of THIS output, every tile (split or not)
    yv = y1.double().view(M, Cout)
    pm, pq = ps1[:P1 * Cout].view(P1, Cout).double(), ps1[P1 * Cout:2 * P1 * Cout].view(P1, Cout).double()
    for t in range(P1):
        rows = yv[t * 64:min(M, (t + 1) * 64)]
        assert torch.allclose(pm[t], rows.mean(0), atol=2e-6 * float(ref.abs().max())), t
        assert torch.allclose(pq[t], ((rows - rows.mean(0)) ** 2).sum(0), rtol=1e-4, atol=1e-5), t


def test_production_plans_at_batch_128(cuda):
    """The ResNet-50 layers this exists for, at the benchmark's batch: 7x7 and 14x14 layers on a 256-CU plan. Output against the
    un-split launch (summation-order noise only), workspace growth bounded, bitwise reproducible."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    N.check(L.osi_set_tuning(b"tail_cus", 256))
    try:
        for Cin, Cout, k, stride, H in [(2048, 512, 1, 1, 7), (512, 512, 3, 1, 7), (1024, 256, 1, 1, 14), (256, 256, 3, 2, 28), (256, 256, 3, 1, 14)]:
            pad = 1 if k == 3 else 0
            g = torch.Generator(device=cuda).manual_seed(Cin + H)
            x = torch.randn(128, H, H, Cin, device=cuda, generator=g)
            w = torch.randn(Cout, k, k, Cin, device=cuda, generator=g) / (Cin * k * k) ** 0.5
            d = N.ConvDesc.make(128, H, H, Cin, Cout, k, stride, pad)
            nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
            N.check(L.osi_set_tuning(b"tail_split", 0))
            nb0 = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
            y0, _, _, _ = _fwd(L, N, T, d, x, w, nb0)
            N.check(L.osi_set_tuning(b"tail_split", 1))
            assert nb0 < nb <= nb0 + 2 * 256 * 64 * 64 * 4, (Cin, Cout, k, H)       # at most ~two tiles per CU go through the slab
            y1, _, _, _ = _fwd(L, N, T, d, x, w, nb)
            y2, _, _, _ = _fwd(L, N, T, d, x, w, nb)
            assert torch.equal(y1, y2)
            assert float((y1 - y0).abs().max()) <= 2e-5 * float(y0.abs().max())
            frac = float((y1 != y0).any(dim=-1).float().mean())
            assert 0 < frac < 0.2, f"only the tiles of the ragged last round take the split path ({frac:.3f} of the rows changed)"
    finally:
        N.check(L.osi_set_tuning(b"tail_cus", 0))
        N.check(L.osi_set_tuning(b"tail_split", 1))


class _Fusion(ctypes.Structure):
    _fields_ = [("relu_mask", ctypes.c_void_p), ("y0", ctypes.c_void_p), ("mean0", ctypes.c_void_p), ("invstd0", ctypes.c_void_p),
                ("y1", ctypes.c_void_p), ("mean1", ctypes.c_void_p), ("invstd1", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("partials_bytes", ctypes.c_size_t), ("scale0", ctypes.c_void_p), ("shift0", ctypes.c_void_p),
                ("pool_idx", ctypes.c_void_p), ("pool_H", ctypes.c_int), ("pool_W", ctypes.c_int), ("addend_stride", ctypes.c_int)]


# (Cin, Cout, k, H, B, cus, two BatchNorm consumers, gate from the bitmask, addend)
DCASES = [(64, 128, 1, 14, 3, 4, False, True, False), (128, 128, 3, 9, 3, 6, True, True, True), (64, 256, 1, 7, 5, 64, False, False, True),
          (256, 512, 1, 7, 8, 256, True, False, False), (128, 64, 3, 12, 2, 9, False, False, False)]


@pytest.mark.parametrize("Cin,Cout,k,H,B,cus,two,bits,add", DCASES)
def test_input_gradient_tail_split_vs_fp64_and_unsplit(cuda, Cin, Cout, k, H, B, cus, two, bits, add):
    """Stride-1 input gradient with the fused epilogue (addend, ReLU gate from the bitmask or recomputed from the pre-BN tensor,
    BatchNorm-backward partial sums for one or two consumers): split remainder vs the un-split launch vs fp64."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    pad = 1 if k == 3 else 0
    g = torch.Generator().manual_seed(Cin + 3 * Cout + H + cus)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, 1, pad)
    M = B * H * H
    dy = torch.randn(B, d.Ho, d.Wo, Cout, generator=g).to(cuda)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cout * k * k) ** 0.5).to(cuda)
    addend = torch.randn(B, H, H, Cin, generator=g).to(cuda) if add else None
    ys = [(torch.randn(M, Cin, generator=g) * 2 + 0.5).to(cuda) for _ in range(2 if two else 1)]
    means = [y.mean(0) for y in ys]
    invs = [1 / torch.sqrt(y.var(0, unbiased=False) + 1e-5) for y in ys]
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(cuda), (torch.randn(Cin, generator=g) * 0.5).to(cuda)
    gate = torch.addcmul(sh, ys[0], sc) > 0                                       # the fma the kernels evaluate
    mask = None
    if bits:
        out = torch.empty(M, Cin, device=cuda)
        mask = torch.zeros(L.osi_bn_relu_mask_bytes(M, Cin), dtype=torch.uint8, device=cuda)
        N.check(L.osi_bn_apply_relu_mask(N.ptr(ys[0]), None, N.ptr(sc), N.ptr(sh), N.ptr(out), N.ptr(mask), M, Cin, T.S()))
        gate = out > 0

    def run(pb):
        parts = torch.full((pb // 4,), float("nan"), device=cuda)
        f = _Fusion(mask.data_ptr() if bits else None, ys[0].data_ptr(), means[0].data_ptr(), invs[0].data_ptr(),
                    ys[1].data_ptr() if two else None, means[1].data_ptr() if two else None, invs[1].data_ptr() if two else None,
                    parts.data_ptr(), pb, None if bits else sc.data_ptr(), None if bits else sh.data_ptr())
        gbuf = torch.full((B, H, H, Cin), float("nan"), device=cuda)
        P = ctypes.c_int()
        N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(gbuf), N.ptr(addend), ctypes.byref(f), 0, ctypes.byref(P), T.S()))
        return gbuf, parts[:3 * P.value * Cin].clone().view(3, P.value, Cin), P.value

    N.check(L.osi_set_tuning(b"tail_cus", cus))
    N.check(L.osi_set_tuning(b"tail_mint", 2)); N.check(L.osi_set_tuning(b"tail_smax", 32))
    try:
        pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d))
        N.check(L.osi_set_tuning(b"tail_split", 0))
        pb0 = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d))
        g0, p0, P0 = run(pb0)
        N.check(L.osi_set_tuning(b"tail_split", 1))
        assert pb > pb0, "this case is meant to have a split remainder"
        g1, p1, P1 = run(pb)
        g2, p2, _ = run(pb)
    finally:
        N.check(L.osi_set_tuning(b"tail_cus", 0))
        N.check(L.osi_set_tuning(b"tail_split", 1))
        N.check(L.osi_set_tuning(b"tail_mint", 16)); N.check(L.osi_set_tuning(b"tail_smax", 8))
    ref = torch.nn.grad.conv2d_input((B, Cin, H, H), T.oihw(w.double()), T.nchw(dy.double()), 1, pad).permute(0, 2, 3, 1)
    if add:
        ref = ref + addend.double()
    ref = ref * gate.view(B, H, H, Cin)
    tol = (2e-6 + 6e-8 * (Cout * k * k) ** 0.5) * float(ref.abs().max()) + 1e-6
    assert float((g1.double() - ref).abs().max()) <= tol and float((g0.double() - ref).abs().max()) <= tol
    assert P1 == P0 == (M + 63) // 64 and not torch.isnan(g1).any()
    assert torch.equal(g1, g2) and torch.equal(p1[:2 + two], p2[:2 + two]), "bitwise reproducible"
    assert not torch.equal(g1, g0)
    # exact zeros where the gate is closed, on both routes
    assert bool((g1.view(M, Cin)[~gate] == 0).all())
    # the partial sums describe THIS masked gradient: sum g, sum g * xhat per row tile
    gv = g1.double().view(M, Cin)
    for t in range(P1):
        rows = slice(t * 64, min(M, (t + 1) * 64))
        scale_ = float(gv[rows].abs().sum(0).max()) + 1e-6
        assert torch.allclose(p1[0, t].double(), gv[rows].sum(0), atol=2e-6 * scale_ + 1e-5), t
        for i in range(2 if two else 1):
            xhat = (ys[i].double()[rows] - means[i].double()) * invs[i].double()
            assert torch.allclose(p1[1 + i, t].double(), (gv[rows] * xhat).sum(0), atol=1e-5 * scale_ * 4 + 1e-5), (t, i)
