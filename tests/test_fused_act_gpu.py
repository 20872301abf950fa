"""Convolutions that apply the producer layer's BatchNorm + ReLU in their operand loader (osi_conv_fwd_act, osi_conv_wgrad_act) and
the input-gradient epilogue that recomputes the ReLU gate from the pre-BN tensor instead of a stored bitmask: the activation
relu(bn(y)) between conv1 -> conv2 -> conv3 of a Bottleneck (torchvision Bottleneck.forward under reference model.py:37) is never
written to HBM. Oracle: torch conv2d / conv2d_weight on the materialised activation in fp64; the gate path must equal the
bitmask path bit for bit."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64, 1, 1, 14, 3), (64, 128, 3, 1, 9, 3), (128, 64, 3, 2, 9, 2), (256, 128, 1, 1, 7, 5), (64, 64, 3, 1, 20, 2),
          (128, 256, 1, 2, 8, 3)]


def _case(cuda, Cin, Cout, k, stride, H, B):
    g = torch.Generator().manual_seed(Cin * 7 + Cout + k + H)
    x = (torch.randn(B, H, H, Cin, generator=g) * 1.5).to(cuda)
    sc = (torch.rand(Cin, generator=g) + 0.5).to(cuda)
    sh = (torch.randn(Cin, generator=g) * 0.7).to(cuda)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(cuda)
    act64 = torch.relu(x.double() * sc.double() + sh.double())          # [B,H,W,Cin]
    return x, sc, sh, w, act64


@pytest.mark.parametrize("Cin,Cout,k,stride,H,B", SHAPES)
@pytest.mark.parametrize("tile", [0, 5, 6])
def test_conv_fwd_with_fused_input_activation(cuda, Cin, Cout, k, stride, H, B, tile):
    import osi_testlib as T
    from openset_imagenet import _native as N
    if tile == 6 and Cout % 128:
        pytest.skip("64x128 tile needs Cout % 128 == 0")
    L = N.lib()
    pad = 1 if k == 3 else 0
    x, sc, sh, w, act64 = _case(cuda, Cin, Cout, k, stride, H, B)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, stride, pad)
    y = torch.full((B, d.Ho, d.Wo, Cout), float("nan"), device=cuda)
    pb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
    ps = torch.full((pb // 4,), float("nan"), device=cuda)
    P, rows = ctypes.c_int(), ctypes.c_int()
    N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), tile, N.ptr(ps), pb,
                               ctypes.byref(P), ctypes.byref(rows), T.S()), "osi_conv_fwd_act")
    ref = F.conv2d(T.nchw(act64), T.oihw(w.double()), None, stride, pad).permute(0, 2, 3, 1)
    err = float((y.double() - ref).abs().max())
    K = Cin * k * k
    assert err <= (2e-6 + 6e-8 * K ** 0.5) * float(ref.abs().max()) + 1e-6, err
    # the BatchNorm partials of the epilogue describe this output: merged they give its per-channel mean
    M = B * d.Ho * d.Wo
    pm = ps[:P.value * Cout].view(P.value, Cout).double()
    cnt = torch.full((P.value, 1), float(rows.value), dtype=torch.float64, device=cuda)
    cnt[-1] = M - (P.value - 1) * rows.value
    assert torch.allclose((pm * cnt).sum(0) / M, y.double().view(M, Cout).mean(0), atol=1e-5)
    # without statistics, and the materialised-activation route through the plain kernel agrees to summation-order noise
    y2 = torch.empty_like(y)
    N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y2), tile, None, 0, None, None, T.S()))
    if tile == 0:   # AUTO + statistics may split the ragged last round along K; without a workspace it cannot
        assert float((y2 - y).abs().max()) <= 1e-5 * float(ref.abs().max())
    else:
        assert torch.equal(y2, y)
    act32 = torch.relu(torch.addcmul(sh, x, sc)).contiguous()
    y3 = T.conv_fwd(act32, w, k, stride, pad, tile)
    assert float((y3 - y).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("Cin,Cout,k,stride,H,B", SHAPES + [(128, 128, 3, 1, 28, 8), (256, 256, 1, 1, 14, 16)])
def test_conv_wgrad_with_fused_input_activation(cuda, Cin, Cout, k, stride, H, B):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    pad = 1 if k == 3 else 0
    x, sc, sh, w, act64 = _case(cuda, Cin, Cout, k, stride, H, B)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, stride, pad)
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(B, d.Ho, d.Wo, Cout, generator=g).to(cuda)
    nb = L.osi_conv_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=cuda)
    dw = torch.full((Cout, k, k, Cin), float("nan"), device=cuda)
    N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(ws), nb, T.S()), "osi_conv_wgrad_act")
    ref = torch.nn.grad.conv2d_weight(T.nchw(act64), (Cout, Cin, k, k), T.nchw(dy.double()), stride, pad).permute(0, 2, 3, 1)
    Kp = B * d.Ho * d.Wo
    err = float((dw.double() - ref).abs().max())
    assert err <= (2e-6 + 6e-8 * Kp ** 0.5) * float(ref.abs().max()) + 1e-6, err
    # bit-reproducible, and equal (to summation noise) to the plain kernel on the materialised activation
    dw2 = torch.empty_like(dw)
    N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw2), N.ptr(ws), nb, T.S()))
    assert torch.equal(dw, dw2)
    act32 = torch.relu(torch.addcmul(sh, x, sc)).contiguous()
    assert float((T.conv_wgrad(dy, act32, k, stride, pad) - dw).abs().max()) <= 2e-5 * float(ref.abs().max())


class _Fusion(ctypes.Structure):
    _fields_ = [("relu_mask", ctypes.c_void_p), ("y0", ctypes.c_void_p), ("mean0", ctypes.c_void_p), ("invstd0", ctypes.c_void_p),
                ("y1", ctypes.c_void_p), ("mean1", ctypes.c_void_p), ("invstd1", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("partials_bytes", ctypes.c_size_t), ("scale0", ctypes.c_void_p), ("shift0", ctypes.c_void_p),
                ("pool_idx", ctypes.c_void_p), ("pool_H", ctypes.c_int), ("pool_W", ctypes.c_int), ("addend_stride", ctypes.c_int)]


@pytest.mark.parametrize("Cin,Cout,k,stride,H,B", [(64, 64, 1, 1, 14, 3), (128, 64, 3, 1, 9, 3), (256, 128, 3, 2, 9, 2), (512, 64, 1, 1, 7, 5)])
def test_dgrad_gate_from_pre_bn_tensor_equals_bitmask(cuda, Cin, Cout, k, stride, H, B):
    """The fused dgrad epilogue with the ReLU gate recomputed as y0 * scale0 + shift0 > 0 gives exactly the bits of the stored
    bitmask route (osi_bn_apply_relu_mask evaluates the same fma): masked gradient and BatchNorm partials identical."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    pad = 1 if k == 3 else 0
    g = torch.Generator().manual_seed(Cin + Cout + H)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, stride, pad)
    M = B * H * H
    dy = torch.randn(B, d.Ho, d.Wo, Cout, generator=g).to(cuda)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cout * k * k) ** 0.5).to(cuda)
    y0 = (torch.randn(M, Cin, generator=g) * 2 + 0.3).to(cuda)
    y0.view(-1)[::97] = 0.0                                               # exact zeros: gate must be off exactly where relu' is 0
    ga, be = (torch.rand(Cin, generator=g) + 0.5).to(cuda), torch.randn(Cin, generator=g).to(cuda)
    wsb = max(L.osi_bn_workspace(M, Cin), L.osi_bn_backward_workspace(M, Cin))
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    st = [torch.empty(Cin, device=cuda) for _ in range(4)]                # mean, invstd, scale, shift
    N.check(L.osi_bn_train_stats(N.ptr(y0), M, Cin, N.ptr(ga), N.ptr(be), 1e-5, 0.1, None, None, *[N.ptr(t) for t in st], N.ptr(ws), wsb, T.S()))
    out = torch.empty(M, Cin, device=cuda)
    mask = torch.zeros(L.osi_bn_relu_mask_bytes(M, Cin), dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_apply_relu_mask(N.ptr(y0), None, N.ptr(st[2]), N.ptr(st[3]), N.ptr(out), N.ptr(mask), M, Cin, T.S()))
    pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d))

    def both_routes():
        res = []
        for use_bits in (True, False):
            parts = torch.full((pb // 4,), float("nan"), device=cuda)
            f = _Fusion(mask.data_ptr() if use_bits else None, y0.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), None, None, None,
                        parts.data_ptr(), pb, None if use_bits else st[2].data_ptr(), None if use_bits else st[3].data_ptr())
            gbuf = torch.full((B, H, H, Cin), float("nan"), device=cuda)
            P = ctypes.c_int()
            N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(gbuf), None, ctypes.byref(f), 0, ctypes.byref(P), T.S()))
            res.append((gbuf, parts[:2 * P.value * Cin].clone()))
        return res

    gate = (out > 0).view(B, H, H, Cin)
    N.check(L.osi_set_tuning(b"dgrad_w3", 0))      # both routes on the same kernel form (same K order): bit for bit
    try:
        res = both_routes()
    finally:
        N.check(L.osi_set_tuning(b"dgrad_w3", 1))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert bool((res[1][0][~gate] == 0).all()) and float(res[1][0][gate].abs().sum()) > 0
    # default knobs: the recomputed-gate route of a 3x3 stride-1 layer runs the row-window form (K tiles in another order): the same
    # gate exactly, the same values up to the summation order
    rw = both_routes()
    assert bool((rw[1][0][~gate] == 0).all())
    assert float((rw[1][0] - res[1][0]).abs().max()) <= 1e-5 * float(res[1][0].abs().max())
    assert float((rw[1][1] - res[1][1]).abs().max()) <= 1e-4 * float(res[1][1].abs().max())


@pytest.mark.parametrize("M,C", [(37, 64), (4 * 49, 2048), (1000, 256)])
def test_block_output_with_fused_shortcut_batchnorm(cuda, M, C):
    """osi_bn_apply_relu_mask2 = bn3 + (downsample BatchNorm applied on the fly) + ReLU + bitmask in one pass, bit-identical to
    materialising the normalised shortcut with osi_bn_apply first."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(M + C)
    y, ry = torch.randn(M, C, generator=g).to(cuda), torch.randn(M, C, generator=g).to(cuda)
    sc, sh, rsc, rsh = [(torch.randn(C, generator=g) * 0.5 + (1 if i % 2 == 0 else 0)).to(cuda) for i in range(4)]
    xd = torch.empty(M, C, device=cuda)
    N.check(L.osi_bn_apply(N.ptr(ry), None, N.ptr(rsc), N.ptr(rsh), N.ptr(xd), M, C, 0, T.S()))
    nb = L.osi_bn_relu_mask_bytes(M, C)
    out1, m1 = torch.empty(M, C, device=cuda), torch.zeros(nb, dtype=torch.uint8, device=cuda)
    out2, m2 = torch.empty(M, C, device=cuda), torch.zeros(nb, dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_apply_relu_mask(N.ptr(y), N.ptr(xd), N.ptr(sc), N.ptr(sh), N.ptr(out1), N.ptr(m1), M, C, T.S()))
    N.check(L.osi_bn_apply_relu_mask2(N.ptr(y), N.ptr(sc), N.ptr(sh), N.ptr(ry), N.ptr(rsc), N.ptr(rsh), N.ptr(out2), N.ptr(m2), M, C, T.S()))
    assert torch.equal(out1, out2) and torch.equal(m1, m2)
    ref = torch.relu(y.double() * sc.double() + sh.double() + ry.double() * rsc.double() + rsh.double())
    assert float((out2.double() - ref).abs().max()) <= 1e-5


@pytest.mark.parametrize("Cin,Cout,H,B,tile", [(256, 64, 14, 3, 0), (512, 128, 9, 2, 5), (1024, 256, 7, 5, 6), (2048, 512, 7, 3, 0)])
def test_conv1_recomputes_the_previous_block_output(cuda, Cin, Cout, H, B, tile):
    """osi_conv_fwd_act2: the A operand relu(x * scale + shift + res) — a whole identity-shortcut block output — recomputed in the
    loader gives exactly the bits of the plain convolution on the tensor osi_bn_apply_relu_mask materialises (same fma, add, max; same
    tile and K order), and its BatchNorm partials are those of the plain fused form."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(Cin + H)
    M = B * H * H
    y3 = (torch.randn(B, H, H, Cin, generator=g) * 1.3).to(cuda)
    res = torch.relu(torch.randn(B, H, H, Cin, generator=g)).to(cuda)
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(cuda), (torch.randn(Cin, generator=g) * 0.5).to(cuda)
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to(cuda)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, 1, 1, 0)
    out = torch.empty(M, Cin, device=cuda)
    mask = torch.zeros(L.osi_bn_relu_mask_bytes(M, Cin), dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_apply_relu_mask(N.ptr(y3), N.ptr(res), N.ptr(sc), N.ptr(sh), N.ptr(out), N.ptr(mask), M, Cin, T.S()))
    pb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
    ys, ps, Ps = [], [], []
    for fused in (False, True):
        y = torch.full((B, H, H, Cout), float("nan"), device=cuda)
        p = torch.full((pb // 4,), float("nan"), device=cuda)
        P, rows = ctypes.c_int(), ctypes.c_int()
        if fused:
            N.check(L.osi_conv_fwd_act2(ctypes.byref(d), N.ptr(y3), N.ptr(sc), N.ptr(sh), N.ptr(res), N.ptr(w), N.ptr(y), tile, N.ptr(p), pb,
                                        ctypes.byref(P), ctypes.byref(rows), T.S()), "osi_conv_fwd_act2")
        else:
            N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(out), N.ptr(w), N.ptr(y), tile if tile else 0, N.ptr(p), pb, ctypes.byref(P),
                                           ctypes.byref(rows), T.S()))
        ys.append(y); ps.append(p[:2 * P.value * Cout].clone()); Ps.append((P.value, rows.value))
    assert Ps[0] == Ps[1] and torch.equal(ys[0], ys[1]) and torch.equal(ps[0], ps[1])
    ref = F.conv2d(T.nchw(torch.relu(y3.double() * sc.double() + sh.double() + res.double())), T.oihw(w.double())).permute(0, 2, 3, 1)
    assert float((ys[1].double() - ref).abs().max()) <= (2e-6 + 6e-8 * Cin ** 0.5) * float(ref.abs().max()) + 1e-6
    # 3x3 / strided descriptors are refused: the input must be a block output feeding a 1x1 stride-1 conv
    d3 = N.ConvDesc.make(B, H, H, Cin, Cout, 3, 1, 1)
    assert L.osi_conv_fwd_act2(ctypes.byref(d3), N.ptr(y3), N.ptr(sc), N.ptr(sh), N.ptr(res), N.ptr(w), N.ptr(ys[0]), 0, None, 0, None, None, T.S()) == -1


@pytest.mark.parametrize("Cin,Cout,H,W,B", [(32, 64, 1, 1, 7), (64, 64, 2, 3, 5), (32, 128, 5, 9, 3), (64, 64, 14, 14, 4), (96, 64, 7, 31, 2),
                                            (64, 128, 28, 28, 3), (32, 64, 9, 56, 2), (64, 64, 6, 63, 1), (128, 192, 7, 7, 9)])
@pytest.mark.parametrize("fused", [False, True])
def test_all_taps_3x3_weight_gradient(cuda, Cin, Cout, H, W, B, fused):
    """k_conv_wgrad3 (one workgroup walks the nine taps of a stride-1 3x3 layer against one staged dY run and one X window): every
    window-pass variant (W <= 15 / 31 / 63), images narrower than the filter, non-square images, ragged runs and split ends, with and
    without the fused input activation — against torch's conv2d_weight in fp64, against the per-tap kernel (osi_set_tuning("wgrad3", 0))
    and against itself (bitwise reproducible)."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(Cin + Cout + H * 7 + W)
    x = (torch.randn(B, H, W, Cin, generator=g) * 1.2).to(cuda)
    dy = torch.randn(B, H, W, Cout, generator=g).to(cuda)
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(cuda), (torch.randn(Cin, generator=g) * 0.6).to(cuda)
    d = N.ConvDesc(B, H, W, Cin, H, W, Cout, 3, 3, 1, 1)

    def run():
        nb = L.osi_conv_wgrad_workspace(ctypes.byref(d))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=cuda)
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=cuda)
        if fused:
            N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(ws), nb, T.S()))
        else:
            N.check(L.osi_conv_wgrad(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(dw), N.ptr(ws), nb, T.S()))
        return dw
    v = ctypes.c_int()
    assert L.osi_get_tuning(b"wgrad3", ctypes.byref(v)) == 0 and v.value == 2
    a, a2 = run(), run()
    per_tap = None
    if Cin % 64 == 0:                      # the per-tap kernel takes input channels in 64s only
        N.check(L.osi_set_tuning(b"wgrad3", 0))
        try:
            per_tap = run()
        finally:
            N.check(L.osi_set_tuning(b"wgrad3", 2))
    act = torch.relu(x.double() * sc.double() + sh.double()) if fused else x.double()
    ref = torch.nn.grad.conv2d_weight(T.nchw(act), (Cout, Cin, 3, 3), T.nchw(dy.double()), 1, 1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max()) + 1e-30
    Kp = B * H * W
    assert torch.isfinite(a).all() and torch.equal(a, a2)
    assert float((a.double() - ref).abs().max()) <= (2e-6 + 6e-8 * Kp ** 0.5) * scale + 1e-6
    assert per_tap is None or float((a - per_tap).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("Cin,Cout,H,W,B", [(32, 64, 2, 2, 7), (64, 64, 4, 6, 5), (64, 128, 14, 14, 4), (128, 64, 28, 28, 3), (96, 64, 8, 30, 2),
                                            (64, 64, 56, 56, 2), (64, 192, 6, 62, 3), (128, 128, 10, 2, 9)])
@pytest.mark.parametrize("fused", [False, True])
def test_all_taps_3x3_stride2_weight_gradient(cuda, Cin, Cout, H, W, B, fused):
    """The stride-2 form of k_conv_wgrad3 (the three stage-entry 3x3 convolutions of ResNet v1.5: four parity sub-grid runs per window,
    five / six window passes): images of one output row / column, non-square images, ragged runs, batch boundaries inside a run, with and
    without the fused input activation — against torch's conv2d_weight in fp64, against the per-tap kernel and against itself."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(Cin + Cout + H * 7 + W + 1)
    x = (torch.randn(B, H, W, Cin, generator=g) * 1.2).to(cuda)
    d = N.ConvDesc.make(B, H, W, Cin, Cout, 3, 2, 1)
    assert (d.Ho, d.Wo) == (H // 2, W // 2)
    dy = torch.randn(B, d.Ho, d.Wo, Cout, generator=g).to(cuda)
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(cuda), (torch.randn(Cin, generator=g) * 0.6).to(cuda)

    def run():
        nb = L.osi_conv_wgrad_workspace(ctypes.byref(d))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=cuda)
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=cuda)
        if fused:
            N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(ws), nb, T.S()))
        else:
            N.check(L.osi_conv_wgrad(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(dw), N.ptr(ws), nb, T.S()))
        return dw
    a, a2 = run(), run()
    per_tap = None
    if Cin % 64 == 0:
        N.check(L.osi_set_tuning(b"wgrad3", 1))       # 1 = all-taps for stride 1 only: this layer takes the per-tap kernel
        try:
            per_tap = run()
        finally:
            N.check(L.osi_set_tuning(b"wgrad3", 2))
    act = torch.relu(x.double() * sc.double() + sh.double()) if fused else x.double()
    ref = torch.nn.grad.conv2d_weight(T.nchw(act), (Cout, Cin, 3, 3), T.nchw(dy.double()), 2, 1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max()) + 1e-30
    Kp = B * d.Ho * d.Wo
    assert torch.isfinite(a).all() and torch.equal(a, a2)
    assert float((a.double() - ref).abs().max()) <= (2e-6 + 6e-8 * Kp ** 0.5) * scale + 1e-6
    assert per_tap is None or float((a - per_tap).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("Cin,Cout,H,B", [(64, 64, 14, 3), (64, 256, 9, 3), (128, 512, 7, 5), (128, 64, 20, 2), (64, 128, 1, 70)])
@pytest.mark.parametrize("fused", [False, True])
def test_persistent_row_walker_for_short_k_1x1(cuda, Cin, Cout, H, B, fused):
    """k_conv1x1_rows (1x1 stride-1 convolutions with Cin = 64 / 128: one workgroup keeps its weight tile in LDS and walks row tiles,
    the next tile's rows prefetched behind the current tile's MFMAs and epilogue) against fp64 and against k_conv_fwd, which it must
    reproduce BIT FOR BIT (same K order per tile, same statistics arithmetic): ragged last tile, more walkers than row tiles, with and
    without the fused input activation, with and without statistics. osi_set_tuning("fwd_rows", 2) takes every eligible shape; the
    default (1) only the 56 x 56 layers of layer1 at production batch sizes (tests/test_production_shapes_gpu.py)."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(Cin + Cout + H + B)
    x = (torch.randn(B, H, H, Cin, generator=g) * 1.3 + 0.2).to(cuda)
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to(cuda)
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(cuda), (torch.randn(Cin, generator=g) * 0.5).to(cuda)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, 1, 1, 0)
    M = B * H * H
    nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))

    def run(stats):
        y = torch.full((B, H, H, Cout), float("nan"), device=cuda)
        ps = torch.full((nb // 4,), float("nan"), device=cuda)
        P, rows = ctypes.c_int(), ctypes.c_int()
        pa = (N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows)) if stats else (None, 0, None, None)
        if fused:
            N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, *pa, T.S()))
        elif stats:
            N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, *pa, T.S()))
        else:
            N.check(L.osi_conv_fwd(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, T.S()))
        return y, (ps[:2 * P.value * Cout].clone() if stats else None), (P.value, rows.value)

    v = ctypes.c_int()
    assert L.osi_get_tuning(b"fwd_rows", ctypes.byref(v)) == 0 and v.value == 1
    N.check(L.osi_set_tuning(b"tail_split", 0))          # the reference launch: k_conv_fwd, one pass (a K-split tail sums in another order)
    N.check(L.osi_set_tuning(b"fwd_rows", 0))
    try:
        y0, p0, g0 = run(True)
        N.check(L.osi_set_tuning(b"fwd_rows", 2))
        y1, p1, g1 = run(True)
        y2, _, _ = run(False)
        y3, p3, _ = run(True)
    finally:
        N.check(L.osi_set_tuning(b"fwd_rows", 1)); N.check(L.osi_set_tuning(b"tail_split", 1))
    a64 = torch.relu(x.double() * sc.double() + sh.double()) if fused else x.double()
    ref = F.conv2d(T.nchw(a64), T.oihw(w.double())).permute(0, 2, 3, 1)
    assert not torch.isnan(y1).any() and float((y1.double() - ref).abs().max()) <= (2e-6 + 6e-8 * Cin ** 0.5) * float(ref.abs().max()) + 1e-6
    assert g1 == g0 == ((M + 63) // 64, 64)
    assert torch.equal(y1, y0) and torch.equal(p1, p0), "the row walker reproduces k_conv_fwd bit for bit (output and BatchNorm partials)"
    assert torch.equal(y2, y1) and torch.equal(y3, y1) and torch.equal(p3, p1)


@pytest.mark.parametrize("Cin,Cout,H,B", [(64, 64, 7, 5), (192, 64, 13, 3), (64, 128, 56, 1), (128, 64, 8, 9), (64, 64, 31, 2)])
def test_row_windows_of_the_3x3_layers(cuda, Cin, Cout, H, B):
    """fwd_w3 / dgrad_w3 (default on): the 3x3 stride-1 forward and in-block input gradient stage one window per tap ROW in column-padded
    coordinates instead of one tile per tap. Same results as the per-tap form up to the order of the K tiles — at the smallest width the
    window is sized for (7), across image boundaries inside a tile, with a ragged last row tile, with and without the fused activation."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    x, sc, sh, w, act64 = _case(cuda, Cin, Cout, 3, 1, H, B)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, 3, 1, 1)
    M = B * H * H
    pb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
    ref = F.conv2d(T.nchw(act64), T.oihw(w.double()), None, 1, 1).permute(0, 2, 3, 1)
    ref_plain = F.conv2d(T.nchw(x.double()), T.oihw(w.double()), None, 1, 1).permute(0, 2, 3, 1)
    tol = (2e-6 + 6e-8 * (9 * Cin) ** 0.5)
    out = {}
    for knob in (1, 0):
        N.check(L.osi_set_tuning(b"fwd_w3", knob)); N.check(L.osi_set_tuning(b"dgrad_w3", knob))
        try:
            y = torch.full((B, H, H, Cout), float("nan"), device=cuda); ps = torch.full((pb // 4,), float("nan"), device=cuda)
            P, rows = ctypes.c_int(), ctypes.c_int()
            N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, N.ptr(ps), pb, ctypes.byref(P),
                                       ctypes.byref(rows), T.S()))
            yp = torch.full_like(y, float("nan"))
            N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(yp), 0, N.ptr(ps), pb, ctypes.byref(P), ctypes.byref(rows), T.S()))
            # in-block input gradient of the same layer: dx [B,H,H,Cin] from dy [B,H,H,Cout], gate recomputed from y0, sums over y0
            g = torch.Generator().manual_seed(Cin + H)
            dy = torch.randn(B, H, H, Cout, generator=g).to(cuda)
            y0 = (torch.randn(M, Cin, generator=g) * 2 + 0.3).to(cuda)
            mean0, inv0 = y0.mean(0), 1 / torch.sqrt(y0.var(0, unbiased=False) + 1e-5)
            fb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d)); parts = torch.full((fb // 4,), float("nan"), device=cuda)
            f = _Fusion(None, y0.data_ptr(), mean0.data_ptr(), inv0.data_ptr(), None, None, None, parts.data_ptr(), fb, sc.data_ptr(), sh.data_ptr())
            dx = torch.full((B, H, H, Cin), float("nan"), device=cuda)
            Pd = ctypes.c_int()
            N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), None, ctypes.byref(f), 0, ctypes.byref(Pd), T.S()))
            torch.cuda.synchronize()
            out[knob] = (y, yp, dx, parts[:2 * Pd.value * Cin].clone(), dy, y0)
        finally:
            N.check(L.osi_set_tuning(b"fwd_w3", 1)); N.check(L.osi_set_tuning(b"dgrad_w3", 1))
    y, yp, dx, parts, dy, y0 = out[1]
    assert float((y.double() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
    assert float((yp.double() - ref_plain).abs().max()) <= tol * float(ref_plain.abs().max()) + 1e-6
    gate = ((y0 * sc + sh) > 0).view(B, H, H, Cin)
    dx64 = F.conv_transpose2d(T.nchw(dy.double()), T.oihw(w.double()), None, 1, 1).permute(0, 2, 3, 1) * gate
    assert float((dx.double() - dx64).abs().max()) <= (2e-6 + 6e-8 * (9 * Cout) ** 0.5) * float(dx64.abs().max()) + 1e-6
    for a, b in zip(out[1][:4], out[0][:4]):          # against the per-tap form: the order of the K tiles only
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
