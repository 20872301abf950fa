"""Winograd F(2x2, 3x3) forms (csrc/conv_wino.hip) of the 3x3 / stride 1 / pad 1 convolutions of ResNet-50 — conv2 of the bottlenecks
without a stride, torchvision's resnet50 under /root/reference/openset_imagenet/model.py:17,37 — through the C ABI, against torch-CPU
fp64 `F.conv2d` / `conv2d_input` under the SAME per-kernel bound as the direct kernels (tests/test_production_shapes_gpu.py:
|err| <= (2e-6 + 6e-8 sqrt(K)) max|ref|; the Winograd form does not get a looser one), at the benchmarked batch (B = 128, every
network shape) and on ragged / odd geometries (a last unit past the tile count, 7 x 7 images whose tiles hang over the border).

  forward          osi_conv_fwd_wino with the fused input activation (what the executor issues for conv2) and plain; the epilogue's
                   BatchNorm partials (one per 16 tiles) merged in fp64 and finished by osi_bn_finalize_stats
  input gradient   osi_conv_dgrad_fused_wino: gate recomputed from the pre-BN tensor, exact zeros behind closed gates, sums of g
                   and g * xhat per 16 tiles
"""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from test_production_shapes_gpu import _bound, _check_partials_and_finalize, _col_stats, _cpu64, _gen, _xhat_sums

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _both_directions_cut():
    """The product default cuts the ragged round stream-K style in the FORWARD only (wino_streamk = 2: in the backward pass the side
    stream's weight gradients fill the idle CUs and the cut costs more than it returns); these tests run with 1 = both directions, so
    that the input gradient's pieces + fix-up stay covered. The whole-network tests run at the default, the knob matrix at 0, 1 and 3."""
    from openset_imagenet import _native as N
    L = N.lib()
    prev = ctypes.c_int()
    N.check(L.osi_get_tuning(b"wino_streamk", ctypes.byref(prev)))
    N.check(L.osi_set_tuning(b"wino_streamk", 1))
    yield
    N.check(L.osi_set_tuning(b"wino_streamk", prev.value))


# (C, H): the four 3x3 stride-1 layer shapes of the network
NETWORK = [(64, 56), (128, 28), (256, 14), (512, 7)]


def _fwd(cuda, B, H, W, Cin, Cout, act, seed):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = _gen(cuda, "wino-fwd", seed, B, H, W, Cin, Cout, act)
    d = N.ConvDesc.make(B, H, W, Cin, Cout, 3, 1, 1)
    assert L.osi_conv_wino_eligible(ctypes.byref(d), 0) == 1
    x = torch.randn(B, H, W, Cin, device=cuda, generator=g) * 1.2 + 0.3
    w = torch.randn(Cout, 3, 3, Cin, device=cuda, generator=g) / math.sqrt(Cin * 9)
    a64 = _cpu64(x)
    sc = sh = None
    if act:
        sc, sh = torch.rand(Cin, device=cuda, generator=g) + 0.5, torch.randn(Cin, device=cuda, generator=g) * 0.5
        a64 = torch.relu(a64 * _cpu64(sc) + _cpu64(sh))
    ref = F.conv2d(T.nchw(a64), T.oihw(_cpu64(w)), None, 1, 1).permute(0, 2, 3, 1).contiguous()
    wb = L.osi_conv_wino_workspace(ctypes.byref(d))
    assert wb >= 16 * Cin * Cout * 4
    ws = torch.empty(wb, dtype=torch.uint8, device=cuda)
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    Pn = (tiles + 15) // 16
    nb = (2 * Pn + 64) * Cout * 4
    ps = torch.full((nb // 4,), float("nan"), device=cuda)
    y = torch.full((B, H, W, Cout), float("nan"), device=cuda)
    P, rows = ctypes.c_int(), ctypes.c_int()
    N.check(L.osi_conv_fwd_wino(ctypes.byref(d), N.ptr(x), N.ptr(sc) if act else None, N.ptr(sh) if act else None, N.ptr(w), N.ptr(y),
                                N.ptr(ws), wb, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), T.S()), "osi_conv_fwd_wino")
    M = B * H * W
    err = float((_cpu64(y) - ref).abs().max())
    assert err <= _bound(Cin * 9, ref), f"{(B, H, W, Cin, Cout)}: {err:.3e} > {_bound(Cin * 9, ref):.3e}"
    assert P.value == Pn and P.value * rows.value == M, "one partial per 16 tiles, every one of the same pixel count"
    # the partial of group p covers the 2x2 pixel blocks of tiles [16 p, 16 p + 16), not rows [rows p, rows p + rows): merged as a set
    _check_partials_and_finalize(L, N, T, cuda, ps, nb, P.value, rows.value, M, Cout, ref.view(M, Cout), g)
    # no statistics requested (eval mode): same output
    y2 = torch.full_like(y, float("nan"))
    N.check(L.osi_conv_fwd_wino(ctypes.byref(d), N.ptr(x), N.ptr(sc) if act else None, N.ptr(sh) if act else None, N.ptr(w), N.ptr(y2),
                                N.ptr(ws), wb, None, 0, None, None, T.S()))
    assert torch.equal(y, y2)
    return err


@pytest.mark.parametrize("C,H", NETWORK)
def test_forward_network_shapes_at_the_benchmarked_batch(cuda, C, H):
    _fwd(cuda, 128, H, H, C, C, True, 0)


@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [(6, 7, 7, 64, 128, True),      # 96 tiles: the second unit is half empty; tiles hang over the border
                                                (4, 7, 7, 32, 64, False),      # Cin = 32: two K slices; plain input
                                                (8, 12, 20, 48, 64, True),     # non-square, Cin % 16 == 0 only
                                                (16, 14, 14, 64, 192, True),   # 784 tiles = 12.25 units, three column units
                                                (16, 14, 14, 128, 64, True)])  # 13 units of 8 K slices on 256 CUs: every unit cut into 8 stream-K pieces
def test_forward_ragged_and_odd_geometries(cuda, B, H, W, Cin, Cout, act):
    _fwd(cuda, B, H, W, Cin, Cout, act, 1)


def _dgrad(cuda, B, H, W, Cin, Cout, seed, partials=True):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = _gen(cuda, "wino-dgrad", seed, B, H, W, Cin, Cout)
    d = N.ConvDesc.make(B, H, W, Cin, Cout, 3, 1, 1)
    assert L.osi_conv_wino_eligible(ctypes.byref(d), 1) == 1
    M = B * H * W
    dy = torch.randn(B, H, W, Cout, device=cuda, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device=cuda, generator=g) / math.sqrt(Cout * 9)
    acc = torch.nn.grad.conv2d_input((B, Cin, H, W), T.oihw(_cpu64(w)), T.nchw(_cpu64(dy)), 1, 1).permute(0, 2, 3, 1).contiguous()
    # the producer's activation relu(bn(y0)) was never stored: gate = fma(y0, scale0, shift0) > 0, pre-activations kept 1e-3 away from
    # zero so that fp32 / fp64 agree on every decision
    sc, sh = torch.rand(Cin, device=cuda, generator=g) + 0.5, torch.randn(Cin, device=cuda, generator=g) * 0.5
    pre = torch.randn(M, Cin, device=cuda, generator=g)
    pre = torch.where(pre >= 0, pre.clamp_min(1e-3), pre.clamp_max(-1e-3))
    y0 = ((pre.double() - sh.double()) / sc.double()).float()
    gate = _cpu64(pre > 0).view(B, H, W, Cin)
    mean0, inv0 = _col_stats(y0)
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    Pn = (tiles + 15) // 16
    pb = 3 * Pn * Cin * 4
    parts = torch.full((pb // 4,), float("nan"), device=cuda)
    f = T.Fusion(None, y0.data_ptr(), mean0.data_ptr(), inv0.data_ptr(), None, None, None, parts.data_ptr() if partials else None, pb if partials else 0,
                 sc.data_ptr(), sh.data_ptr())
    wb = L.osi_conv_wino_workspace(ctypes.byref(d))
    ws = torch.empty(wb, dtype=torch.uint8, device=cuda)
    dx = torch.full((B, H, W, Cin), float("nan"), device=cuda)
    P = ctypes.c_int()
    N.check(L.osi_conv_dgrad_fused_wino(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), ctypes.byref(f), N.ptr(ws), wb, ctypes.byref(P), T.S()),
            "osi_conv_dgrad_fused_wino")
    ref = acc * gate
    got = _cpu64(dx)
    err = float((got - ref).abs().max())
    assert err <= _bound(Cout * 9, ref), f"{(B, H, W, Cin, Cout)}: {err:.3e} > {_bound(Cout * 9, ref):.3e}"
    assert bool((got[gate == 0] == 0).all()), "exact zeros behind a closed gate"
    assert P.value == Pn
    if partials:
        sg_want, sgx_want, l1 = _xhat_sums(ref.view(M, Cin), y0, mean0, inv0)
        p = _cpu64(parts[:2 * Pn * Cin]).view(2, Pn, Cin)
        l1g = ref.view(M, Cin).abs().sum(0)
        assert float(((p[0].sum(0) - sg_want).abs() / (l1g + 1e-3)).max()) <= 1e-5, "sum g"
        assert float(((p[1].sum(0) - sgx_want).abs() / (l1 + 1e-3)).max()) <= 2e-5, "sum g * xhat"
    return dx


@pytest.mark.parametrize("C,H", NETWORK)
def test_input_gradient_network_shapes_at_the_benchmarked_batch(cuda, C, H):
    _dgrad(cuda, 128, H, H, C, C, 0)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 14, 14, 64, 64),      # 147 tiles: the last statistics group and the last unit are ragged
                                            (6, 7, 7, 128, 64),       # tiles over the border; more input- than output-gradient channels
                                            (5, 10, 6, 64, 48),       # non-square; Cout % 16 == 0 only
                                            (5, 14, 14, 64, 160)])    # 10 K slices per unit, 4 units: stream-K pieces + fix-up with the fused epilogue
def test_input_gradient_ragged_and_odd_geometries(cuda, B, H, W, Cin, Cout):
    a = _dgrad(cuda, B, H, W, Cin, Cout, 2)
    b = _dgrad(cuda, B, H, W, Cin, Cout, 2, partials=False)
    assert torch.equal(a, b), "the sums are optional and do not change dx"


def test_eligibility_and_argument_checks(cuda):
    from openset_imagenet import _native as N
    L = N.lib()
    mk = N.ConvDesc.make
    ok = mk(128, 14, 14, 256, 256, 3, 1, 1)
    assert L.osi_conv_wino_eligible(ctypes.byref(ok), 0) == 1 and L.osi_conv_wino_eligible(ctypes.byref(ok), 1) == 1
    for bad in (mk(128, 28, 28, 256, 256, 3, 2, 1), mk(128, 14, 14, 256, 256, 1, 1, 0), mk(128, 14, 14, 40, 64, 3, 1, 1), mk(128, 14, 14, 64, 96, 3, 1, 1)):
        assert L.osi_conv_wino_eligible(ctypes.byref(bad), 0) == 0
    # forward: every 16-tile statistics group must hold the same number of pixels (3 images of 7 x 7 tiles = 147 tiles: not a multiple of 16)
    rag = mk(3, 14, 14, 64, 64, 3, 1, 1)
    assert L.osi_conv_wino_eligible(ctypes.byref(rag), 0) == 0 and L.osi_conv_wino_eligible(ctypes.byref(rag), 1) == 1
    t = torch.zeros(64, device=cuda)
    assert L.osi_conv_fwd_wino(ctypes.byref(rag), N.ptr(t), None, None, N.ptr(t), N.ptr(t), N.ptr(t), 1 << 20, None, 0, None, None, None) == -1
    # workspace too small
    assert L.osi_conv_fwd_wino(ctypes.byref(ok), N.ptr(t), None, None, N.ptr(t), N.ptr(t), N.ptr(t), 1024, None, 0, None, None, None) == -1


def test_executor_takes_the_winograd_forms(cuda):
    """The executor's conv2 forward / in-block input gradient run the Winograd kernels by default (knobs fwd_wino / dgrad_wino = 1), and the
    whole network agrees with the direct-kernel executor to fp32 rounding (the tight whole-network bars against the fp64 oracle are
    tests/test_production_shapes_gpu.py and tests/test_gate_pinned_gpu.py, which run with the defaults, i.e. with these kernels)."""
    from openset_imagenet import ResNet50, EntropicOpensetLoss, tools, _native as N
    L = N.lib()
    v = ctypes.c_int()
    for k in (b"fwd_wino", b"dgrad_wino", b"wgrad_wino"):
        N.check(L.osi_get_tuning(k, ctypes.byref(v)))
        assert v.value == 1
    outs = {}
    try:
        for mode in (1, 0):
            N.check(L.osi_set_tuning(b"fwd_wino", mode))
            N.check(L.osi_set_tuning(b"dgrad_wino", mode))
            N.check(L.osi_set_tuning(b"wgrad_wino", mode))
            tools.set_device_gpu(0)
            torch.manual_seed(3)
            model = tools.device(ResNet50(10, 10, False))
            x = torch.rand(8, 3, 64, 64, generator=torch.Generator().manual_seed(5))
            y = torch.tensor([0, -1, 3, 9, -1, 5, 2, 7])
            model.train()
            logits, _ = model(tools.device(x))
            EntropicOpensetLoss(10, 1.0)(logits, tools.device(y)).backward()
            torch.cuda.synchronize()
            outs[mode] = (logits.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()})
            del model
    finally:
        N.check(L.osi_set_tuning(b"fwd_wino", 1))
        N.check(L.osi_set_tuning(b"dgrad_wino", 1))
        N.check(L.osi_set_tuning(b"wgrad_wino", 1))
    # weight gradients of the two executors: the same sums in another order
    for k in outs[0][1]:
        a, b = outs[1][1][k], outs[0][1][k]
        assert float((a - b).norm() / (b.norm() + 1e-30)) <= 2e-2, k       # free-running: a few ReLU decisions may differ
    assert float((outs[1][0] - outs[0][0]).abs().max()) <= 5e-5
    assert not torch.equal(outs[1][0], outs[0][0]), "the two executors run different conv2 kernels"


def _wgrad(cuda, B, H, W, Cin, Cout, act, seed):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = _gen(cuda, "wino-wgrad", seed, B, H, W, Cin, Cout, act)
    d = N.ConvDesc.make(B, H, W, Cin, Cout, 3, 1, 1)
    nb = L.osi_conv_wgrad_wino_workspace(ctypes.byref(d))
    assert nb > 0
    x = torch.randn(B, H, W, Cin, device=cuda, generator=g) * 1.2 + 0.3
    dy = torch.randn(B, H, W, Cout, device=cuda, generator=g)
    a64 = _cpu64(x)
    sc = sh = None
    if act:
        sc, sh = torch.rand(Cin, device=cuda, generator=g) + 0.5, torch.randn(Cin, device=cuda, generator=g) * 0.5
        a64 = torch.relu(a64 * _cpu64(sc) + _cpu64(sh))
    ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
    dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=cuda)
    N.check(L.osi_conv_wgrad_wino(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc) if act else None, N.ptr(sh) if act else None, N.ptr(dw), N.ptr(ws), nb,
                                  T.S()), "osi_conv_wgrad_wino")
    ref = torch.nn.grad.conv2d_weight(T.nchw(a64), (Cout, Cin, 3, 3), T.nchw(_cpu64(dy)), 1, 1).permute(0, 2, 3, 1)
    err = float((_cpu64(dw) - ref).abs().max())
    Kp = B * H * W
    assert err <= _bound(Kp, ref), f"{(B, H, W, Cin, Cout)}: {err:.3e} > {_bound(Kp, ref):.3e}"
    dw2 = torch.full_like(dw, float("nan"))
    N.check(L.osi_conv_wgrad_wino(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc) if act else None, N.ptr(sh) if act else None, N.ptr(dw2), N.ptr(ws), nb,
                                  T.S()))
    assert torch.equal(dw, dw2), "split-K over the tile axis with a fixed-order reduce: bitwise reproducible"
    return err


@pytest.mark.parametrize("C,H", NETWORK)
def test_weight_gradient_network_shapes_at_the_benchmarked_batch(cuda, C, H):
    """osi_conv_wgrad_wino as the executor issues it for conv2 (fused input activation) against torch-CPU fp64 conv2d_weight under the
    per-kernel bound of the direct weight-gradient kernels (K = the pixels summed over)."""
    _wgrad(cuda, 128, H, H, C, C, True, 0)


@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [(6, 7, 7, 64, 128, True),      # tiles over the border, a ragged last K step
                                                (3, 14, 14, 128, 64, False),   # plain input; 147 tiles
                                                (5, 10, 6, 64, 64, True)])     # non-square
def test_weight_gradient_ragged_and_odd_geometries(cuda, B, H, W, Cin, Cout, act):
    _wgrad(cuda, B, H, W, Cin, Cout, act, 1)


def test_pretransformed_weights_give_the_same_bits(cuda):
    """osi_conv_wino_transform_weights + the `_pre` calls (what the executor issues: every 3x3 layer's weights transformed on the side stream
    at the start of a forward pass) = the self-contained calls, bit for bit."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    B, H, C = 16, 14, 128
    g = _gen(cuda, "wino-pre")
    d = N.ConvDesc.make(B, H, H, C, C, 3, 1, 1)
    x = torch.randn(B, H, H, C, device=cuda, generator=g); w = torch.randn(C, 3, 3, C, device=cuda, generator=g) * 0.05
    sc, sh = torch.rand(C, device=cuda, generator=g) + 0.5, torch.randn(C, device=cuda, generator=g) * 0.5
    wb = L.osi_conv_wino_workspace(ctypes.byref(d)); ws = torch.empty(wb, dtype=torch.uint8, device=cuda)
    ub, sb = L.osi_conv_wino_weights_bytes(ctypes.byref(d)), L.osi_conv_wino_slab_bytes()
    assert ub == 16 * C * C * 4 and ub + sb == wb
    u = torch.empty(ub, dtype=torch.uint8, device=cuda); slab = torch.empty(sb, dtype=torch.uint8, device=cuda)
    ya, yb = torch.full((B, H, H, C), float("nan"), device=cuda), torch.full((B, H, H, C), float("nan"), device=cuda)
    N.check(L.osi_conv_fwd_wino(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(ya), N.ptr(ws), wb, None, 0, None, None, T.S()))
    N.check(L.osi_conv_wino_transform_weights(ctypes.byref(d), N.ptr(w), 0, N.ptr(u), ub, T.S()))
    N.check(L.osi_conv_fwd_wino_pre(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(u), N.ptr(yb), N.ptr(slab), sb, None, 0, None, None, T.S()))
    assert torch.equal(ya, yb)
    y0 = torch.randn(B, H, H, C, device=cuda, generator=g)
    f = T.Fusion(None, y0.data_ptr(), None, None, None, None, None, None, 0, sc.data_ptr(), sh.data_ptr())
    P = ctypes.c_int()
    N.check(L.osi_conv_dgrad_fused_wino(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(ya), ctypes.byref(f), N.ptr(ws), wb, ctypes.byref(P), T.S()))
    N.check(L.osi_conv_wino_transform_weights(ctypes.byref(d), N.ptr(w), 1, N.ptr(u), ub, T.S()))
    N.check(L.osi_conv_dgrad_fused_wino_pre(ctypes.byref(d), N.ptr(x), N.ptr(u), N.ptr(yb), ctypes.byref(f), N.ptr(slab), sb, ctypes.byref(P), T.S()))
    assert torch.equal(ya, yb)


def test_plan_cus_above_the_device_do_not_resize_the_persistent_grid(cuda):
    """Knob tail_cus names the CU count the direct kernels' tail plan balances for (0 .. 4096, any chip); the Winograd kernels are one
    workgroup per HARDWARE CU with two slab slots each, so a larger figure must neither grow their grid nor let a workgroup write past the
    stream-K slab (osi_conv_wino_slab_bytes() is sized for the device's CUs). Same bits as the default plan, canary behind the slab intact."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    B, H, Cin, Cout = 16, 14, 128, 64                      # 13 units of 8 K slices: every unit is cut into stream-K pieces (slab in use)
    g = _gen(cuda, "wino-cus")
    d = N.ConvDesc.make(B, H, H, Cin, Cout, 3, 1, 1)
    x = torch.randn(B, H, H, Cin, device=cuda, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device=cuda, generator=g) * 0.05
    ub, sb = L.osi_conv_wino_weights_bytes(ctypes.byref(d)), L.osi_conv_wino_slab_bytes()
    u = torch.empty(ub, dtype=torch.uint8, device=cuda)
    N.check(L.osi_conv_wino_transform_weights(ctypes.byref(d), N.ptr(w), 0, N.ptr(u), ub, T.S()))
    guard = 1 << 20
    outs = []
    try:
        for cus in (0, 512, 4096):
            N.check(L.osi_set_tuning(b"tail_cus", cus))
            slab = torch.full((sb + guard,), 0xA5, dtype=torch.uint8, device=cuda)
            y = torch.full((B, H, H, Cout), float("nan"), device=cuda)
            N.check(L.osi_conv_fwd_wino_pre(ctypes.byref(d), N.ptr(x), None, None, N.ptr(u), N.ptr(y), N.ptr(slab), sb, None, 0, None, None, T.S()))
            torch.cuda.synchronize()
            assert bool((slab[sb:] == 0xA5).all()), f"tail_cus = {cus}: a workgroup wrote past the stream-K slab"
            outs.append(y)
    finally:
        N.check(L.osi_set_tuning(b"tail_cus", 0))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
