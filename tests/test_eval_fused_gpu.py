"""Inference forms (include/osi.h, "Inference forms"): validate() / get_arrays() of the reference run the model in eval mode
(/root/reference/openset_imagenet/train.py:142-234 — model.eval() at :166 / :207, BatchNorm on running statistics); there every
BatchNorm's scale / shift exist before its convolution is launched, so the convolution's epilogue applies BatchNorm (+ shortcut) (+ ReLU)
and the pre-BN tensor, the block-output pass and the ReLU bitmask disappear.

  osi_conv_fwd_epilogue            = osi_conv_fwd + osi_bn_apply on the same accumulators, BIT-identical (and within the direct kernels'
                                     bound of torch-CPU fp64), on every launch form the executor's plan takes: row walker, 64x64 / 64x128
                                     tiles, stride-2 3x3 and 1x1, row windows, K-split tail (slab given) — with / without shortcut and ReLU
  osi_conv_fwd_wino_epilogue_pre   = osi_conv_fwd_wino_pre + osi_bn_apply, bit-identical
  osi_bn_eval_coeffs_multi         = osi_bn_eval_coeffs per layer, bit-identical
  executor, option eval_fused      1 (default) vs 0 (the training topology on running statistics): identical logits / features, at a small
                                     geometry and at the benchmarked one (B = 128, 224 x 224) — where the fused forward is also held to the
                                     1e-4 logit bar against the fp64 oracle in eval mode
"""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from test_production_shapes_gpu import _bound, _cpu64, _gen

pytestmark = pytest.mark.gpu


def _epi(N, sc, sh, res, relu):
    return N.ConvEpilogue(sc.data_ptr(), sh.data_ptr(), None if res is None else res.data_ptr(), int(relu))


def _unfused(L, N, T, d, x, w, sc, sh, res, relu, M, Cout, stats_ws=None):
    """the training topology on the same inputs: conv -> pre-BN tensor -> osi_bn_apply. With `stats_ws` the convolution runs as
    osi_conv_fwd_bnstats, i.e. with the K-split tail the plan gives a launch that has workspace for its slab."""
    y = torch.full((M, Cout), float("nan"), device=x.device)
    if stats_ws is None:
        N.check(L.osi_conv_fwd(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, T.S()))
    else:
        P, rows = ctypes.c_int(), ctypes.c_int()
        N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, N.ptr(stats_ws), stats_ws.numel(), ctypes.byref(P),
                                       ctypes.byref(rows), T.S()))
    out = torch.full((M, Cout), float("nan"), device=x.device)
    N.check(L.osi_bn_apply(N.ptr(y), N.ptr(res), N.ptr(sc), N.ptr(sh), N.ptr(out), M, Cout, int(relu), T.S()))
    return out


# (B, H, Cin, Cout, k, stride, shortcut, relu, tail): one case per launch form of osi_conv_fwd_epilogue's plan
FORMS = [
    pytest.param(128, 56, 64, 256, 1, 1, True, True, False, id="rows-64-256-shortcut"),        # persistent row walker (layer1 conv3)
    pytest.param(128, 56, 64, 64, 1, 1, False, True, False, id="rows-64-64"),                  # layer1.0 conv1
    pytest.param(128, 28, 128, 512, 1, 1, True, True, False, id="64x64-128-512-shortcut"),      # layer2 conv3
    pytest.param(128, 56, 256, 512, 1, 2, False, False, False, id="ds-256-512-s2-norelu"),      # projection shortcut: BatchNorm only
    pytest.param(128, 56, 128, 128, 3, 2, False, True, False, id="3x3s2-128"),                  # layer2.0 conv2
    pytest.param(128, 7, 2048, 512, 1, 1, False, True, False, id="64x128-2048-512"),            # few tiles: the wide column tile
    pytest.param(128, 7, 2048, 512, 1, 1, False, True, True, id="tail-2048-512"),               # the same launch with its K-split tail
    pytest.param(128, 7, 512, 2048, 1, 1, True, True, False, id="64x64-512-2048-shortcut"),     # layer4 conv3 (12.25 rounds: no tail split)
    pytest.param(128, 14, 1024, 256, 1, 1, True, True, True, id="tail-1024-256-shortcut"),      # 6.125 rounds: K-split tail + shortcut in the fix-up
    pytest.param(128, 14, 256, 256, 3, 1, False, True, True, id="w3-tail-256"),                 # row windows + K-split tail (Winograd off)
    pytest.param(8, 28, 128, 128, 3, 1, False, True, False, id="w3-128-small"),                 # row windows, single pass
    pytest.param(5, 9, 64, 192, 1, 1, True, False, False, id="ragged-64-192"),                  # M = 405: a ragged last row tile
]


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,shortcut,relu,tail", FORMS)
def test_conv_epilogue_equals_conv_then_bn_apply(cuda, B, H, Cin, Cout, k, stride, shortcut, relu, tail):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = _gen(cuda, "epi", B, H, Cin, Cout, k, stride)
    pad = k // 2
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, stride, pad)
    M = B * d.Ho * d.Wo
    x = torch.rand(B, H, H, Cin, device=cuda, generator=g)              # a finished activation (post-ReLU: non-negative)
    w = torch.randn(Cout, k, k, Cin, device=cuda, generator=g) / math.sqrt(Cin * k * k)
    sc = torch.rand(Cout, device=cuda, generator=g) + 0.5
    sh = torch.randn(Cout, device=cuda, generator=g) * 0.3
    res = torch.randn(M, Cout, device=cuda, generator=g) if shortcut else None
    nb = L.osi_conv_fwd_epilogue_workspace(ctypes.byref(d)) if tail else 0
    if tail:
        assert nb > 0, "this shape's plan has a K-split tail at a 256-CU chip"
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=cuda)
    out = torch.full((M, Cout), float("nan"), device=cuda)
    e = _epi(N, sc, sh, res, relu)
    N.check(L.osi_conv_fwd_epilogue(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(out), ctypes.byref(e), N.ptr(ws) if tail else None, nb, T.S()),
            "osi_conv_fwd_epilogue")
    stats_ws = None
    if tail:
        stats_ws = torch.empty(L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d)), dtype=torch.uint8, device=cuda)
    want = _unfused(L, N, T, d, x, w, sc, sh, res, relu, M, Cout, stats_ws)
    assert torch.equal(out, want), f"max diff {float((out - want).abs().max()):.3e}"
    # and against fp64: the direct kernels' bound on the convolution, carried through the affine
    conv64 = F.conv2d(T.nchw(_cpu64(x)), T.oihw(_cpu64(w)), None, stride, pad).permute(0, 2, 3, 1).reshape(M, Cout)
    ref = conv64 * _cpu64(sc) + _cpu64(sh)
    if shortcut:
        ref = ref + _cpu64(res)
    if relu:
        ref = torch.relu(ref)
    bound = _bound(Cin * k * k, conv64) * float(sc.max()) + 4e-7 * float(ref.abs().max())
    err = float((_cpu64(out) - ref).abs().max())
    assert err <= bound, f"{err:.3e} > {bound:.3e}"
    if relu:
        assert float(out.min()) >= 0.0


def test_conv_epilogue_argument_checks(cuda):
    from openset_imagenet import _native as N
    L = N.lib()
    t = torch.zeros(4096, device=cuda)
    d = N.ConvDesc.make(2, 8, 8, 64, 64, 1, 1, 0)
    e = N.ConvEpilogue(t.data_ptr(), t.data_ptr(), None, 1)
    assert L.osi_conv_fwd_epilogue(ctypes.byref(d), N.ptr(t), N.ptr(t), N.ptr(t), None, None, 0, None) == -1
    bad = N.ConvEpilogue(None, t.data_ptr(), None, 1)
    assert L.osi_conv_fwd_epilogue(ctypes.byref(d), N.ptr(t), N.ptr(t), N.ptr(t), ctypes.byref(bad), None, 0, None) == -1
    alias = N.ConvEpilogue(t.data_ptr(), t.data_ptr(), t.data_ptr(), 1)           # residual must not alias out
    assert L.osi_conv_fwd_epilogue(ctypes.byref(d), N.ptr(t), N.ptr(t), N.ptr(t), ctypes.byref(alias), None, 0, None) == -1
    stem = N.ConvDesc.make(2, 32, 32, 4, 64, 7, 2, 3)
    assert L.osi_conv_fwd_epilogue(ctypes.byref(stem), N.ptr(t), N.ptr(t), N.ptr(t), ctypes.byref(e), None, 0, None) == -1
    assert L.osi_conv_fwd_epilogue_workspace(ctypes.byref(stem)) == 0
    # Winograd twin: no shortcut
    dw = N.ConvDesc.make(16, 14, 14, 64, 64, 3, 1, 1)
    assert L.osi_conv_fwd_wino_epilogue_pre(ctypes.byref(dw), N.ptr(t), N.ptr(t), N.ptr(t), ctypes.byref(alias), N.ptr(t), 1 << 30, None) == -1


@pytest.mark.parametrize("B,H,C,Cout", [(128, 56, 64, 64), (128, 28, 128, 128), (128, 14, 256, 256), (128, 7, 512, 512),      # the network's four
                                        (16, 14, 128, 64), (6, 7, 64, 128)])                                                   # stream-K pieces; tiles over the border
def test_winograd_epilogue_equals_winograd_then_bn_apply(cuda, B, H, C, Cout):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = _gen(cuda, "wino-epi", B, H, C, Cout)
    d = N.ConvDesc.make(B, H, H, C, Cout, 3, 1, 1)
    assert L.osi_conv_wino_eligible(ctypes.byref(d), 0) == 1
    M = B * H * H
    x = torch.rand(B, H, H, C, device=cuda, generator=g)
    w = torch.randn(Cout, 3, 3, C, device=cuda, generator=g) / math.sqrt(C * 9)
    sc = torch.rand(Cout, device=cuda, generator=g) + 0.5
    sh = torch.randn(Cout, device=cuda, generator=g) * 0.3
    ub, sb = L.osi_conv_wino_weights_bytes(ctypes.byref(d)), L.osi_conv_wino_slab_bytes()
    u = torch.empty(ub, dtype=torch.uint8, device=cuda); slab = torch.empty(sb, dtype=torch.uint8, device=cuda)
    N.check(L.osi_conv_wino_transform_weights(ctypes.byref(d), N.ptr(w), 0, N.ptr(u), ub, T.S()))
    for relu in (1, 0):
        out = torch.full((M, Cout), float("nan"), device=cuda)
        e = _epi(N, sc, sh, None, relu)
        N.check(L.osi_conv_fwd_wino_epilogue_pre(ctypes.byref(d), N.ptr(x), N.ptr(u), N.ptr(out), ctypes.byref(e), N.ptr(slab), sb, T.S()),
                "osi_conv_fwd_wino_epilogue_pre")
        y = torch.full((M, Cout), float("nan"), device=cuda)
        N.check(L.osi_conv_fwd_wino_pre(ctypes.byref(d), N.ptr(x), None, None, N.ptr(u), N.ptr(y), N.ptr(slab), sb, None, 0, None, None, T.S()))
        want = torch.full((M, Cout), float("nan"), device=cuda)
        N.check(L.osi_bn_apply(N.ptr(y), None, N.ptr(sc), N.ptr(sh), N.ptr(want), M, Cout, relu, T.S()))
        assert torch.equal(out, want), f"relu={relu}: max diff {float((out - want).abs().max()):.3e}"
    conv64 = F.conv2d(T.nchw(_cpu64(x)), T.oihw(_cpu64(w)), None, 1, 1).permute(0, 2, 3, 1).reshape(M, Cout)
    ref = conv64 * _cpu64(sc) + _cpu64(sh)
    bound = _bound(C * 9, conv64) * float(sc.max()) + 4e-7 * float(ref.abs().max())
    assert float((_cpu64(out) - ref).abs().max()) <= bound


def test_eval_coefficients_of_all_layers_in_one_launch(cuda):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = _gen(cuda, "coeffs")
    Cs = [64, 64, 256, 128, 512, 1024, 2048, 4, 300] * 7 + [64]          # 64 layers = OSI_BN_MULTI_MAX, odd channel counts included
    layers, keep, single = (N.BnEvalLayer * len(Cs))(), [], []
    for i, C in enumerate(Cs):
        rm, rv = torch.randn(C, device=cuda, generator=g) * 0.2, torch.rand(C, device=cuda, generator=g) + 0.3
        ga, be = torch.rand(C, device=cuda, generator=g) + 0.5, torch.randn(C, device=cuda, generator=g) * 0.3
        sc, sh = torch.full((C,), float("nan"), device=cuda), torch.full((C,), float("nan"), device=cuda)
        layers[i] = N.BnEvalLayer(rm.data_ptr(), rv.data_ptr(), ga.data_ptr(), be.data_ptr(), sc.data_ptr(), sh.data_ptr(), C)
        s1, h1 = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
        N.check(L.osi_bn_eval_coeffs(N.ptr(rm), N.ptr(rv), N.ptr(ga), N.ptr(be), 1e-5, C, N.ptr(s1), N.ptr(h1), T.S()))
        keep.append((rm, rv, ga, be, sc, sh)); single.append((s1, h1))
    N.check(L.osi_bn_eval_coeffs_multi(layers, len(Cs), 1e-5, T.S()), "osi_bn_eval_coeffs_multi")
    for (rm, rv, ga, be, sc, sh), (s1, h1) in zip(keep, single):
        assert torch.equal(sc, s1) and torch.equal(sh, h1)
        want = _cpu64(ga) / torch.sqrt(_cpu64(rv) + 1e-5)
        assert float((_cpu64(sc) / want - 1).abs().max()) <= 1e-6
    assert L.osi_bn_eval_coeffs_multi(layers, 65, 1e-5, None) == -1 and L.osi_bn_eval_coeffs_multi(layers, 0, 1e-5, None) == -1


@pytest.fixture
def no_tail_split():
    """The training topology in eval mode has no workspace for the slab of a K-split tail (its statistics workspace is not passed), the
    inference forms do and take the plan's split: another summation order in the ragged round's tiles. The bit-for-bit comparisons of
    the two executors therefore run with the tail split off in both (a plan knob: set before the executor is created); the comparisons
    against the fp64 oracle run at the defaults."""
    from openset_imagenet import _native as N
    N.check(N.lib().osi_set_tuning(b"tail_split", 0))
    yield
    N.check(N.lib().osi_set_tuning(b"tail_split", 1))


def _eval_pair(cuda, model, x):
    """logits / features of the fused inference forward and of the training topology on running statistics (eval_fused = 0)"""
    from openset_imagenet import _native as N
    model.eval()
    outs = {}
    for fused in (1, 0):
        net = model._net(x.shape[0], x.shape[2], x.shape[3])
        N.check(N.lib().osi_resnet50_set_option(net.h, b"eval_fused", fused))
        with torch.no_grad():
            lg, ft = model(x)
        torch.cuda.synchronize()
        outs[fused] = (lg.clone(), ft.clone())
    N.check(N.lib().osi_resnet50_set_option(model._net(x.shape[0], x.shape[2], x.shape[3]).h, b"eval_fused", 1))
    return outs


@pytest.mark.parametrize("B,HW,C,after_training", [(8, 64, 12, False), (8, 64, 12, True), (3, 96, 5, False)])
def test_fused_inference_forward_equals_the_training_topology(cuda, no_tail_split, B, HW, C, after_training):
    """Same fmas on the same accumulators: identical outputs. `after_training`: a training step ran first, so the executor owns its side
    stream (projection shortcut beside the main branch, Winograd weight transforms aside) — the schedule changes, the bits do not."""
    from openset_imagenet import ResNet50, EntropicOpensetLoss
    from oracle import resnet50_oracle as R
    gen = torch.Generator().manual_seed(77)
    sd = R.randomize_bn(R.init_state(C, C, True, generator=gen), generator=gen)
    model = ResNet50(C, C, True)
    model.load_state_dict(sd)
    model = model.to(cuda)
    x = torch.rand(B, 3, HW, HW, generator=gen).to(cuda)
    if after_training:
        model.train()
        lg, _ = model(x)
        EntropicOpensetLoss(C, 1.0)(lg, torch.randint(-1, C, (B,), generator=gen).to(cuda)).backward()
        model.load_state_dict(sd)           # undo the running-statistics update: both forwards below score the same model
    outs = _eval_pair(cuda, model, x)
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    ref = R.forward({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, x.cpu().double(), training=False)
    assert float((outs[1][0].cpu().double() - ref[0]).abs().max()) <= 1e-4 * max(1.0, float(ref[0].abs().max()))
    # the inference forward (the default again) leaves no backward state and no ReLU / arg-max decisions behind: both read-outs are refused
    # (real buffers throughout — a refused call launches nothing, an accepted one must still be harmless)
    from openset_imagenet import _native as N
    with torch.no_grad():
        model(x)
    net = model._net(B, HW, HW)
    Cg, Hg, Wg = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    N.check(N.lib().osi_resnet50_debug_gate_shape(net.h, 1, ctypes.byref(Cg), ctypes.byref(Hg), ctypes.byref(Wg)))
    gate = torch.zeros(B, Cg.value, Hg.value, Wg.value, dtype=torch.uint8, device=cuda)
    assert N.lib().osi_resnet50_debug_gate(net.h, N.ptr(model._ws), 1, N.ptr(gate), None, torch.cuda.current_stream().cuda_stream) == -3
    dl = torch.zeros(B, C, device=cuda)
    assert N.lib().osi_resnet50_backward(net.h, N.ptr(model._flat_params), N.ptr(model._flat_grads), N.ptr(model._ws), N.ptr(dl), None, 0, 1,
                                         torch.cuda.current_stream().cuda_stream) == -3
    torch.cuda.synchronize()


def test_fused_inference_forward_at_the_benchmarked_batch(cuda):
    """B = 128, 224 x 224, C = 30 (Protocol 2), default plans: fused = unfused to fp32 rounding (the K-split tails of the fused launches sum
    in another order), and max |logit - fp64 oracle (eval mode)| <= 1e-4 relative to the logit scale (running statistics drawn at random:
    logits are O(1..100), unlike the O(1) train-mode logits of an initialised network)."""
    import time
    from openset_imagenet import ResNet50
    from oracle import resnet50_oracle as R
    from openset_imagenet import _native as N
    C, B = 30, 128
    gen = torch.Generator().manual_seed(4321)
    sd = R.randomize_bn(R.init_state(C, C, False, generator=gen), generator=gen)
    x = torch.rand(B, 3, 224, 224, generator=gen)
    xd = x.to(cuda)
    try:      # the same plans in both executors (no K-split tails): bit for bit
        N.check(N.lib().osi_set_tuning(b"tail_split", 0))
        model = ResNet50(C, C, False)
        model.load_state_dict(sd)
        model = model.to(cuda)
        outs = _eval_pair(cuda, model, xd)
        assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
        del model
    finally:
        N.check(N.lib().osi_set_tuning(b"tail_split", 1))
    model = ResNet50(C, C, False)
    model.load_state_dict(sd)
    model = model.to(cuda)
    outs = _eval_pair(cuda, model, xd)
    assert float((outs[1][0] - outs[0][0]).abs().max()) <= 2e-6 * float(outs[0][0].abs().max())
    t0 = time.time()
    with torch.no_grad():
        rl, rf = R.forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double(), training=False)
    scale = max(1.0, float(rl.abs().max()))
    e_l, e_f = float((outs[1][0].cpu().double() - rl).abs().max()), float((outs[1][1].cpu().double() - rf).abs().max())
    print(f"fp64 oracle eval forward at B={B}: {time.time() - t0:.1f} s; max|logit - oracle| = {e_l:.2e} at |logit| <= {scale:.1f}, features {e_f:.2e}")
    assert e_l <= 1e-4 * scale and e_f <= 1e-4 * max(1.0, float(rf.abs().max()))
