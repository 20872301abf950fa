"""GPU parity of every HIP kernel against torch-CPU arithmetic (fp32, with an fp64 arbiter), through the C ABI.

Tolerances: the MFMA f32 path is an exact-f32 fma chain, so a conv differs from ATen's CPU conv only by summation order:
|err| <= ~1e-6 * sum|a*b|. Tests bound the error relative to the fp64 result and require it to be no worse than a few times
the error torch-CPU-fp32 itself makes against fp64.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _err(a, ref64):
    return float((a.double().cpu() - ref64).abs().max())


def _check_vs64(got, ref32, ref64, what, slack=4.0, floor=2e-6):
    """got must be as close to the fp64 truth as torch-CPU fp32 is (times slack), relative to the tensor's scale."""
    scale = float(ref64.abs().max()) + 1e-30
    e_got, e_ref = _err(got, ref64) / scale, _err(ref32, ref64) / scale
    assert e_got <= max(slack * e_ref, floor), f"{what}: rel err {e_got:.3e} vs torch-cpu-fp32 {e_ref:.3e}"


# (Cin, Cout, k, stride, pad, H) — every (k, stride) family of SURVEY.md Appendix A on reduced shapes, + ragged M
CONV_CASES = [
    (64, 64, 1, 1, 0, 14), (64, 256, 1, 1, 0, 9), (256, 64, 1, 1, 0, 12), (256, 128, 1, 1, 0, 7),
    (64, 64, 3, 1, 1, 14), (128, 128, 3, 2, 1, 14), (128, 128, 3, 2, 1, 9), (256, 512, 1, 2, 0, 14), (64, 128, 1, 2, 0, 7),
    (512, 512, 3, 1, 1, 7), (2048, 512, 1, 1, 0, 7), (512, 2048, 1, 1, 0, 7), (128, 64, 3, 1, 1, 5),
]


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H", CONV_CASES)
@pytest.mark.parametrize("B", [3])
def test_conv_fwd_dgrad_wgrad(cuda, Cin, Cout, k, stride, pad, H, B):
    import osi_testlib as T
    g = torch.Generator().manual_seed(Cin * 7 + Cout + k + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    y32 = F.conv2d(x, w, None, stride, pad)
    y64 = F.conv2d(x.double(), w.double(), None, stride, pad)
    dy = torch.randn(y32.shape, generator=g)
    dx32 = torch.nn.grad.conv2d_input(x.shape, w, dy, stride, pad)
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride, pad)
    dw32 = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride, pad)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride, pad)

    xg, wg, dyg = T.nhwc(x).to(cuda), T.krsc(w).to(cuda), T.nhwc(dy).to(cuda)
    # an exact-f32 fma chain of length K: error grows ~ sqrt(K) * 2^-24 relative to the output scale
    fl = lambda K: 2e-6 + 6e-8 * K ** 0.5
    tiles = [0, 4] + ([1, 3] if Cout % 128 == 0 else []) + [2]
    for tile in tiles:
        y = T.conv_fwd(xg, wg, k, stride, pad, tile)
        _check_vs64(T.nchw(y), y32, y64, f"conv fwd tile {tile}", floor=fl(Cin * k * k))
    tiles = [0, 4] + ([1, 3] if Cin % 128 == 0 else []) + [2]
    for tile in tiles:
        dx = T.conv_dgrad(dyg, wg, H, H, k, stride, pad, tile=tile)
        assert not torch.isnan(dx).any(), "dgrad left input-gradient elements unwritten"
        _check_vs64(T.nchw(dx), dx32, dx64, f"conv dgrad tile {tile}", floor=fl(Cout * k * k))
    # accumulate mode: dx = base + dgrad
    base = torch.randn(B, H, H, Cin, generator=g)
    acc = T.conv_dgrad(dyg, wg, H, H, k, stride, pad, accumulate_into=base.to(cuda).clone())
    _check_vs64(T.nchw(acc), dx32 + T.nchw(base), dx64 + T.nchw(base).double(), "conv dgrad accumulate", floor=fl(Cout * k * k))
    dw = T.conv_wgrad(dyg, xg, k, stride, pad)
    assert not torch.isnan(dw).any()
    _check_vs64(T.oihw(dw), dw32, dw64, "conv wgrad", floor=fl(B * y32.shape[2] * y32.shape[3]))


def test_conv_large_m_splitk(cuda):
    """A shape whose weight gradient needs a deep split-K (K = B*Ho*Wo = 50176) and ragged last split."""
    import osi_testlib as T
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H = 4, 64, 64, 112
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24
    dy = torch.randn(B, Cout, H, H, generator=g)
    dw32 = torch.nn.grad.conv2d_weight(x, w.shape, dy, 1, 1)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, 1)
    dw = T.conv_wgrad(T.nhwc(dy).to(cuda), T.nhwc(x).to(cuda), 3, 1, 1)
    _check_vs64(T.oihw(dw), dw32, dw64, "wgrad split-K", slack=6.0)
    dw2 = T.conv_wgrad(T.nhwc(dy).to(cuda), T.nhwc(x).to(cuda), 3, 1, 1)
    assert torch.equal(dw, dw2), "split-K reduction must be bitwise reproducible"


@pytest.mark.parametrize("B,H", [(2, 32), (3, 45)])
def test_stem(cuda, B, H):
    """7x7 stride-2 stem through the NCHW->NHWC4 staging + packed weights, fwd and wgrad."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    g = torch.Generator().manual_seed(H)
    x = torch.rand(B, 3, H, H, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    y32, y64 = F.conv2d(x, w, None, 2, 3), F.conv2d(x.double(), w.double(), None, 2, 3)
    xg = x.to(cuda)
    x4 = torch.empty(B, H, H, 4, device=cuda)
    N.check(N.lib().osi_nchw3_to_nhwc4(N.ptr(xg), N.ptr(x4), B, H, H, T.S()))
    assert torch.equal(x4[..., :3].cpu(), x.permute(0, 2, 3, 1)) and float(x4[..., 3].abs().max()) == 0
    wk = T.krsc(w).to(cuda)                       # [64][7][7][3] — the parameter's physical layout
    wp = torch.empty(64, 224, device=cuda)
    N.check(N.lib().osi_stem_weight_pack(N.ptr(wk), N.ptr(wp), 64, T.S()))
    y = T.conv_fwd(x4, wp, 7, 2, 3)
    _check_vs64(T.nchw(y), y32, y64, "stem fwd")
    dy = torch.randn(y32.shape, generator=g)
    dw32 = torch.nn.grad.conv2d_weight(x, w.shape, dy, 2, 3)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 2, 3)
    gp = T.conv_wgrad(T.nhwc(dy).to(cuda), x4, 7, 2, 3)
    gk = torch.empty(64, 7, 7, 3, device=cuda)
    N.check(N.lib().osi_stem_grad_unpack(N.ptr(gp), N.ptr(gk), 64, T.S()))
    _check_vs64(T.oihw(gk), dw32, dw64, "stem wgrad")


@pytest.mark.parametrize("B,C,H", [(4, 64, 14), (3, 256, 7), (2, 2048, 3), (5, 12, 6), (2, 64, 56)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_batchnorm(cuda, B, C, H, relu, res):
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(C + H)
    y = torch.randn(B, C, H, H, generator=g) * 2 + torch.randn(1, C, 1, 1, generator=g) * 5   # non-zero means
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm0, rv0 = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    resid = torch.randn(B, C, H, H, generator=g) if res else None
    dout = torch.randn(B, C, H, H, generator=g)

    def ref(dt):
        yy = y.to(dt).clone().requires_grad_(True)
        ga, be = gamma.to(dt).clone().requires_grad_(True), beta.to(dt).clone().requires_grad_(True)
        rm, rv = rm0.to(dt).clone(), rv0.to(dt).clone()
        o = F.batch_norm(yy, rm, rv, ga, be, True, 0.1, 1e-5)
        if res:
            o = o + resid.to(dt)
        if relu:
            o = F.relu(o)
        o.backward(dout.to(dt))
        return o.detach(), yy.grad, ga.grad, be.grad, rm, rv
    r32, r64 = ref(torch.float32), ref(torch.float64)

    M = B * H * H
    yg = T.nhwc(y).to(cuda)
    dev = lambda t: t.to(cuda).contiguous()
    ga, be, rm, rv = dev(gamma), dev(beta), dev(rm0), dev(rv0)
    mean, invstd, scale, shift = (torch.empty(C, device=cuda) for _ in range(4))
    wsb = max(L.osi_bn_workspace(M, C), L.osi_bn_backward_workspace(M, C))
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_train_stats(N.ptr(yg), M, C, N.ptr(ga), N.ptr(be), 1e-5, 0.1, N.ptr(rm), N.ptr(rv), N.ptr(mean), N.ptr(invstd),
                                 N.ptr(scale), N.ptr(shift), N.ptr(ws), wsb, T.S()))
    out = torch.empty_like(yg)
    rg = T.nhwc(resid).to(cuda) if res else None
    N.check(L.osi_bn_apply(N.ptr(yg), N.ptr(rg), N.ptr(scale), N.ptr(shift), N.ptr(out), M, C, int(relu), T.S()))
    _check_vs64(T.nchw(out), r32[0], r64[0], "bn out")
    _check_vs64(rm, r32[4], r64[4], "running_mean")
    _check_vs64(rv, r32[5], r64[5], "running_var")
    dg, db = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    dy = torch.empty_like(yg)
    gm = torch.empty_like(yg)
    dog = T.nhwc(dout).to(cuda)
    N.check(L.osi_bn_backward(N.ptr(dog), N.ptr(out) if relu else None, N.ptr(yg), N.ptr(mean), N.ptr(invstd), N.ptr(ga), N.ptr(dy),
                              N.ptr(gm), N.ptr(dg), N.ptr(db), M, C, N.ptr(ws), wsb, T.S()))
    _check_vs64(T.nchw(dy), r32[1], r64[1], "bn dy", slack=8.0, floor=5e-6)
    _check_vs64(dg, r32[2], r64[2], "bn dgamma", slack=8.0, floor=5e-6)
    _check_vs64(db, r32[3], r64[3], "bn dbeta", slack=8.0, floor=5e-6)
    mask = (r64[0] > 0) if relu else torch.ones_like(r64[0], dtype=torch.bool)
    assert torch.allclose(T.nchw(gm).cpu(), dout * mask, atol=0), "masked upstream gradient"
    # in-place form (dy aliases dout) gives the same bits
    N.check(L.osi_bn_backward(N.ptr(dog), N.ptr(out) if relu else None, N.ptr(yg), N.ptr(mean), N.ptr(invstd), N.ptr(ga), N.ptr(dog),
                              None, N.ptr(dg), N.ptr(db), M, C, N.ptr(ws), wsb, T.S()))
    assert torch.equal(dog, dy)


def test_bn_eval_coeffs(cuda):
    from openset_imagenet import _native as N
    import osi_testlib as T
    C = 64
    g = torch.Generator().manual_seed(0)
    rm, rv, ga, be = torch.randn(C, generator=g), torch.rand(C, generator=g) + .1, torch.randn(C, generator=g), torch.randn(C, generator=g)
    x = torch.randn(2, C, 5, 5, generator=g)
    ref = F.batch_norm(x, rm, rv, ga, be, False, 0.1, 1e-5)
    sc, sh = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    rmg, rvg, gag, beg = rm.to(cuda), rv.to(cuda), ga.to(cuda), be.to(cuda)   # keep alive: launches are asynchronous
    N.check(N.lib().osi_bn_eval_coeffs(N.ptr(rmg), N.ptr(rvg), N.ptr(gag), N.ptr(beg), 1e-5, C, N.ptr(sc), N.ptr(sh), T.S()))
    xg = T.nhwc(x).to(cuda)
    out = torch.empty_like(xg)
    N.check(N.lib().osi_bn_apply(N.ptr(xg), None, N.ptr(sc), N.ptr(sh), N.ptr(out), 50, C, 0, T.S()))
    assert torch.allclose(T.nchw(out).cpu(), ref, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("B,C,H", [(2, 64, 16), (3, 64, 15), (1, 8, 7)])
def test_maxpool(cuda, B, C, H):
    import osi_testlib as T
    from openset_imagenet import _native as N
    g = torch.Generator().manual_seed(H)
    x = F.relu(torch.randn(B, C, H, H, generator=g)).requires_grad_(True)   # post-ReLU input: many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Ho = y.shape[2]
    xg = T.nhwc(x.detach()).to(cuda)
    yg = torch.empty(B, Ho, Ho, C, device=cuda)
    idx = torch.empty(B * Ho * Ho * C, dtype=torch.uint8, device=cuda)
    N.check(N.lib().osi_maxpool3x3s2_fwd(N.ptr(xg), N.ptr(yg), N.ptr(idx), B, H, H, C, T.S()))
    assert torch.equal(T.nchw(yg).cpu(), y.detach())
    dx = torch.empty_like(xg)
    dyg = T.nhwc(dy).to(cuda)
    N.check(N.lib().osi_maxpool3x3s2_bwd(N.ptr(dyg), N.ptr(idx), N.ptr(dx), B, H, H, C, T.S()))
    assert torch.allclose(T.nchw(dx).cpu(), x.grad, atol=1e-6), "maxpool backward (first-max tie rule)"


def test_avgpool(cuda):
    import osi_testlib as T
    from openset_imagenet import _native as N
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 2048, 7, 7, generator=g).requires_grad_(True)
    y = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    dy = torch.randn(3, 2048, generator=g)
    y.backward(dy)
    xg = T.nhwc(x.detach()).to(cuda)
    yg = torch.empty(3, 2048, device=cuda)
    N.check(N.lib().osi_avgpool_fwd(N.ptr(xg), N.ptr(yg), 3, 49, 2048, T.S()))
    assert torch.allclose(yg.cpu(), y.detach(), atol=1e-6)
    dx = torch.empty_like(xg)
    dyg = dy.to(cuda)
    N.check(N.lib().osi_avgpool_bwd(N.ptr(dyg), N.ptr(dx), 3, 49, 2048, T.S()))
    assert torch.allclose(T.nchw(dx).cpu(), x.grad, atol=1e-7)


@pytest.mark.parametrize("B,K,O,bias", [(5, 2048, 116, True), (7, 30, 30, False), (3, 151, 151, False), (128, 2048, 152, True)])
def test_linear(cuda, B, K, O, bias):
    import osi_testlib as T
    from openset_imagenet import _native as N
    g = torch.Generator().manual_seed(K + O)
    x, w, b = torch.randn(B, K, generator=g), torch.randn(O, K, generator=g) / K ** .5, torch.randn(O, generator=g)
    dy = torch.randn(B, O, generator=g)

    def ref(dt):
        xx, ww, bb = (t.to(dt).clone().requires_grad_(True) for t in (x, w, b))
        y = F.linear(xx, ww, bb if bias else None)
        y.backward(dy.to(dt))
        return y.detach(), xx.grad, ww.grad, bb.grad if bias else None
    r32, r64 = ref(torch.float32), ref(torch.float64)
    xg, wg, bg, dyg = x.to(cuda), w.to(cuda), b.to(cuda), dy.to(cuda)
    y = torch.empty(B, O, device=cuda)
    N.check(N.lib().osi_linear_fwd(N.ptr(xg), N.ptr(wg), N.ptr(bg) if bias else None, N.ptr(y), B, K, O, T.S()))
    _check_vs64(y, r32[0], r64[0], "linear fwd")
    dx, dw, db = torch.empty_like(xg), torch.empty_like(wg), torch.empty_like(bg)
    N.check(N.lib().osi_linear_bwd(N.ptr(dyg), N.ptr(xg), N.ptr(wg), N.ptr(dx), 0, N.ptr(dw), N.ptr(db) if bias else None, B, K, O, T.S()))
    _check_vs64(dx, r32[1], r64[1], "linear dx")
    _check_vs64(dw, r32[2], r64[2], "linear dw")
    if bias:
        _check_vs64(db, r32[3], r64[3], "linear db")


def _run_loss(cuda, mode, z, y, w=1.0, cw=None, feats=None, xi=0.0, alpha=0.0, want_grad=True):
    from openset_imagenet import _native as N
    import osi_testlib as T
    zg, yg = z.to(cuda).contiguous(), y.to(cuda)
    loss = torch.empty((), device=cuda)
    dz = torch.full_like(zg, float("nan")) if want_grad else None
    fg = feats.to(cuda).contiguous() if feats is not None else None
    df = torch.full_like(fg, float("nan")) if fg is not None else None
    cwg = cw.to(cuda).contiguous() if cw is not None else None
    N.check(N.lib().osi_loss_fwd_bwd(mode, N.ptr(zg), N.ptr(yg), z.shape[0], z.shape[1], float(w), -1, N.ptr(cwg), N.ptr(fg),
                                     feats.shape[1] if feats is not None else 0, float(xi), float(alpha), N.ptr(loss), N.ptr(dz), N.ptr(df), T.S()))
    return loss.cpu(), None if dz is None else dz.cpu(), None if df is None else df.cpu()


def test_losses_vs_reference_golden(cuda, golden_dir):
    """HIP loss kernels against vectors produced by the reference's own losses.py / CrossEntropyLoss calls."""
    from openset_imagenet import _native as N
    G = np.load(f"{golden_dir}/losses_reference.npz")
    for fam, mode in (("eos", N.LOSS_ENTROPIC), ("sm", N.LOSS_SOFTMAX), ("gb", N.LOSS_GARBAGE)):
        for name in G[f"{fam}.names"]:
            p = f"{fam}.{name}."
            z, y = torch.from_numpy(G[p + "logits"]), torch.from_numpy(G[p + "target"])
            w = float(G[p + "w"]) if fam == "eos" else 1.0
            cw = torch.from_numpy(G[p + "class_weights"]) if fam == "gb" else None
            loss, dz, _ = _run_loss(cuda, mode, z, y, w, cw)
            ref_loss, ref_dz = float(G[p + "loss"]), torch.from_numpy(G[p + "dlogits"])
            if np.isnan(ref_loss):
                assert torch.isnan(loss), f"{p}: all-ignored batch must give NaN like torch"
                continue
            assert abs(float(loss) - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss)), f"{p} loss {float(loss)} vs {ref_loss}"
            # expf(x) carries ~|x| ulp of relative error: scale the bound with the logit range
            tol = 2e-7 + 1e-6 * float(ref_dz.abs().max()) + 2e-8 * float(z.abs().max())
            assert float((dz - ref_dz).abs().max()) <= tol, f"{p} dlogits"
            loss2, _, _ = _run_loss(cuda, mode, z, y, w, cw, want_grad=False)
            assert torch.equal(loss, loss2), "loss-only launch must give the same bits"


def test_objectosphere_vs_oracle(cuda):
    from oracle import losses_oracle as LO
    from openset_imagenet import _native as N
    g = torch.Generator().manual_seed(9)
    B, C = 16, 30
    z = torch.randn(B, C, generator=g) * 2
    y = torch.randint(-1, C, (B,), generator=g)
    f = torch.randn(B, C, generator=g) * 3
    f[0] = 0  # |f| = 0: gradient defined as 0
    for xi, alpha, w in ((10.0, 1e-2, 1.0), (3.0, 0.5, 0.5)):
        z64, f64 = z.double().requires_grad_(True), f.double().requires_grad_(True)
        J = LO.objectosphere_loss(z64, y, f64, w, xi, alpha)
        J.backward()
        loss, dz, df = _run_loss(cuda, N.LOSS_ENTROPIC, z, y, w, None, f, xi, alpha)
        assert abs(float(loss) - float(J)) <= 2e-6 * max(1, abs(float(J)))
        assert float((dz.double() - z64.grad).abs().max()) < 1e-6
        assert float((df.double() - torch.nan_to_num(f64.grad)).abs().max()) < 1e-6 * max(1.0, float(f64.grad[1:].abs().max()))


def test_softmax_kernel(cuda):
    from openset_imagenet import _native as N
    import osi_testlib as T
    z = torch.randn(37, 151) * 4
    out = torch.empty(37, 151, device=cuda)
    zg = z.to(cuda)
    N.check(N.lib().osi_softmax(N.ptr(zg), N.ptr(out), 37, 151, T.S()))
    assert torch.allclose(out.cpu(), torch.softmax(z, 1), atol=1e-7, rtol=1e-5)


@pytest.mark.parametrize("n", [4, 1000, 23759172])
def test_adam_sgd_vs_torch(cuda, n):
    from openset_imagenet import _native as N
    import osi_testlib as T
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (0.1 + i) for i in range(3)]
    for kind in ("adam", "sgd"):
        pt = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([pt], lr=1e-3) if kind == "adam" else torch.optim.SGD([pt], lr=1e-2, momentum=0.9)
        pg = p0.to(cuda).clone()
        s1, s2 = torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
        for i, gr in enumerate(grads):
            pt.grad = gr.clone()
            opt.step()
            gg = gr.to(cuda)
            if kind == "adam":
                N.check(N.lib().osi_adam_step(N.ptr(pg), N.ptr(gg), N.ptr(s1), N.ptr(s2), n, 1e-3, 0.9, 0.999, 1e-8, i + 1, 1.0, T.S()))
            else:
                N.check(N.lib().osi_sgd_step(N.ptr(pg), N.ptr(gg), N.ptr(s1), n, 1e-2, 0.9, int(i == 0), 1.0, T.S()))
        d = float((pg.cpu() - pt.detach()).abs().max())
        assert d <= 1e-6, f"{kind}: max param diff {d}"


@pytest.mark.parametrize("B,C,H,res", [(3, 64, 14, False), (2, 256, 7, True), (5, 12, 6, True), (2, 2048, 3, False)])
def test_batchnorm_relu_bitmask_forms(cuda, B, C, H, res):
    """The bitmask forms (forward writes 1 bit/element, backward consumes it) give the same bits as the activation-mask forms."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(C * 3 + H)
    M = B * H * H
    y = (torch.randn(M, C, generator=g)).to(cuda)
    resid = torch.randn(M, C, generator=g).to(cuda) if res else None
    ga, be = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    dout = torch.randn(M, C, generator=g).to(cuda)
    mean, invstd, scale, shift = (torch.empty(C, device=cuda) for _ in range(4))
    wsb = max(L.osi_bn_workspace(M, C), L.osi_bn_backward_workspace(M, C))
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_train_stats(N.ptr(y), M, C, N.ptr(ga), N.ptr(be), 1e-5, 0.1, None, None, N.ptr(mean), N.ptr(invstd), N.ptr(scale),
                                 N.ptr(shift), N.ptr(ws), wsb, T.S()))
    out1, out2 = torch.empty_like(y), torch.empty_like(y)
    mb = L.osi_bn_relu_mask_bytes(M, C)
    assert mb == ((M * C // 4 + 63) // 64) * 32
    mask = torch.zeros(mb, dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_apply(N.ptr(y), N.ptr(resid), N.ptr(scale), N.ptr(shift), N.ptr(out1), M, C, 1, T.S()))
    N.check(L.osi_bn_apply_relu_mask(N.ptr(y), N.ptr(resid), N.ptr(scale), N.ptr(shift), N.ptr(out2), N.ptr(mask), M, C, T.S()))
    assert torch.equal(out1, out2)
    # unpack the mask on the host and compare with out > 0
    words = mask.cpu().numpy().view(np.uint64).reshape(-1, 4)
    n4 = M * C // 4
    idx = np.arange(n4)
    bits = np.stack([(words[idx >> 6, c] >> (idx & 63).astype(np.uint64)) & 1 for c in range(4)], axis=1).reshape(-1).astype(bool)
    assert np.array_equal(bits, (out1.cpu().numpy().reshape(-1) > 0))
    res_a, res_b = [], []
    for which, store in ((0, res_a), (1, res_b)):
        dy, gm = torch.empty_like(y), torch.empty_like(y)
        dg, db = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
        if which == 0:
            N.check(L.osi_bn_backward(N.ptr(dout), N.ptr(out1), N.ptr(y), N.ptr(mean), N.ptr(invstd), N.ptr(ga), N.ptr(dy), N.ptr(gm),
                                      N.ptr(dg), N.ptr(db), M, C, N.ptr(ws), wsb, T.S()))
        else:
            N.check(L.osi_bn_backward_relu_mask(N.ptr(dout), N.ptr(mask), N.ptr(y), N.ptr(mean), N.ptr(invstd), N.ptr(ga), N.ptr(dy),
                                                N.ptr(gm), N.ptr(dg), N.ptr(db), M, C, N.ptr(ws), wsb, T.S()))
        torch.cuda.synchronize()
        store += [dy, gm, dg, db]
    for a, b in zip(res_a, res_b):
        assert torch.equal(a, b)


@pytest.mark.parametrize("Cin,Cout,k,stride,H,B", [(64, 64, 1, 1, 14, 3), (128, 128, 3, 2, 9, 3), (256, 512, 1, 1, 7, 5), (64, 256, 3, 1, 12, 2),
                                                    (64, 64, 1, 1, 56, 3),    # 147 row tiles: past bn_single_p
                                                    (64, 64, 1, 1, 51, 5)])   # 204 row tiles, the last one ragged: the wide form's unrolled loop
def test_conv_epilogue_batchnorm_statistics(cuda, Cin, Cout, k, stride, H, B):
    """BN statistics emitted by the conv-forward epilogue + osi_bn_finalize_stats == statistics of the conv output (fp64),
    for every tile shape (ragged last row tile included), and the conv output itself is unchanged."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    pad = 1 if k == 3 else 0
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=g) + 0.3
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    y64 = F.conv2d(x.double(), w.double(), None, stride, pad)
    mean64 = y64.mean(dim=(0, 2, 3)); var64 = y64.var(dim=(0, 2, 3), unbiased=False)
    xg, wg = T.nhwc(x).to(cuda), T.krsc(w).to(cuda)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, stride, pad)
    M = B * d.Ho * d.Wo
    ga, be = (torch.rand(Cout, generator=g) + 0.5).to(cuda), torch.randn(Cout, generator=g).to(cuda)
    nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
    for tile in [0, 4, 5, 2] + ([1, 3, 6] if Cout % 128 == 0 else []):
        ps = torch.full((nb // 4,), float("nan"), device=cuda)
        y = torch.empty(B, d.Ho, d.Wo, Cout, device=cuda)
        P, rows = ctypes.c_int(), ctypes.c_int()
        N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(xg), N.ptr(wg), N.ptr(y), tile, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), T.S()))
        if tile == 0:   # AUTO with statistics may split the ragged last round along K (another summation order): close, not equal ...
            assert float((y - T.conv_fwd(xg, wg, k, stride, pad, tile)).abs().max()) <= 1e-5 * float(y64.abs().max())
            N.check(L.osi_set_tuning(b"tail_split", 0))
            try:        # ... and with the split switched off the output is the plain kernel's, bit for bit
                y0 = torch.empty_like(y)
                N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(xg), N.ptr(wg), N.ptr(y0), tile, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), T.S()))
                assert torch.equal(y0, T.conv_fwd(xg, wg, k, stride, pad, tile))
            finally:
                N.check(L.osi_set_tuning(b"tail_split", 1))
            N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(xg), N.ptr(wg), N.ptr(y), tile, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), T.S()))
        else:
            assert torch.equal(y, T.conv_fwd(xg, wg, k, stride, pad, tile))
        # every finalisation form the partial count admits: one 256-thread launch (P <= bn_single_p), one 1024-thread launch
        # (P <= bn_wide_p), two levels (above); the knobs force each in turn
        for single_p, wide_p in ((128, 2048), (1, 2048), (1, 0)):
            N.check(L.osi_set_tuning(b"bn_single_p", single_p)); N.check(L.osi_set_tuning(b"bn_wide_p", wide_p))
            try:
                mean, invstd, scale, shift = (torch.empty(Cout, device=cuda) for _ in range(4))
                rm, rv = torch.zeros(Cout, device=cuda), torch.ones(Cout, device=cuda)
                N.check(L.osi_bn_finalize_stats(N.ptr(ps), nb, P.value, rows.value, M, Cout, N.ptr(ga), N.ptr(be), 1e-5, 0.1, N.ptr(rm),
                                                N.ptr(rv), N.ptr(mean), N.ptr(invstd), N.ptr(scale), N.ptr(shift), T.S()))
                torch.cuda.synchronize()
            finally:
                N.check(L.osi_set_tuning(b"bn_single_p", 128)); N.check(L.osi_set_tuning(b"bn_wide_p", 2048))
            form = f"tile {tile} single_p {single_p} wide_p {wide_p}"
            assert float((mean.cpu().double() - mean64).abs().max()) <= 2e-6 * float(y64.abs().max()), f"{form} mean"
            inv64 = 1 / torch.sqrt(var64 + 1e-5)
            assert float(((invstd.cpu().double() - inv64) / inv64).abs().max()) <= 2e-5, f"{form} invstd"
            assert torch.allclose(rv.cpu().double(), 0.9 + 0.1 * var64 * M / (M - 1), rtol=2e-5), form
            assert torch.allclose(scale, ga * invstd) and torch.allclose(shift, be - mean * scale, atol=1e-6), form


class _Fusion(ctypes.Structure):
    _fields_ = [("relu_mask", ctypes.c_void_p), ("y0", ctypes.c_void_p), ("mean0", ctypes.c_void_p), ("invstd0", ctypes.c_void_p),
                ("y1", ctypes.c_void_p), ("mean1", ctypes.c_void_p), ("invstd1", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("partials_bytes", ctypes.c_size_t), ("scale0", ctypes.c_void_p), ("shift0", ctypes.c_void_p),
                ("pool_idx", ctypes.c_void_p), ("pool_H", ctypes.c_int), ("pool_W", ctypes.c_int), ("addend_stride", ctypes.c_int)]   # osi_dgrad_fusion (ABI 4)


@pytest.mark.parametrize("H,B,two,tail", [(12, 3, True, 0), (14, 5, False, 1), (9, 2, True, 0)])
def test_sparse_shortcut_gradient_and_even_pixel_addend(cuda, H, B, two, tail):
    """A stride-2 1x1 shortcut reaches only the even-even pixels of its input: osi_conv_dgrad(accumulate = 2) writes just those and
    leaves the rest of dx untouched, and osi_conv_dgrad_fused with addend_stride = 2 reads the addend only there. Both together equal
    the dense pair (zero-filled shortcut gradient, dense addend) bit for bit: output, and the BatchNorm partial sums."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    Cin, Cds, C1 = 128, 256, 128         # block input channels; shortcut 1x1 s2 Cin -> Cds; conv1 1x1 s1 Cin -> C1 (4 K tiles)
    g = torch.Generator().manual_seed(H * 7 + B)
    dds = N.ConvDesc.make(B, H, H, Cin, Cds, 1, 2, 0)
    d1 = N.ConvDesc.make(B, H, H, Cin, C1, 1, 1, 0)
    M = B * H * H
    dy_ds = torch.randn(B, dds.Ho, dds.Wo, Cds, generator=g).to(cuda)
    w_ds = (torch.randn(Cds, 1, 1, Cin, generator=g) / Cds ** 0.5).to(cuda)
    dy1 = torch.randn(B, H, H, C1, generator=g).to(cuda)
    w1 = (torch.randn(C1, 1, 1, Cin, generator=g) / C1 ** 0.5).to(cuda)
    # the sparse shortcut gradient: even-even pixels written, everything else untouched
    dense = torch.full((B, H, H, Cin), float("nan"), device=cuda)
    N.check(L.osi_conv_dgrad(ctypes.byref(dds), N.ptr(dy_ds), N.ptr(w_ds), N.ptr(dense), 0, 0, T.S()))
    sparse = torch.full((B, H, H, Cin), float("nan"), device=cuda)
    N.check(L.osi_conv_dgrad(ctypes.byref(dds), N.ptr(dy_ds), N.ptr(w_ds), N.ptr(sparse), 2, 0, T.S()))
    torch.cuda.synchronize()
    assert torch.equal(sparse[:, ::2, ::2], dense[:, ::2, ::2])
    untouched = torch.ones(H, H, dtype=torch.bool); untouched[::2, ::2] = False
    assert bool(torch.isnan(sparse[:, untouched.to(cuda)]).all()) and float(dense[:, untouched.to(cuda)].abs().max()) == 0.0
    assert L.osi_conv_dgrad(ctypes.byref(dds), N.ptr(dy_ds), N.ptr(w_ds), N.ptr(sparse), 3, 0, T.S()) != 0
    # the consumer: conv1's input gradient completes the sum in place, masks it and emits the BatchNorm reductions
    ys = [(torch.randn(M, Cin, generator=g) * 2 + 1).to(cuda) for _ in range(2 if two else 1)]
    stats = [[(torch.randn(Cin, generator=g) + 1).to(cuda), (torch.rand(Cin, generator=g) + 0.5).to(cuda)] for _ in ys]   # mean, invstd
    mask = torch.randint(0, 256, (L.osi_bn_relu_mask_bytes(M, Cin),), generator=g, dtype=torch.uint8).to(cuda)
    outs = []
    N.check(L.osi_set_tuning(b"tail_split", tail))
    if tail:    # a plan that splits this small launch: 32 tiles on "24 CUs" = one full round + 8 remainder tiles cut in 3
        N.check(L.osi_set_tuning(b"tail_mint", 1)); N.check(L.osi_set_tuning(b"tail_smax", 32)); N.check(L.osi_set_tuning(b"tail_cus", 24))
    try:
        pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d1))     # includes the slab of the plan in force
        for buf, stride in ((dense.clone(), 1), (sparse.clone(), 2)):
            parts = torch.full((pb // 4,), float("nan"), device=cuda)
            f = _Fusion(mask.data_ptr(), ys[0].data_ptr(), stats[0][0].data_ptr(), stats[0][1].data_ptr(),
                        ys[1].data_ptr() if two else None, stats[1][0].data_ptr() if two else None, stats[1][1].data_ptr() if two else None,
                        parts.data_ptr(), pb, None, None, None, 0, 0, stride)
            P = ctypes.c_int()
            N.check(L.osi_conv_dgrad_fused(ctypes.byref(d1), N.ptr(dy1), N.ptr(w1), N.ptr(buf), N.ptr(buf), ctypes.byref(f), 0, ctypes.byref(P), T.S()))
            torch.cuda.synchronize()
            outs.append((buf, None, parts.clone()))
    finally:
        N.check(L.osi_set_tuning(b"tail_split", 1))
        if tail:
            N.check(L.osi_set_tuning(b"tail_mint", 16)); N.check(L.osi_set_tuning(b"tail_smax", 8)); N.check(L.osi_set_tuning(b"tail_cus", 0))
    assert not bool(torch.isnan(outs[1][0]).any())
    assert torch.equal(outs[0][0], outs[1][0]), "even-pixel addend must give the dense result bit for bit"
    k = (3 if two else 2) * P.value * Cin      # the partial sums in use (sum g, sum g*xhat0 [, sum g*xhat1]); the rest is workspace
    assert torch.equal(outs[0][2][:2 * P.value * Cin], outs[1][2][:2 * P.value * Cin])
    if two:
        a, b = outs[0][2][2 * P.value * Cin:k], outs[1][2][2 * P.value * Cin:k]
        assert torch.equal(a, b) and not bool(torch.isnan(a).any())
    # stride-2 addends are refused where they make no sense: a strided convolution, no addend
    f = _Fusion(mask.data_ptr(), ys[0].data_ptr(), stats[0][0].data_ptr(), stats[0][1].data_ptr(), None, None, None, parts.data_ptr(), pb,
                None, None, None, 0, 0, 2)
    assert L.osi_conv_dgrad_fused(ctypes.byref(d1), N.ptr(dy1), N.ptr(w1), N.ptr(buf), None, ctypes.byref(f), 0, ctypes.byref(P), T.S()) != 0


@pytest.mark.parametrize("Cin,Cout,k,stride,H,B,two", [(64, 64, 1, 1, 14, 3, False), (128, 64, 3, 1, 9, 3, True), (256, 128, 3, 2, 9, 2, False),
                                                     (64, 128, 1, 2, 8, 3, True), (512, 64, 1, 1, 7, 5, False)])
def test_dgrad_fused_epilogue_vs_unfused(cuda, Cin, Cout, k, stride, H, B, two):
    """dgrad with the fused epilogue (addend + ReLU bitmask + BatchNorm reductions) followed by osi_bn_backward_fused equals
    plain dgrad + osi_bn_backward_relu_mask on the same data: masked gradient bit for bit, BN outputs to fp32 summation noise."""
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    pad = 1 if k == 3 else 0
    g = torch.Generator().manual_seed(Cin + 3 * Cout + H)
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, stride, pad)
    M = B * H * H
    dy = torch.randn(B, d.Ho, d.Wo, Cout, generator=g).to(cuda)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cout * k * k) ** 0.5).to(cuda)
    addend = torch.randn(B, H, H, Cin, generator=g).to(cuda)
    # the "previous layer": y0 (and y1) pre-BN tensors, their batch statistics, the ReLU bitmask of their BN output
    ys = [(torch.randn(M, Cin, generator=g) * 2 + 1).to(cuda) for _ in range(2 if two else 1)]
    ga = [(torch.rand(Cin, generator=g) + 0.5).to(cuda) for _ in ys]
    be = [torch.randn(Cin, generator=g).to(cuda) for _ in ys]
    wsb = max(L.osi_bn_workspace(M, Cin), L.osi_bn_backward_workspace(M, Cin))
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    stats = []
    for yv, gv, bv in zip(ys, ga, be):
        st = [torch.empty(Cin, device=cuda) for _ in range(4)]
        N.check(L.osi_bn_train_stats(N.ptr(yv), M, Cin, N.ptr(gv), N.ptr(bv), 1e-5, 0.1, None, None, *[N.ptr(t) for t in st], N.ptr(ws), wsb, T.S()))
        stats.append(st)
    out = torch.empty(M, Cin, device=cuda)
    mask = torch.zeros(L.osi_bn_relu_mask_bytes(M, Cin), dtype=torch.uint8, device=cuda)
    N.check(L.osi_bn_apply_relu_mask(N.ptr(ys[0]), None, N.ptr(stats[0][2]), N.ptr(stats[0][3]), N.ptr(out), N.ptr(mask), M, Cin, T.S()))
    # reference: plain dgrad (+addend), then the unfused masked BatchNorm backward per consumer
    dx_raw = addend.clone()
    N.check(L.osi_conv_dgrad(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx_raw), 1, 0, T.S()))
    ref = []
    for yv, gv, st in zip(ys, ga, stats):
        dyo, gm, dg, db = torch.empty(M, Cin, device=cuda), torch.empty(M, Cin, device=cuda), torch.empty(Cin, device=cuda), torch.empty(Cin, device=cuda)
        N.check(L.osi_bn_backward_relu_mask(N.ptr(dx_raw), N.ptr(mask), N.ptr(yv), N.ptr(st[0]), N.ptr(st[1]), N.ptr(gv), N.ptr(dyo), N.ptr(gm),
                                            N.ptr(dg), N.ptr(db), M, Cin, N.ptr(ws), wsb, T.S()))
        ref.append((dyo, gm, dg, db))
    # fused
    pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d))
    parts = torch.full((pb // 4,), float("nan"), device=cuda)
    f = _Fusion(mask.data_ptr(), ys[0].data_ptr(), stats[0][0].data_ptr(), stats[0][1].data_ptr(),
                ys[1].data_ptr() if two else None, stats[1][0].data_ptr() if two else None, stats[1][1].data_ptr() if two else None,
                parts.data_ptr(), pb, None, None)
    gbuf = torch.full((B, H, H, Cin), float("nan"), device=cuda)
    P = ctypes.c_int()
    # fused == unfused bit for bit is a statement about the EPILOGUE: the K-split tail (another summation order of the same products,
    # tests/test_tail_split_gpu.py) is switched off for this launch
    N.check(L.osi_set_tuning(b"tail_split", 0))
    try:
        N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(gbuf), N.ptr(addend), ctypes.byref(f), 0, ctypes.byref(P), T.S()))
    finally:
        N.check(L.osi_set_tuning(b"tail_split", 1))
    assert torch.equal(gbuf.view(M, Cin), ref[0][1]), "masked gradient must match bit for bit"
    pm = parts.view(3, -1)[:, :P.value * Cin].reshape(3, P.value, Cin) if False else None
    for j, (yv, gv, st) in enumerate(zip(ys, ga, stats)):
        dyo, dg, db = torch.empty(M, Cin, device=cuda), torch.empty(Cin, device=cuda), torch.empty(Cin, device=cuda)
        psum_g = parts.data_ptr()
        psum_gx = parts.data_ptr() + 4 * (1 + j) * P.value * Cin
        for wide_p in (2048, 0):        # the partial sums merged by one 1024-thread launch, and by the two-level pair
            N.check(L.osi_set_tuning(b"bn_wide_p", wide_p))
            try:
                dyo.fill_(float("nan")); dg.fill_(float("nan")); db.fill_(float("nan"))
                N.check(L.osi_bn_backward_fused(N.ptr(gbuf), N.ptr(yv), N.ptr(st[0]), N.ptr(st[1]), N.ptr(gv), psum_g, psum_gx, P.value,
                                                N.ptr(dyo), N.ptr(dg), N.ptr(db), M, Cin, N.ptr(ws), wsb, T.S()))
                torch.cuda.synchronize()
            finally:
                N.check(L.osi_set_tuning(b"bn_wide_p", 2048))
            rdy, _, rdg, rdb = ref[j]
            scale = float(rdy.abs().max()) + 1e-30
            assert float((dyo - rdy).abs().max()) <= 2e-5 * scale, f"dy consumer {j} wide_p {wide_p}"
            assert float((dg - rdg).abs().max()) <= 2e-5 * (float(rdg.abs().max()) + 1e-30), f"dgamma consumer {j} wide_p {wide_p}"
            assert float((db - rdb).abs().max()) <= 2e-5 * (float(rdb.abs().max()) + 1e-30), f"dbeta consumer {j} wide_p {wide_p}"


@pytest.mark.parametrize("P,C", [(392, 256), (1568, 128), (6272, 64), (98, 512)])
def test_bn_merge_with_large_mean_and_outlier_first_tile(cuda, P, C):
    """The one-pass merges of the conv-epilogue partials (k_bn_stats_final_wide, k_bn_stats_group) subtract S1^2 / M from S2: that
    only works while the pivot of the shifted sums is near the batch mean. Hard case: |mean| / sigma = 1e4 and a FIRST tile (the
    top-left corner of image 0: border pixels) 50 sigma away from everything else. Every finalisation form (one 256-thread
    launch, one 1024-thread launch, two levels) against fp64."""
    from openset_imagenet import _native as N
    import osi_testlib as T
    L = N.lib()
    g = torch.Generator().manual_seed(P + C)
    rows, M = 64, P * 64 - 17
    mu, sigma = 100.0, 0.01
    y = (mu + sigma * torch.randn(M, C, generator=g, dtype=torch.float64))
    y[:rows] += 50 * sigma
    y = y.float().double()                                   # the values the conv would have produced (fp32)
    pad = torch.cat([y, torch.full((P * rows - M, C), float("nan"), dtype=torch.float64)])
    tiles = pad.view(P, rows, C)
    pm = torch.nanmean(tiles, dim=1)
    pq = torch.nansum((tiles - pm[:, None]) ** 2, dim=1)
    rmean = y.mean(0); rM2 = ((y - rmean) ** 2).sum(0)
    gamma, beta = torch.ones(C, device=cuda), torch.zeros(C, device=cuda)
    for single_p, wide_p in ((128, 2048), (1, 2048), (1, 0)):
        N.check(L.osi_set_tuning(b"bn_single_p", single_p)); N.check(L.osi_set_tuning(b"bn_wide_p", wide_p))
        try:
            nb = max(L.osi_bn_workspace(M, C), 2 * P * C * 4 + 2 * 32 * C * 4 + 1024)
            ps = torch.zeros(nb // 4, device=cuda)
            ps[:P * C] = pm.float().flatten().to(cuda); ps[P * C:2 * P * C] = pq.float().flatten().to(cuda)
            o = [torch.empty(C, device=cuda) for _ in range(4)]
            N.check(L.osi_bn_finalize_stats(N.ptr(ps), nb, P, rows, M, C, N.ptr(gamma), N.ptr(beta), 1e-5, 0.1, None, None,
                                            *[N.ptr(t) for t in o], T.S()), "osi_bn_finalize_stats")
        finally:
            N.check(L.osi_set_tuning(b"bn_single_p", 128)); N.check(L.osi_set_tuning(b"bn_wide_p", 2048))
        inv = 1 / torch.sqrt(rM2 / M + 1e-5)
        assert float((o[0].cpu().double() - rmean).abs().max()) <= 5e-5, (single_p, wide_p)           # |mean| = 100: 6 ulp of fp32
        assert float((o[1].cpu().double() / inv - 1).abs().max()) <= 2e-4, (single_p, wide_p, float((o[1].cpu().double() / inv - 1).abs().max()))
