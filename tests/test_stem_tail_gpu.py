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


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 96, 128), (2, 224, 224), (5, 32, 96)])
def test_stem_direct_forward(cuda, B, H, W):
    """The direct form of the stem convolution (k_stem_fwd_direct: 8 x 16 output tiles, input patch + whole weight matrix in LDS,
    K = 148 instead of 224) against torch conv2d in fp64, against the implicit-GEMM form it replaces (same products, another
    summation order), with its per-tile BatchNorm partials finalised by osi_bn_finalize_stats; bitwise reproducible."""
    import ctypes
    import torch.nn.functional as F
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(B + H + W)
    x = torch.rand(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    y64 = F.conv2d(x.double(), w.double(), None, 2, 3)
    x4 = torch.zeros(B, H, W, 4, device=cuda)
    x4[..., :3] = x.permute(0, 2, 3, 1).to(cuda)
    wp = torch.empty(64, 224, device=cuda)
    N.check(L.osi_stem_weight_pack(N.ptr(T.krsc(w).to(cuda)), N.ptr(wp), 64, T.S()))
    d = N.ConvDesc.make(B, H, W, 4, 64, 7, 2, 3)
    M = B * d.Ho * d.Wo
    nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))

    def run():
        ps = torch.full((nb // 4,), float("nan"), device=cuda)
        y = torch.full((B, d.Ho, d.Wo, 64), float("nan"), device=cuda)
        P, rows = ctypes.c_int(), ctypes.c_int()
        N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x4), N.ptr(wp), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), T.S()))
        return y, ps, P.value, rows.value
    y1, ps1, P1, rows1 = run()
    y2, ps2, _, _ = run()
    N.check(L.osi_set_tuning(b"stem_direct", 0))
    try:
        y0, _, P0, rows0 = run()
    finally:
        N.check(L.osi_set_tuning(b"stem_direct", 1))
    assert (P1, rows1) == (M // 128, 128) and rows0 == 128
    ref = y64.permute(0, 2, 3, 1)
    tol = (2e-6 + 6e-8 * 147 ** 0.5) * float(ref.abs().max())
    assert float((y1.cpu().double() - ref).abs().max()) <= tol and float((y0.cpu().double() - ref).abs().max()) <= tol
    assert torch.equal(y1, y2) and torch.equal(ps1[:2 * P1 * 64], ps2[:2 * P1 * 64])
    assert float((y1 - y0).abs().max()) <= 1e-5 * float(ref.abs().max()) and not torch.equal(y1, y0)
    # eval-mode entry point (no statistics) runs the same kernel: same bits
    y3 = torch.empty_like(y1)
    N.check(L.osi_conv_fwd(ctypes.byref(d), N.ptr(x4), N.ptr(wp), N.ptr(y3), 0, T.S()))
    assert torch.equal(y3, y1)
    # the per-tile partials finalise to the batch statistics of this output
    ga, be = torch.rand(64, device=cuda) + 0.5, torch.randn(64, device=cuda)
    mean, invstd, scale, shift = (torch.empty(64, device=cuda) for _ in range(4))
    N.check(L.osi_bn_finalize_stats(N.ptr(ps1), nb, P1, rows1, M, 64, N.ptr(ga), N.ptr(be), 1e-5, 0.1, None, None,
                                    N.ptr(mean), N.ptr(invstd), N.ptr(scale), N.ptr(shift), T.S()))
    yv = y1.double().view(M, 64)
    assert float((mean.double() - yv.mean(0)).abs().max()) <= 2e-6 * float(yv.abs().max())
    inv64 = 1 / torch.sqrt(yv.var(0, unbiased=False) + 1e-5)
    assert float(((invstd.double() - inv64) / inv64).abs().max()) <= 2e-5


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 96, 128), (4, 224, 224), (5, 32, 96)])
def test_stem_direct_weight_gradient(cuda, B, H, W):
    """Direct stem weight gradient (k_stem_wgrad_direct: the whole 64 x 147 gradient in the registers of a five-wave workgroup,
    dY rows + input patch staged per 8 x 16 tile, one partial per workgroup) against torch conv2d_weight in fp64 and against the
    implicit-GEMM form + unpack it replaces; written in the parameter layout; bitwise reproducible; refused for ragged geometries."""
    import ctypes
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(7 * B + H + W)
    x = torch.rand(B, 3, H, W, generator=g)
    d = N.ConvDesc.make(B, H, W, 4, 64, 7, 2, 3)
    dy = torch.randn(B, 64, d.Ho, d.Wo, generator=g)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 7, 7), dy.double(), 2, 3)
    x4 = torch.zeros(B, H, W, 4, device=cuda)
    x4[..., :3] = x.permute(0, 2, 3, 1).to(cuda)
    dyg = T.nhwc(dy).to(cuda)
    nb = L.osi_stem_wgrad_direct_workspace(ctypes.byref(d))
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
    outs = []
    for _ in range(2):
        dw = torch.full((64, 7, 7, 3), float("nan"), device=cuda)
        N.check(L.osi_stem_wgrad_direct(ctypes.byref(d), N.ptr(dyg), N.ptr(x4), N.ptr(dw), N.ptr(ws), nb, T.S()))
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    got = T.oihw(outs[0]).cpu().double()
    K = B * d.Ho * d.Wo
    scale = float(dw64.abs().max())
    assert float((got - dw64).abs().max()) <= (2e-6 + 6e-8 * K ** 0.5) * scale + 1e-6
    gp = T.conv_wgrad(dyg, x4, 7, 2, 3)                                    # implicit-GEMM form, packed [64][224]
    gk = torch.empty(64, 7, 7, 3, device=cuda)
    N.check(L.osi_stem_grad_unpack(N.ptr(gp), N.ptr(gk), 64, T.S()))
    assert float((gk - outs[0]).abs().max()) <= 2e-5 * scale
    ragged = N.ConvDesc.make(B, 75, 91, 4, 64, 7, 2, 3)
    assert L.osi_stem_wgrad_direct_workspace(ctypes.byref(ragged)) == 0
    assert L.osi_stem_wgrad_direct(ctypes.byref(ragged), N.ptr(dyg), N.ptr(x4), N.ptr(outs[0]), N.ptr(ws), nb, T.S()) == -1


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 96, 128), (4, 224, 224)])
def test_stem_weight_gradient_with_fused_tail(cuda, B, H, W):
    """conv1's weight gradient with the stem tail (bn1 -> ReLU -> max-pool) differentiated inside its operand loader
    (osi_stem_wgrad_fused) against (a) the fp64 evaluation of the same mathematics on the SAME decisions (max-pool scatter through
    the stored arg-max bytes, BatchNorm backward, conv2d_weight), and (b) the route it replaces (osi_bn_relu_maxpool_bwd writing the
    112x112x64 gradient + osi_stem_wgrad_direct reading it back)."""
    import ctypes
    import torch.nn.functional as F
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(31 * B + H)
    x = (torch.rand(B, 3, H, W, generator=g) * 0.4 + torch.tensor([0.1, 0.3, 0.6]).view(1, 3, 1, 1))
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    d = N.ConvDesc.make(B, H, W, 4, 64, 7, 2, 3)
    Ho, Wo = d.Ho, d.Wo
    Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    M = B * Ho * Wo
    x4 = torch.zeros(B, H, W, 4, device=cuda)
    x4[..., :3] = x.permute(0, 2, 3, 1).to(cuda)
    y = T.nhwc(F.conv2d(x, w, None, 2, 3)).to(cuda).contiguous()                      # [B,Ho,Wo,64]
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(cuda), (torch.randn(64, generator=g) * 0.3).to(cuda)
    yv = y.view(M, 64)
    mean = yv.mean(0)
    invstd = 1 / torch.sqrt(yv.var(0, unbiased=False) + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    pooled = torch.empty(B, Hp, Wp, 64, device=cuda)
    idx = torch.zeros(B * Hp * Wp * 16, dtype=torch.int32, device=cuda)
    N.check(L.osi_bn_relu_maxpool_fwd(N.ptr(y), N.ptr(scale), N.ptr(shift), N.ptr(pooled), N.ptr(idx), B, Ho, Wo, 64, T.S()))
    gpool = torch.randn(B, Hp, Wp, 64, generator=g).to(cuda)
    # ---- route it replaces: materialised dy, then the direct weight gradient
    wsb = L.osi_bn_backward_workspace(M, 64)
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    dy = torch.empty(B, Ho, Wo, 64, device=cuda)
    dg0, db0 = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
    N.check(L.osi_bn_relu_maxpool_bwd(N.ptr(gpool), N.ptr(idx), N.ptr(y), N.ptr(mean), N.ptr(invstd), N.ptr(gamma), N.ptr(dy), N.ptr(dg0), N.ptr(db0),
                                      B, Ho, Wo, 64, N.ptr(ws), wsb, T.S()))
    nbd = L.osi_stem_wgrad_direct_workspace(ctypes.byref(d))
    wsd = torch.empty(nbd, dtype=torch.uint8, device=cuda)
    dw_direct = torch.empty(64, 7, 7, 3, device=cuda)
    N.check(L.osi_stem_wgrad_direct(ctypes.byref(d), N.ptr(dy), N.ptr(x4), N.ptr(dw_direct), N.ptr(wsd), nbd, T.S()))
    # ---- fused: dY built in the weight gradient's operand loader from gpool, the arg-max bytes and y
    nbf = L.osi_stem_wgrad_fused_workspace(ctypes.byref(d))
    wsf = torch.empty(nbf, dtype=torch.uint8, device=cuda)
    fused = []
    for _ in range(2):
        dwf = torch.full((64, 7, 7, 3), float("nan"), device=cuda)
        N.check(L.osi_stem_wgrad_fused(ctypes.byref(d), N.ptr(gpool), N.ptr(idx), N.ptr(y), N.ptr(x4), N.ptr(gamma), N.ptr(mean), N.ptr(invstd),
                                       N.ptr(dg0), N.ptr(db0), N.ptr(dwf), N.ptr(wsf), nbf, T.S()))
        fused.append(dwf)
    assert torch.equal(fused[0], fused[1])
    # the reductions-only call (dy = NULL) produces the same dgamma / dbeta as the full BatchNorm backward
    dg, db = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
    N.check(L.osi_bn_relu_maxpool_bwd(N.ptr(gpool), N.ptr(idx), N.ptr(y), N.ptr(mean), N.ptr(invstd), N.ptr(gamma), None, N.ptr(dg), N.ptr(db),
                                      B, Ho, Wo, 64, N.ptr(ws), wsb, T.S()))
    assert torch.equal(dg, dg0) and torch.equal(db, db0)
    # ---- fp64 evaluation of the same mathematics on the same decisions
    ib = idx.view(B, Hp, Wp, 16).cpu().numpy().view("uint8").reshape(B, Hp, Wp, 64)        # one byte per channel
    ib = torch.from_numpy(ib.astype("int64"))
    gp = gpool.cpu().double()
    g64 = torch.zeros(B, Ho, Wo, 64, dtype=torch.float64)
    gate = (ib >= 128)
    tap = ib % 128
    for r in range(3):
        for s_ in range(3):
            sel = gate & (tap == r * 3 + s_)
            hh = (torch.arange(Hp) * 2 - 1 + r).clamp(0, Ho - 1)
            ww = (torch.arange(Wp) * 2 - 1 + s_).clamp(0, Wo - 1)
            contrib = torch.where(sel, gp, torch.zeros_like(gp))
            g64.index_put_((torch.arange(B).view(B, 1, 1), hh.view(1, Hp, 1), ww.view(1, 1, Wp)), contrib, accumulate=True)
    y64, mu64, is64 = y.cpu().double(), mean.cpu().double(), invstd.cpu().double()
    xh = (y64 - mu64) * is64
    c1, c2 = g64.view(M, 64).mean(0), (g64 * xh).view(M, 64).mean(0)
    dy64 = (g64 - c1 - xh * c2) * (gamma.cpu().double() * is64)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 7, 7), dy64.permute(0, 3, 1, 2), 2, 3)
    old, fus = T.oihw(dw_direct).cpu().double(), T.oihw(fused[0]).cpu().double()
    rel = lambda a: float((a - dw64).norm() / dw64.norm())
    print(f"B{B} {H}x{W}: rel-L2 vs fp64: fused loader {rel(fus):.2e}, materialised route {rel(old):.2e}")
    assert rel(old) <= 5e-5 and rel(fus) <= 5e-5
    # the fused loader evaluates the expressions of k_bn_bwd_apply<3> on the same operands (up to fma contraction)
    assert float((fus - old).abs().max()) <= 2e-5 * float(dw64.abs().max())
    assert float((fus - dw64).abs().max()) <= 2e-4 * float(dw64.abs().max())


@pytest.mark.parametrize("B,Hs,Ws,Cout", [(2, 16, 16, 64), (3, 24, 40, 256), (4, 112, 112, 64)])
def test_pool_mode_dgrad_emits_the_stem_batchnorm_reductions(cuda, B, Hs, Ws, Cout):
    """Pool mode of osi_conv_dgrad_fused (the input gradient of layer1.0.conv1, whose input is the stem's max-pooled activation): dx is
    the plain input gradient + addend, and the per-row-tile partials finished by osi_bn_backward_reduce are bn1's dgamma / dbeta —
    equal (to fp32 summation noise) to the reductions osi_bn_relu_maxpool_bwd computes by scanning the 112 x 112 tensor."""
    import ctypes
    import osi_testlib as T
    from openset_imagenet import _native as N
    L = N.lib()

    class Fusion(ctypes.Structure):
        _fields_ = [("relu_mask", ctypes.c_void_p), ("y0", ctypes.c_void_p), ("mean0", ctypes.c_void_p), ("invstd0", ctypes.c_void_p),
                    ("y1", ctypes.c_void_p), ("mean1", ctypes.c_void_p), ("invstd1", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                    ("partials_bytes", ctypes.c_size_t), ("scale0", ctypes.c_void_p), ("shift0", ctypes.c_void_p),
                    ("pool_idx", ctypes.c_void_p), ("pool_H", ctypes.c_int), ("pool_W", ctypes.c_int), ("addend_stride", ctypes.c_int)]
    g = torch.Generator().manual_seed(B + Hs + Ws + Cout)
    C = 64
    Hp, Wp = (Hs + 2 - 3) // 2 + 1, (Ws + 2 - 3) // 2 + 1
    Ms = B * Hs * Ws
    y = (torch.randn(B, Hs, Ws, C, generator=g) * 1.5 + 0.2).to(cuda)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(cuda), (torch.randn(C, generator=g) * 0.3).to(cuda)
    mean = y.view(Ms, C).mean(0)
    invstd = 1 / torch.sqrt(y.view(Ms, C).var(0, unbiased=False) + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    pooled = torch.empty(B, Hp, Wp, C, device=cuda)
    idx = torch.zeros(B * Hp * Wp * (C // 4), dtype=torch.int32, device=cuda)
    N.check(L.osi_bn_relu_maxpool_fwd(N.ptr(y), N.ptr(scale), N.ptr(shift), N.ptr(pooled), N.ptr(idx), B, Hs, Ws, C, T.S()))
    # the conv behind the pool: 1x1, C -> Cout; its input gradient (+ an addend: the shortcut branch's gradient) is dJ/dpooled
    d = N.ConvDesc.make(B, Hp, Wp, C, Cout, 1, 1, 0)
    dy = torch.randn(B, Hp, Wp, Cout, generator=g).to(cuda)
    w = (torch.randn(Cout, 1, 1, C, generator=g) / Cout ** 0.5).to(cuda)
    addend = torch.randn(B, Hp, Wp, C, generator=g).to(cuda)
    ref_dx = addend.clone()
    N.check(L.osi_conv_dgrad(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(ref_dx), 1, 0, T.S()))
    pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d))
    parts = torch.full((pb // 4,), float("nan"), device=cuda)
    f = Fusion(None, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), None, None, None, parts.data_ptr(), pb, None, None, idx.data_ptr(), Hs, Ws)
    dx = addend.clone()
    P = ctypes.c_int()
    N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), N.ptr(dx), ctypes.byref(f), 0, ctypes.byref(P), T.S()))
    assert torch.equal(dx, ref_dx), "pool mode leaves the input gradient itself untouched (same kernel, same bits)"
    wsb = L.osi_bn_backward_workspace(Ms, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    dg, db = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    N.check(L.osi_bn_backward_reduce(parts.data_ptr(), parts.data_ptr() + 4 * P.value * C, P.value, N.ptr(dg), N.ptr(db), Ms, C, N.ptr(ws), wsb, T.S()))
    dg0, db0 = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    N.check(L.osi_bn_relu_maxpool_bwd(N.ptr(dx), N.ptr(idx), N.ptr(y), N.ptr(mean), N.ptr(invstd), N.ptr(gamma), None, N.ptr(dg0), N.ptr(db0),
                                      B, Hs, Ws, C, N.ptr(ws), wsb, T.S()))
    # fp64 truth through the arg-max bytes
    ib = torch.from_numpy(idx.view(B, Hp, Wp, C // 4).cpu().numpy().view("uint8").reshape(B, Hp, Wp, C).astype("int64"))
    gate, tap = ib >= 128, ib % 128
    hh = (torch.arange(Hp).view(1, Hp, 1, 1) * 2 - 1 + tap // 3).clamp(0, Hs - 1)
    ww = (torch.arange(Wp).view(1, 1, Wp, 1) * 2 - 1 + tap % 3).clamp(0, Ws - 1)
    bidx = torch.arange(B).view(B, 1, 1, 1).expand_as(ib)
    cidx = torch.arange(C).view(1, 1, 1, C).expand_as(ib)
    xh = ((y.cpu().double() - mean.cpu().double()) * invstd.cpu().double())[bidx, hh, ww, cidx]
    gq = torch.where(gate, dx.cpu().double(), torch.zeros(1, dtype=torch.float64))
    db64, dg64 = gq.sum(dim=(0, 1, 2)), (gq * xh).sum(dim=(0, 1, 2))
    sc = float(dg64.abs().max()) + float(db64.abs().max())
    assert float((dg.cpu().double() - dg64).abs().max()) <= 2e-5 * sc and float((db.cpu().double() - db64).abs().max()) <= 2e-5 * sc
    assert float((dg0.cpu().double() - dg64).abs().max()) <= 2e-5 * sc and float((db0.cpu().double() - db64).abs().max()) <= 2e-5 * sc
