"""Whole-network backward, tight: all 162 gradients of the HIP path against the fp64 oracle run UNDER THE HIP PATH'S OWN ReLU /
arg-max decisions (reference side: j.backward(), openset_imagenet/train.py:132-138).

Why: ResNet-50's gradient is a piecewise-smooth function whose pieces are selected by 49 ReLUs and one max-pool arg-max. Two
correct fp32 implementations disagree on a few dozen of the ~1e7 decisions (pre-activations within rounding of zero), each flip is
an O(1) change, and torch-CPU fp32 itself sits 1.4e-2 .. 2e-2 (relative L2, every tensor) from the fp64 run because of it. With the
decisions pinned that figure drops to 4.7e-5 median / 1.3e-4 max (tests/test_oracle.py::test_gate_pinned_fp32_vs_fp64_anchor), so
the bar here is 5e-4 per tensor instead of the ~1e-1 of the free-running comparison: scratch-buffer recycling, fork / join
events, the gate recomputed from the pre-BN tensor, every fused epilogue — one wrong wire anywhere moves a tensor by O(1).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRAD_TOL = 5e-4      # relative L2 per tensor, HIP fp32 vs fp64 oracle under the same decisions
LOGIT_TOL = 1e-4


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("tag,B,HW,C,seed,loss", [("b8_96_c30", 8, 96, 30, 11, "entropic"), ("b16_64_c116", 16, 64, 116, 5, "entropic"),
                                                  ("b6_75x91_c152_garbage", 6, (75, 91), 152, 21, "garbage"),
                                                  ("b4_224_c30", 4, 224, 30, 31, "entropic")])   # the benchmark's geometry: every layer's real H x W
def test_all_gradients_vs_fp64_oracle_under_the_hip_gates(cuda, tag, B, HW, C, seed, loss):
    from openset_imagenet import ResNet50, EntropicOpensetLoss, GarbageLoss
    from oracle import resnet50_oracle as R, losses_oracle as L
    from osi_testlib import hip_gates
    H, W = (HW, HW) if isinstance(HW, int) else HW
    gen = torch.Generator().manual_seed(seed)
    sd = R.randomize_bn(R.init_state(C, C, False, generator=gen), generator=gen)   # non-trivial gamma / beta: the affine part counts
    model = ResNet50(C, C, False)
    model.load_state_dict(sd)
    model = model.to(cuda).train()
    x = torch.rand(B, 3, H, W, generator=gen)
    if loss == "garbage":
        y = torch.randint(0, C, (B,), generator=gen)
        cw = 0.5 + torch.rand(C, generator=gen)
        hip_loss, ref_fn = GarbageLoss(cw), (lambda lg, t, f: L.garbage_loss(lg, t, cw.to(lg.dtype)))
    else:
        y = torch.randint(-1, C, (B,), generator=gen)
        hip_loss, ref_fn = EntropicOpensetLoss(C, 1.0), (lambda lg, t, f: L.entropic_openset_loss(lg, t, 1.0))
    logits, feats = model(x.to(cuda))
    j = hip_loss(logits, y.to(cuda))
    j.backward()
    torch.cuda.synchronize()
    gates = hip_gates(model)
    assert len(gates["relu"]) == 49 and gates["pool_idx"].shape == gates["relu"][0].shape

    def sd64():
        return {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    free = {}
    r_free = R.forward_backward(sd64(), x.double(), y, ref_fn, record_gates=free)
    r_pin = R.forward_backward(sd64(), x.double(), y, ref_fn, gates=gates)

    # (d) the HIP path's decisions are the fp64 run's decisions except for a handful of pre-activations within rounding of zero
    flips, pool_flips, total = R.gate_disagreements(gates, free)
    print(f"{tag}: {flips} of {total} ReLU decisions and {pool_flips} arg-max positions differ from the free fp64 run")
    assert flips <= 2e-5 * total + 20 and pool_flips <= 2e-5 * gates["pool_idx"].numel() + 5
    # the stem gate is consistent with the pooled values the forward produced
    assert float((logits.detach().cpu().double() - r_pin[0]).abs().max()) <= LOGIT_TOL
    assert float((logits.detach().cpu().double() - r_free[0]).abs().max()) <= LOGIT_TOL
    assert abs(float(j) - float(r_pin[2])) <= 1e-5 * max(1.0, abs(float(r_pin[2])))

    named = dict(model.named_parameters())
    errs = {}
    for k in R.param_keys(sd):
        g = named[k].grad.detach().cpu()
        assert g.shape == r_pin[3][k].shape and torch.isfinite(g).all(), k
        errs[k] = _rel(g, r_pin[3][k])
    worst = max(errs, key=errs.get)
    unpinned = np.median([_rel(named[k].grad.detach().cpu(), r_free[3][k]) for k in errs])
    print(f"{tag}: gradient rel-L2 vs fp64 under the HIP gates: median {np.median(list(errs.values())):.2e} max {errs[worst]:.2e} ({worst}); "
          f"against the free-running fp64 oracle the median is {unpinned:.2e}")
    assert len(errs) == 162
    for k, e in errs.items():
        assert e <= GRAD_TOL, f"grad {k}: rel-L2 {e:.2e} > {GRAD_TOL:.0e} under pinned gates"
    assert np.median(list(errs.values())) <= 1.5e-4


def test_all_gradients_at_the_benchmarked_batch(cuda):
    """The same check on the BENCHMARK's own configuration: Protocol 2 (C = 30, entropic open-set loss), B = 128 at 224 x 224 — the
    executor's production launch plans, scratch-buffer recycling and side-stream overlap at full size. One fp64 oracle forward + backward
    under the HIP path's decisions (~1 minute and ~40 GB on the box's CPUs; the free-running second pass of the small cases is skipped)."""
    import time
    from openset_imagenet import ResNet50, EntropicOpensetLoss
    from oracle import resnet50_oracle as R, losses_oracle as L
    from osi_testlib import hip_gates
    C, B = 30, 128
    gen = torch.Generator().manual_seed(77)
    sd = R.randomize_bn(R.init_state(C, C, False, generator=gen), generator=gen)
    model = ResNet50(C, C, False)
    model.load_state_dict(sd)
    model = model.to(cuda).train()
    x = torch.rand(B, 3, 224, 224, generator=gen)
    y = torch.randint(0, C, (B,), generator=gen)
    y[torch.rand(B, generator=gen) < 0.5] = -1
    logits, _ = model(x.to(cuda))
    j = EntropicOpensetLoss(C, 1.0)(logits, y.to(cuda))
    j.backward()
    torch.cuda.synchronize()
    gates = hip_gates(model)
    t0 = time.time()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    r_pin = R.forward_backward(sd64, x.double(), y, lambda lg, t, f: L.entropic_openset_loss(lg, t, 1.0), gates=gates)
    del gates
    import resource
    print(f"fp64 oracle forward + backward at B = {B}: {time.time() - t0:.0f} s, peak host memory {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20:.1f} GiB")
    assert float((logits.detach().cpu().double() - r_pin[0]).abs().max()) <= LOGIT_TOL
    assert abs(float(j.detach()) - float(r_pin[2])) <= 1e-5 * max(1.0, abs(float(r_pin[2])))
    named = dict(model.named_parameters())
    errs = {k: _rel(named[k].grad.detach().cpu(), r_pin[3][k]) for k in R.param_keys(sd)}
    worst = max(errs, key=errs.get)
    print(f"B = {B}, 224 x 224, C = {C}: gradient rel-L2 vs fp64 under the HIP gates: median {np.median(list(errs.values())):.2e} max {errs[worst]:.2e} ({worst})")
    assert len(errs) == 162
    for k, e in errs.items():
        assert e <= GRAD_TOL, f"grad {k}: rel-L2 {e:.2e} > {GRAD_TOL:.0e} under pinned gates"
    assert np.median(list(errs.values())) <= 1.5e-4


def test_gates_read_back_are_the_decisions_the_forward_took(cuda):
    """The debug read-out against an independent recomputation: block-output gates == (stored activation > 0), the stem gate ==
    (pooled value > 0) with the arg-max pointing at an element that attains the window maximum of the oracle's bn1 output."""
    from openset_imagenet import ResNet50
    from oracle import resnet50_oracle as R
    from osi_testlib import hip_gates
    gen = torch.Generator().manual_seed(3)
    C, B, HW = 10, 4, 64
    sd = R.randomize_bn(R.init_state(C, C, False, generator=gen), generator=gen)
    model = ResNet50(C, C, False)
    model.load_state_dict(sd)
    model = model.to(cuda).train()
    x = torch.rand(B, 3, HW, HW, generator=gen)
    with torch.no_grad():
        model(x.to(cuda))
    torch.cuda.synchronize()
    gates = hip_gates(model)
    taps, rec = {}, {}
    R.forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double(), True, taps=taps, record_gates=rec)
    flips, pool_flips, total = R.gate_disagreements(gates, rec)
    assert flips <= 20 and pool_flips <= 5, (flips, pool_flips)
    # arg-max indices address the 3x3 / stride 2 / pad 1 window of their output pixel
    idx = gates["pool_idx"]
    Hs = HW // 2
    ho = torch.arange(idx.shape[2]).view(1, 1, -1, 1)
    wo = torch.arange(idx.shape[3]).view(1, 1, 1, -1)
    h, w = idx // Hs, idx % Hs
    assert ((h - 2 * ho).abs() <= 1).all() and ((w - 2 * wo).abs() <= 1).all() and (idx >= 0).all()
    # every gate has the oracle's shape
    for g, r in zip(gates["relu"], rec["relu"]):
        assert g.shape == r.shape
