#!/usr/bin/env python3
"""Dev (GPU box): the Winograd kernels of the step, layer by layer, in a FIXED launch sequence for a counter pass:
    rocprofv3 --kernel-trace --pmc <counters> -d <dir> -o p --output-format csv -- python3 tools/wino_pmc_run.py <labels.json> [B] [reps]
For each of the four 3x3 stride-1 shapes of ResNet-50: forward with the fused input activation + BatchNorm partials
(osi_conv_fwd_wino_pre), in-block fused input gradient (osi_conv_dgrad_fused_wino_pre), weight gradient with the fused input activation
(osi_conv_wgrad_wino) — as the executor issues them, `reps` launches each after one warm-up, plus the direct twins of the forward / input
gradient (osi_conv_fwd_act / osi_conv_dgrad_fused) for the same counters. The k-th dispatch of a kernel family in the trace is the k-th
entry of that family's label list (written to <labels.json>), which is how tools/wino_pmc_summary.py names the rows."""
import ctypes
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from openset_imagenet import _native as N  # noqa: E402
import osi_testlib as T  # noqa: E402

labels_path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = N.lib(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
labels = {"k_wino<": [], "k_wino_wgrad<": [], "k_conv_fwd<": [], "k_conv_dgrad<": []}
g = torch.Generator(device=dev).manual_seed(1)
for C, H in [(64, 56), (128, 28), (256, 14), (512, 7)]:
    d = N.ConvDesc.make(B, H, H, C, C, 3, 1, 1)
    x = torch.randn(B, H, H, C, device=dev, generator=g); w = torch.randn(C, 3, 3, C, device=dev, generator=g) * 0.05
    y = torch.empty(B, H, H, C, device=dev)
    sc, sh = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.5
    tiles = B * ((H + 1) // 2) ** 2
    nb = max(L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d)), (2 * ((tiles + 15) // 16) + 64) * C * 4)
    ps = torch.empty(nb // 4, device=dev)
    ub, sb = L.osi_conv_wino_weights_bytes(ctypes.byref(d)), L.osi_conv_wino_slab_bytes()
    uf, ubw = torch.empty(ub, dtype=torch.uint8, device=dev), torch.empty(ub, dtype=torch.uint8, device=dev)
    slab = torch.empty(sb, dtype=torch.uint8, device=dev)
    N.check(L.osi_conv_wino_transform_weights(ctypes.byref(d), N.ptr(w), 0, N.ptr(uf), ub, st))
    N.check(L.osi_conv_wino_transform_weights(ctypes.byref(d), N.ptr(w), 1, N.ptr(ubw), ub, st))
    P, rows = ctypes.c_int(), ctypes.c_int()
    pb = max(L.osi_conv_dgrad_fused_workspace(ctypes.byref(d)), 3 * ((tiles + 15) // 16) * C * 4)
    parts = torch.empty(pb // 4, device=dev)
    mean0, inv0 = torch.randn(C, device=dev, generator=g) * 0.1, torch.rand(C, device=dev, generator=g) + 0.5
    f = T.Fusion(None, x.data_ptr(), mean0.data_ptr(), inv0.data_ptr(), None, None, None, parts.data_ptr(), pb, sc.data_ptr(), sh.data_ptr())
    dy = torch.randn(B, H, H, C, device=dev, generator=g); dx = torch.empty(B, H, H, C, device=dev)
    gwb = L.osi_conv_wgrad_wino_workspace(ctypes.byref(d)); gwws = torch.empty(max(gwb, 16), dtype=torch.uint8, device=dev)
    dw = torch.empty(C, 3, 3, C, device=dev)
    fams = [
        ("k_wino<", f"fwd {C}@{H}", lambda: L.osi_conv_fwd_wino_pre(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(uf), N.ptr(y), N.ptr(slab), sb, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st)),
        ("k_wino<", f"dgrad {C}@{H}", lambda: L.osi_conv_dgrad_fused_wino_pre(ctypes.byref(d), N.ptr(dy), N.ptr(ubw), N.ptr(dx), ctypes.byref(f), N.ptr(slab), sb, ctypes.byref(P), st)),
        ("k_wino_wgrad<", f"wgrad {C}@{H}", lambda: L.osi_conv_wgrad_wino(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(gwws), gwb, st)),
        ("k_conv_fwd<", f"fwd-direct {C}@{H}", lambda: L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st)),
        ("k_conv_dgrad<", f"dgrad-direct {C}@{H}", lambda: L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), None, ctypes.byref(f), 0, ctypes.byref(P), st)),
    ]
    for fam, label, fn in fams:
        for i in range(reps + 1):
            N.check(fn(), label)
            labels[fam].append(label + (" (warm-up)" if i == 0 else ""))
        torch.cuda.synchronize()
json.dump({"batch": B, "reps": reps, "labels": labels}, open(labels_path, "w"))
print("ok")
