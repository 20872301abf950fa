"""Dev tool: the Winograd forms of the 3x3 stride-1 layers next to the direct kernels, per layer shape at batch B, exactly as the executor
calls them (forward: fused input activation + BatchNorm partials; input gradient: in-block fused epilogue with its sums).
usage: python tools/time_wino.py [B]      (under rocprofv3 --kernel-trace --stats the kernels of one form separate: k_wino / fixup / weights)"""
import ctypes, os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd"), os.path.join(ROOT, "tests")]
import torch
from openset_imagenet import _native as N
import osi_testlib as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
only = sys.argv[2] if len(sys.argv) > 2 else "all"      # "wino" / "direct": one family only (profiling)
L = N.lib(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream


def bench(fn):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20)
    return sorted(best)[2]


print(f"B={B}: us per launch (direct-conv TFLOP/s)   fwd direct | fwd wino | dgrad direct | dgrad wino | wgrad direct | wgrad wino")
for C, H in [(64, 56), (128, 28), (256, 14), (512, 7)]:
    d = N.ConvDesc.make(B, H, H, C, C, 3, 1, 1)
    M = B * H * H
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05; y = torch.empty(B, H, H, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.5
    nb = max(L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d)), (2 * ((B * ((H + 1) // 2) ** 2 + 15) // 16) + 64) * C * 4)
    ps = torch.empty(nb // 4, device=dev)
    wb = L.osi_conv_wino_workspace(ctypes.byref(d)); ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    P, rows = ctypes.c_int(), ctypes.c_int()
    pb = max(L.osi_conv_dgrad_fused_workspace(ctypes.byref(d)), 3 * ((B * ((H + 1) // 2) ** 2 + 15) // 16) * C * 4)
    parts = torch.empty(pb // 4, device=dev)
    mean0, inv0 = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    f = T.Fusion(None, x.data_ptr(), mean0.data_ptr(), inv0.data_ptr(), None, None, None, parts.data_ptr(), pb, sc.data_ptr(), sh.data_ptr())
    dy = torch.randn(B, H, H, C, device=dev); dx = torch.empty(B, H, H, C, device=dev)
    fns = {
        "fd": lambda: N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st)),
        "fw": lambda: N.check(L.osi_conv_fwd_wino(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), N.ptr(ws), wb, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st)),
        "dd": lambda: N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), None, ctypes.byref(f), 0, ctypes.byref(P), st)),
        "dw": lambda: N.check(L.osi_conv_dgrad_fused_wino(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), ctypes.byref(f), N.ptr(ws), wb, ctypes.byref(P), st)),
    }
    gb = L.osi_conv_wgrad_workspace(ctypes.byref(d)); gws = torch.empty(max(gb, 16), dtype=torch.uint8, device=dev)
    gwb = L.osi_conv_wgrad_wino_workspace(ctypes.byref(d)); gwws = torch.empty(max(gwb, 16), dtype=torch.uint8, device=dev)
    dw = torch.empty(C, 3, 3, C, device=dev)
    fns["gd"] = lambda: N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(gws), gb, st))
    fns["gw"] = lambda: N.check(L.osi_conv_wgrad_wino(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(gwws), gwb, st))
    gf = 2.0 * M * C * C * 9 / 1e6
    out = []
    for k in ("fd", "fw", "dd", "dw", "gd", "gw"):
        if (only == "wino" and k[1] != "w") or (only == "direct" and k[1] != "d"):
            out.append("      -      "); continue
        ms = bench(fns[k])
        out.append(f"{ms * 1e3:7.1f} ({gf / ms / 1e3:5.1f})")
    print(f"{C:4d}->{C:4d} @{H:2d}^2  " + " | ".join(out), flush=True)
