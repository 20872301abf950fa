"""Dev tool: TFLOP/s of a few forward convolutions exactly as the executor calls them (BatchNorm statistics epilogue; fused input
activation where the executor fuses it) after a burst that settles the clock; OSI_HIP_LIB selects the library (ablated builds).
usage: python tools/time_fwd.py [B]"""
import ctypes, os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
# (Cin, Cout, k, stride, H, fused input activation)
SHAPES = [(64, 64, 3, 1, 56, 1), (64, 256, 1, 1, 56, 1), (256, 64, 1, 1, 56, 0), (128, 128, 3, 1, 28, 1), (512, 128, 1, 1, 28, 0),
          (256, 256, 3, 1, 14, 1), (256, 1024, 1, 1, 14, 1), (1024, 256, 1, 1, 14, 0), (512, 512, 3, 1, 7, 1), (512, 2048, 1, 1, 7, 1)]
L = N.lib(); dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
out = []
for Cin, Cout, k, s, H, act in SHAPES:
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, s, 1 if k == 3 else 0)
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05; y = torch.empty(B, d.Ho, d.Wo, Cout, device=dev)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.5
    nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d)); ps = torch.empty(max(nb, 16) // 4, device=dev)
    P, rows = ctypes.c_int(), ctypes.c_int()
    if act:
        fn = lambda: N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st))
    else:
        fn = lambda: N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st))
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
    ms = sorted(best)[2]
    out.append(f"{2.0 * B * d.Ho * d.Wo * Cout * Cin * k * k / ms / 1e9:6.1f}")
print(f"{os.path.basename(os.environ.get('OSI_HIP_LIB', 'libosi_hip.so')):24s}", " ".join(out), flush=True)
