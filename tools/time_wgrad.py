"""Dev tool: TFLOP/s of the weight gradient of a few 3x3 layers exactly as the executor calls it (fused input activation), after a
burst that settles the clock; OSI_HIP_LIB selects the library (tools/probes/ablate.sh runs it over ablated builds).
usage: python tools/time_wgrad.py [B]"""
import ctypes, os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(64, 64, 3, 1, 56), (128, 128, 3, 1, 28), (256, 256, 3, 1, 14), (512, 512, 3, 1, 7), (128, 128, 3, 2, 56), (256, 256, 3, 2, 28)]
if os.environ.get("OSI_TW_SHAPES") == "1x1":      # bottleneck conv1 (plain input) and conv3 (fused input activation) of each stage
    SHAPES = [(256, 64, 1, 1, 56), (64, 256, 1, 1, 56), (512, 128, 1, 1, 28), (128, 512, 1, 1, 28), (1024, 256, 1, 1, 14), (256, 1024, 1, 1, 14),
              (2048, 512, 1, 1, 7), (512, 2048, 1, 1, 7)]
L = N.lib(); dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
out = []
for Cin, Cout, k, s, H in SHAPES:
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, s, 1 if k == 3 else 0)
    x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, d.Ho, d.Wo, Cout, device=dev); dw = torch.empty(Cout, k, k, Cin, device=dev)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.5
    nb = L.osi_conv_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    if k == 3 or Cin < Cout:
        fn = lambda: N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(ws), nb, st))
    else:
        fn = lambda: N.check(L.osi_conv_wgrad(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(dw), N.ptr(ws), nb, st))
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
print(os.path.basename(os.environ.get("OSI_HIP_LIB", "libosi_hip.so")), " ".join(out), flush=True)
