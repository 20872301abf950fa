"""Dev tool: TFLOP/s of a few input-gradient convolutions in the executor's "in-block" fused form (ReLU gate recomputed from the
producer's pre-BN tensor + BatchNorm-backward sums in the epilogue) after a burst that settles the clock; OSI_HIP_LIB selects the
library (ablated builds). usage: python tools/time_dgrad.py [B]"""
import ctypes, os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd"), os.path.join(ROOT, "tests")]
import torch
from openset_imagenet import _native as N
import osi_testlib as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (128, 128, 3, 1, 28), (128, 512, 1, 1, 28), (256, 256, 3, 1, 14), (256, 1024, 1, 1, 14),
          (512, 512, 3, 1, 7), (512, 2048, 1, 1, 7)]
L = N.lib(); dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
out = []
for Cin, Cout, k, s, H in SHAPES:
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, s, 1 if k == 3 else 0)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(B, d.Ho, d.Wo, Cout, device=dev); dx = torch.empty(B, H, H, Cin, device=dev)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.5
    pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d)); parts = torch.empty(max(pb, 16) // 4, device=dev)
    y0 = torch.randn(B * H * H, Cin, device=dev)
    mean0, inv0 = y0.mean(0), 1 / torch.sqrt(y0.var(0, unbiased=False) + 1e-5)
    f = T.Fusion(None, y0.data_ptr(), mean0.data_ptr(), inv0.data_ptr(), None, None, None, parts.data_ptr(), pb, sc.data_ptr(), sh.data_ptr())
    P = ctypes.c_int()
    fn = lambda: N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), None, ctypes.byref(f), 0, ctypes.byref(P), st))
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
