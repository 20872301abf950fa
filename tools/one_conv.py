"""Dev tool: launch ONE conv configuration a few times (for rocprofv3 --pmc runs).
usage: one_conv.py mode(fwd|dgrad|wgrad) Cin Cout k stride H tile [B] [reps]"""
import ctypes, os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import _native as N
mode = sys.argv[1]
Cin, Cout, k, s, H, tile = (int(a) for a in sys.argv[2:8])
B = int(sys.argv[8]) if len(sys.argv) > 8 else 128
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 3
L = N.lib(); dev = torch.device("cuda")
pad = 1 if k == 3 else 0
d = N.ConvDesc.make(B, H, H, Cin, Cout, k, s, pad)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
y = torch.empty(B, d.Ho, d.Wo, Cout, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
nb = L.osi_conv_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    if mode == "fwd":
        N.check(L.osi_conv_fwd(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), tile, st))
    elif mode == "dgrad":
        N.check(L.osi_conv_dgrad(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), 0, tile, st))
    else:
        N.check(L.osi_conv_wgrad(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(dw), N.ptr(ws), nb, st))
torch.cuda.synchronize()
print("done", mode, Cin, Cout, k, s, H, tile, "GFLOP", 2.0 * B * d.Ho * d.Wo * Cout * Cin * k * k / 1e9)
