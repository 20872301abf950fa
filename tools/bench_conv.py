"""Dev tool: per-layer TFLOP/s of the implicit-GEMM conv kernels on the 23 unique ResNet-50 shapes
for every tile configuration. Usage: python tools/bench_conv.py [B] [reps]"""
import ctypes, os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# (Cin, Cout, k, stride, Hin, count)
SHAPES = [(64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2), (256, 128, 1, 1, 56, 1),
          (128, 128, 3, 2, 56, 1), (128, 512, 1, 1, 28, 4), (256, 512, 1, 2, 56, 1), (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3),
          (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1), (256, 1024, 1, 1, 14, 6), (512, 1024, 1, 2, 28, 1), (1024, 256, 1, 1, 14, 5),
          (256, 256, 3, 1, 14, 5), (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (1024, 2048, 1, 2, 14, 1),
          (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2)]
L = N.lib()
S = lambda: torch.cuda.current_stream().cuda_stream
dev = torch.device("cuda")


STEADY_MS = float(os.environ.get("OSI_STEADY_MS", "0"))   # > 0: time only after this many ms of back-to-back launches of the same call
# (the shader clock needs ~30 ms of load to settle after an idle gap; without this the table is 10-15 % low)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    if STEADY_MS > 0:
        import time
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < STEADY_MS:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()      # keeps the queue short; the gap is microseconds, far below the clock's time constant
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
print(f"B={B}  TFLOP/s per tile: auto,128x128,64x128,64x64,64x64_S1,64x128_S1,128x128_S1,128x64_S1[,auto with fused input activation[,auto recomputing a block output (1x1 s1)]]; wgrad: plain, fused")
for Cin, Cout, k, s, H, cnt in SHAPES:
    pad = 1 if k == 3 else 0
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, s, pad)
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    y = torch.empty(B, d.Ho, d.Wo, Cout, device=dev)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    nb = L.osi_conv_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    flop = 2.0 * B * d.Ho * d.Wo * Cout * Cin * k * k
    res = {}
    for name, cdim in (("fwd", Cout), ("dgrad", Cin)):
        r = []
        for tile in (0, 1, 3, 4, 5, 6, 7, 8):
            if tile in (1, 3, 6, 7) and cdim % 128:
                r.append(0.0); continue
            if name == "fwd":
                f = lambda: N.check(L.osi_conv_fwd(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), tile, S()))
            else:
                f = lambda: N.check(L.osi_conv_dgrad(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), 0, tile, S()))
            r.append(flop / timeit(f) / 1e9)
        res[name] = r
    f = lambda: N.check(L.osi_conv_wgrad(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(dw), N.ptr(ws), nb, S()))
    res["wgrad"] = [flop / timeit(f) / 1e9]
    # fused input activation (BatchNorm + ReLU of the producer applied in the loader): AUTO tile
    sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.5
    f = lambda: N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, None, 0, None, None, S()))
    res["fwd"].append(flop / timeit(f) / 1e9)
    if k == 1 and s == 1:   # conv1 recomputing the previous block output: relu(x * sc + sh + res)
        res_t = torch.rand_like(x)
        f = lambda: N.check(L.osi_conv_fwd_act2(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(res_t), N.ptr(w), N.ptr(y), 0, None, 0, None, None, S()))
        res["fwd"].append(flop / timeit(f) / 1e9)
    f = lambda: N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(ws), nb, S()))
    res["wgrad"].append(flop / timeit(f) / 1e9)
    for n in res:
        tot[n][0] += cnt * flop; tot[n][1] += cnt * flop / (res[n][0] * 1e9)
    fm = lambda r: "/".join(f"{v:5.1f}" for v in r)
    print(f"{Cin:4d}->{Cout:4d} k{k} s{s} H{H:3d} x{cnt} | fwd {fm(res['fwd'])} | dgrad {fm(res['dgrad'])} | wgrad {fm(res['wgrad'])}", flush=True)
for n, (fl, ms) in tot.items():
    print(f"{n}: {fl / 1e9:.1f} GFLOP in {ms:.2f} ms with AUTO tiles = {fl / ms / 1e9:.1f} TFLOP/s (stem excluded)")
