"""Dev tool (GPU box): the "block input" input gradient (conv1 of every bottleneck: k_conv_dgrad<1,1,1,3> — addend, stored ReLU bitmask, output,
BatchNorm-backward sums of one or two consumers in the epilogue), launch by launch as the executor issues it at batch B, against both of its
floors: the matrix pipe (2 M N K FLOP at 157.3 TFLOP/s) and HBM (dY + addend + y0 [+ y1] + mask read, dX written, at 6.0 TB/s).
    python tools/time_dgrad_blockin.py [B]        OSI_DEV=1 OSI_HIP_LIB=<lib> selects another build of the library (A/B)
Shapes in FORWARD terms (Cin -> Cout of conv1, H): the gradient has K = Cout, N = Cin."""
import ctypes, os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd"), os.path.join(ROOT, "tests")]
import torch
from openset_imagenet import _native as N
import osi_testlib as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
# (Cin, Cout, H, consumers, sparse addend, launches per step)
SHAPES = [(256, 64, 56, 2, 0, 1), (256, 64, 56, 1, 0, 1), (256, 128, 56, 1, 1, 1), (512, 128, 28, 2, 0, 1), (512, 128, 28, 1, 0, 2), (512, 256, 28, 1, 1, 1),
          (1024, 256, 14, 2, 0, 1), (1024, 256, 14, 1, 0, 4), (1024, 512, 14, 1, 1, 1), (2048, 512, 7, 2, 0, 1), (2048, 512, 7, 1, 0, 1)]
L = N.lib(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
print(f"B={B}  fwd-shape          us/launch   TFLOP/s  of MFMA floor  of HBM floor (6.0 TB/s)   x launches/step")
total = 0.0
for Cin, Cout, H, cons, sparse, count in SHAPES:
    d = N.ConvDesc.make(B, H, H, Cin, Cout, 1, 1, 0)
    M = B * H * H
    g = torch.Generator(device=dev).manual_seed(Cin * 7 + Cout + H)
    dy = torch.randn(B, H, H, Cout, device=dev, generator=g); w = torch.randn(Cout, 1, 1, Cin, device=dev, generator=g) * 0.05
    pre = torch.randn(M, Cin, device=dev, generator=g)
    ones, zeros = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    out = torch.empty(M, Cin, device=dev)
    mask = torch.zeros(L.osi_bn_relu_mask_bytes(M, Cin), dtype=torch.uint8, device=dev)
    N.check(L.osi_bn_apply_relu_mask(N.ptr(pre), None, N.ptr(ones), N.ptr(zeros), N.ptr(out), N.ptr(mask), M, Cin, st))
    del out, pre
    ys = [torch.randn(M, Cin, device=dev, generator=g) * 2 + 0.5 for _ in range(cons)]
    stats = [(y.mean(0), 1 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)) for y in ys]
    addend = torch.randn(B, H, H, Cin, device=dev, generator=g)
    dx = addend if sparse else torch.empty(B, H, H, Cin, device=dev)
    pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d)); parts = torch.empty(max(pb, 16) // 4, device=dev)
    f = T.Fusion(mask.data_ptr(), ys[0].data_ptr(), stats[0][0].data_ptr(), stats[0][1].data_ptr(),
                 ys[1].data_ptr() if cons == 2 else None, stats[1][0].data_ptr() if cons == 2 else None, stats[1][1].data_ptr() if cons == 2 else None,
                 parts.data_ptr(), pb, None, None, None, 0, 0, 2 if sparse else 1)
    P = ctypes.c_int()
    fn = lambda: N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), N.ptr(addend), ctypes.byref(f), 0, ctypes.byref(P), st))
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
    gf = 2.0 * M * Cin * Cout / 1e9
    byt = 4.0 * M * (Cout + Cin * (1 + cons + (0.25 if sparse else 1.0)) + Cin / 32.0)
    t_mfma, t_hbm = gf / 157.3e3 * 1e3, byt / 6.0e12 * 1e3
    total += ms * count
    print(f"  {Cin:4d}->{Cout:3d} @{H:2d} c{cons}{' sparse' if sparse else '       '} {ms * 1e3:8.1f}  {gf / ms:8.1f}   {t_mfma / ms:8.2f}      {t_hbm / ms:8.2f}              x{count}", flush=True)
print(f"sum over the step's 15 launches: {total:.3f} ms")
