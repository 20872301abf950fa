"""Dev tool: the forward / input-gradient convolutions exactly as the executor calls them (statistics epilogue, fused input
activation, fused dgrad epilogue), with the balanced remainder (osi_set_tuning("tail_split")) off and on, interleaved in ONE
process on the same buffers (variants A/B/A/B..., median of the rounds). Usage: python tools/bench_tail.py [B] [rounds] [fwd|dgrad|both]
Any other process-wide knob can take the place of tail_split: OSI_AB_KNOB=wave_prio OSI_AB_VALUES=0,2 (restored to the first value)."""
import ctypes, os, statistics, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
WHAT = sys.argv[3] if len(sys.argv) > 3 else "both"
# (Cin, Cout, k, stride, Hin, count, fused input activation in the executor's forward)
SHAPES = [(64, 64, 1, 1, 56, 1, 0), (64, 64, 3, 1, 56, 3, 1), (64, 256, 1, 1, 56, 4, 1), (256, 64, 1, 1, 56, 2, 0), (256, 128, 1, 1, 56, 1, 0),
          (128, 128, 3, 2, 56, 1, 1), (128, 512, 1, 1, 28, 4, 1), (256, 512, 1, 2, 56, 1, 0), (512, 128, 1, 1, 28, 3, 0), (128, 128, 3, 1, 28, 3, 1),
          (512, 256, 1, 1, 28, 1, 0), (256, 256, 3, 2, 28, 1, 1), (256, 1024, 1, 1, 14, 6, 1), (512, 1024, 1, 2, 28, 1, 0), (1024, 256, 1, 1, 14, 5, 0),
          (256, 256, 3, 1, 14, 5, 1), (1024, 512, 1, 1, 14, 1, 0), (512, 512, 3, 2, 14, 1, 1), (512, 2048, 1, 1, 7, 3, 1), (1024, 2048, 1, 2, 14, 1, 0),
          (2048, 512, 1, 1, 7, 2, 0), (512, 512, 3, 1, 7, 2, 1)]
KNOB = os.environ.get("OSI_AB_KNOB", "tail_split").encode()
VA, VB = (int(v) for v in os.environ.get("OSI_AB_VALUES", "0,1").split(","))
L = N.lib()
S = lambda: torch.cuda.current_stream().cuda_stream
dev = torch.device("cuda")


class Fusion(ctypes.Structure):
    _fields_ = [("relu_mask", ctypes.c_void_p), ("y0", ctypes.c_void_p), ("mean0", ctypes.c_void_p), ("invstd0", ctypes.c_void_p),
                ("y1", ctypes.c_void_p), ("mean1", ctypes.c_void_p), ("invstd1", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("partials_bytes", ctypes.c_size_t), ("scale0", ctypes.c_void_p), ("shift0", ctypes.c_void_p),
                ("pool_idx", ctypes.c_void_p), ("pool_H", ctypes.c_int), ("pool_W", ctypes.c_int), ("addend_stride", ctypes.c_int)]


def burst(fn, ms=40):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ab(fn):
    """median ms of fn with tail_split 0 and 1, interleaved rounds, each after a burst that keeps the clock settled"""
    t = {VA: [], VB: []}
    for r in range(ROUNDS):
        for v in (VA, VB) if r % 2 == 0 else (VB, VA):
            N.check(L.osi_set_tuning(KNOB, v))
            burst(fn)
            t[v].append(timed(fn))
    N.check(L.osi_set_tuning(KNOB, 1 if KNOB == b"tail_split" else VA))
    return statistics.median(t[VA]), statistics.median(t[VB])


tot = {"fwd": [0.0, 0.0, 0.0], "dgrad": [0.0, 0.0, 0.0]}
print(f"B={B}: TFLOP/s as the executor calls it, {KNOB.decode()} {VA} -> {VB} (median of {ROUNDS} interleaved rounds)")
for Cin, Cout, k, s, H, cnt, fused in SHAPES:
    pad = 1 if k == 3 else 0
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, s, pad)
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    y = torch.empty(B, d.Ho, d.Wo, Cout, device=dev)
    flop = 2.0 * B * d.Ho * d.Wo * Cout * Cin * k * k
    line = f"{Cin:4d}->{Cout:4d} k{k} s{s} H{H:3d} x{cnt}"
    if WHAT in ("fwd", "both"):
        nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d))
        ps = torch.empty(nb // 4, device=dev)
        P, rows = ctypes.c_int(), ctypes.c_int()
        sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.5
        if fused:
            f = lambda: N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), S()))
        else:
            f = lambda: N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), S()))
        t0, t1 = ab(f)
        tot["fwd"][0] += cnt * flop; tot["fwd"][1] += cnt * t0; tot["fwd"][2] += cnt * t1
        line += f" | fwd{'*' if fused else ' '} {flop / t0 / 1e9:6.1f} -> {flop / t1 / 1e9:6.1f} ({(t0 - t1) * 1e3 * cnt:+6.1f} us/step)"
    if WHAT in ("dgrad", "both"):
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        M = B * H * H
        y0 = torch.randn(M, Cin, device=dev)
        mean0, inv0 = torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5
        sc0, sh0 = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.5
        pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d))
        parts = torch.empty(max(pb // 4, 4), device=dev)
        fz = Fusion(None, y0.data_ptr(), mean0.data_ptr(), inv0.data_ptr(), None, None, None, parts.data_ptr(), pb, sc0.data_ptr(), sh0.data_ptr())
        Pd = ctypes.c_int()
        f = lambda: N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), None, ctypes.byref(fz), 0, ctypes.byref(Pd), S()))
        t0, t1 = ab(f)
        tot["dgrad"][0] += cnt * flop; tot["dgrad"][1] += cnt * t0; tot["dgrad"][2] += cnt * t1
        line += f" | dgrad(fused) {flop / t0 / 1e9:6.1f} -> {flop / t1 / 1e9:6.1f} ({(t0 - t1) * 1e3 * cnt:+6.1f} us/step)"
    print(line, flush=True)
for n, (fl, a, b) in tot.items():
    if fl:
        print(f"{n}: {fl / 1e9:.1f} GFLOP  off {a:.3f} ms = {fl / a / 1e9:.1f} TFLOP/s   on {b:.3f} ms = {fl / b / 1e9:.1f} TFLOP/s (stem excluded)")
