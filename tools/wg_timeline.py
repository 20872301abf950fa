"""Dev tool (GPU box): where does a conv LAUNCH lose its matrix-pipe time at the dispatch level? Runs one convolution through the
diagnostic build of the C ABI (make -C openset-imagenet_amd/csrc stamps -> libosi_hip_stamps.so: every workgroup stores the 100 MHz
real-time counter at kernel entry / K loop entered / K loop done / epilogue done + its HW_ID / XCC_ID) and reconstructs the per-CU
timeline of the last of a block of back-to-back launches (steady clocks).

    python tools/wg_timeline.py fwd|fwd_act|dgrad|dgrad_inblock|wgrad|wgrad_act Cin Cout k stride H [B] [--json out.json]

Reported, as shares of (CUs x launch span):
  mfma_work      sum over workgroups of their K tiles x 16 MFMAs x 64 cycles per SIMD, at the launch's clock (GRBM-free estimate: the
                 median in-kernel clock is not read here; the share is computed against the NOMINAL 2.4 GHz and against the span)
  ramp           CU time before the CU's first workgroup entered its K loop (launch latency + prologue of the first round)
  tail           CU time after the CU's last workgroup left its K loop (ragged end: other CUs still working, epilogue of the last round)
  between        everything else that is not matrix work: prologues / epilogues of later rounds that nothing covered, loop stalls
plus workgroup phase lengths (prologue / loop / epilogue, median and p90) and resident workgroups per CU over time.
"""
import argparse
import ctypes
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("OSI_HIP_LIB", os.path.join(ROOT, "openset-imagenet_amd", "csrc", "libosi_hip_stamps.so"))
os.environ["OSI_DEV"] = "1"          # the package only honours OSI_HIP_LIB in dev mode
if not os.path.isfile(os.environ["OSI_HIP_LIB"]):      # built on demand: __graft_entry__.build() only builds the product
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "openset-imagenet_amd", "csrc"), "stamps"], check=True)
import torch  # noqa: E402
from openset_imagenet import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode"); ap.add_argument("Cin", type=int); ap.add_argument("Cout", type=int); ap.add_argument("k", type=int)
    ap.add_argument("stride", type=int); ap.add_argument("H", type=int); ap.add_argument("B", type=int, nargs="?", default=128)
    ap.add_argument("--warm", type=int, default=300); ap.add_argument("--json", default=None)
    a = ap.parse_args()
    L = N.lib()
    setst = L.osi_debug_set_stamps
    setst.restype = None; setst.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda")
    Cin, Cout, k, s, H, B = a.Cin, a.Cout, a.k, a.stride, a.H, a.B
    pad = 1 if k == 3 else 0
    d = N.ConvDesc.make(B, H, H, Cin, Cout, k, s, pad)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, H, Cin, device=dev, generator=g)
    w = torch.randn(Cout, k, k, Cin, device=dev, generator=g) * 0.05
    y = torch.empty(B, d.Ho, d.Wo, Cout, device=dev)
    dy = torch.randn(B, d.Ho, d.Wo, Cout, device=dev, generator=g)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    sc, sh = torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.5
    st = torch.cuda.current_stream().cuda_stream
    nb = L.osi_conv_fwd_bnstats_workspace(ctypes.byref(d)); ps = torch.empty(max(nb, 16) // 4, device=dev)
    wb = L.osi_conv_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(max(wb, 16), dtype=torch.uint8, device=dev)
    P, rows = ctypes.c_int(), ctypes.c_int()
    import osi_testlib as T
    if a.mode == "dgrad_inblock":
        pb = L.osi_conv_dgrad_fused_workspace(ctypes.byref(d)); parts = torch.empty(max(pb, 16) // 4, device=dev)
        y0 = torch.randn(B * H * H, Cin, device=dev, generator=g)
        mean0, inv0 = y0.mean(0), 1 / torch.sqrt(y0.var(0, unbiased=False) + 1e-5)
        f = T.Fusion(None, y0.data_ptr(), mean0.data_ptr(), inv0.data_ptr(), None, None, None, parts.data_ptr(), pb, sc.data_ptr(), sh.data_ptr())

    def launch():
        if a.mode == "fwd":
            N.check(L.osi_conv_fwd_bnstats(ctypes.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st))
        elif a.mode == "fwd_act":
            N.check(L.osi_conv_fwd_act(ctypes.byref(d), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(w), N.ptr(y), 0, N.ptr(ps), nb, ctypes.byref(P), ctypes.byref(rows), st))
        elif a.mode == "dgrad":
            N.check(L.osi_conv_dgrad(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), 0, 0, st))
        elif a.mode == "dgrad_inblock":
            N.check(L.osi_conv_dgrad_fused(ctypes.byref(d), N.ptr(dy), N.ptr(w), N.ptr(dx), None, ctypes.byref(f), 0, ctypes.byref(P), st))
        elif a.mode == "wgrad":
            N.check(L.osi_conv_wgrad(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(dw), N.ptr(ws), wb, st))
        elif a.mode == "wgrad_act":
            N.check(L.osi_conv_wgrad_act(ctypes.byref(d), N.ptr(dy), N.ptr(x), N.ptr(sc), N.ptr(sh), N.ptr(dw), N.ptr(ws), wb, st))
        else:
            raise SystemExit("unknown mode")

    cap = 1 << 20                                        # workgroups: larger than any grid launched here
    buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
    setst(None)
    for _ in range(a.warm):                              # steady clocks (the shader clock needs ~30 ms of load after idle)
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record(); torch.cuda.synchronize()
    us_plain = e0.elapsed_time(e1) / 20 * 1e3
    setst(buf.data_ptr())
    launch()
    torch.cuda.synchronize()
    setst(None)
    s8 = buf.view(cap, 8).cpu().numpy()
    live = s8[:, 3] > 0                                  # workgroups that ran to the end (early-exit padding blocks never stamp [3])
    v = s8[live]
    t0 = v[:, 0].min()
    tb, tl0, tl1, te = [(v[:, i] - t0) / 100.0 for i in range(4)]     # microseconds (100 MHz counter)
    span = float(te.max())
    cu_key = (v[:, 5] & 0xF) * 65536 + ((v[:, 4] >> 8) & 0xFF)       # (XCC, SE/SH/CU bits of HW_ID)
    groups = defaultdict(list)
    for i, kk in enumerate(cu_key):
        groups[int(kk)].append(i)
    ncu = len(groups)
    flops = 2.0 * B * d.Ho * d.Wo * Cout * Cin * k * k
    # matrix work: the launch's FLOPs at the nominal pipe rate (256 FLOP / cycle / CU at 2.4 GHz)
    work_us_per_cu = flops / (ncu * 256.0 * 2400.0)   # microseconds of pipe time per CU
    ramp = sum(min(tl0[i] for i in idx) for idx in groups.values())
    tail = sum(span - max(tl1[i] for i in idx) for idx in groups.values())
    total = ncu * span
    import numpy as np
    res = {
        "mode": a.mode, "shape": {"Cin": Cin, "Cout": Cout, "k": k, "stride": s, "H": H, "B": B},
        "workgroups": int(live.sum()), "cus_seen": ncu, "workgroups_per_cu": round(float(live.sum()) / ncu, 2),
        "launch_us_plain": round(us_plain, 1), "launch_span_us_stamped": round(span, 1),
        "tflops_plain": round(flops / us_plain / 1e6, 1),
        "share_of_cu_time": {
            "mfma_work_at_2.4GHz": round(work_us_per_cu * ncu / total, 4),
            "ramp_before_first_loop": round(ramp / total, 4),
            "tail_after_last_loop": round(tail / total, 4),
        },
        "workgroup_us": {n_: {"median": round(float(np.median(x_)), 2), "p90": round(float(np.percentile(x_, 90)), 2)}
                         for n_, x_ in (("prologue", tl0 - tb), ("k_loop", tl1 - tl0), ("epilogue", te - tl1), ("lifetime", te - tb))},
        "first_workgroup_start_us": {"median_over_cus": round(float(np.median([min(tb[i] for i in idx) for idx in groups.values()])), 2),
                                     "max_over_cus": round(float(max(min(tb[i] for i in idx) for idx in groups.values())), 2)},
    }
    res["share_of_cu_time"]["between"] = round(1 - sum(res["share_of_cu_time"].values()), 4)
    # resident workgroups (started, not finished) and workgroups inside their K loop, per CU, over 20 slices of the span
    edges = np.linspace(0, span, 21)
    mid = (edges[:-1] + edges[1:]) / 2
    res["per_cu_over_time"] = {"t_us": [round(float(m), 1) for m in mid],
                               "resident": [round(float(((tb <= m) & (te > m)).sum()) / ncu, 2) for m in mid],
                               "in_k_loop": [round(float(((tl0 <= m) & (tl1 > m)).sum()) / ncu, 2) for m in mid]}
    # per-CU workgroup counts: the quantisation the dispatcher produced
    cnt = np.array([len(idx) for idx in groups.values()])
    res["workgroups_per_cu_distribution"] = {int(c): int((cnt == c).sum()) for c in np.unique(cnt)}
    fin = np.array([max(te[i] for i in idx) for idx in groups.values()])
    res["cu_finish_us"] = {"min": round(float(fin.min()), 1), "median": round(float(np.median(fin)), 1), "max": round(float(fin.max()), 1)}
    print(json.dumps(res))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
