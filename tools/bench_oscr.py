"""Dev tool: OSCR curve (next-row f4) — GPU time through the C ABI vs the reference's loop form on the host.
usage: python tools/bench_oscr.py [N] [C]"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import numpy as np
import torch
from openset_imagenet.util import calculate_oscr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 116


def reference_loop(gt, scores, unk_label=-1):
    """port of the reference's loop form (util.py:101-122), the CPU baseline of this row"""
    kn, unk = gt >= 0, gt == unk_label
    total_kn, total_unk = kn.sum(), unk.sum()
    pred, mx = scores.argmax(1), scores.max(1)
    tgt = scores[kn][np.arange(kn.sum()), gt[kn]]
    ccr, fpr = [], []
    for tau in np.unique(tgt)[:-1]:
        ccr.append(((pred[kn] == gt[kn]) & (tgt > tau)).sum() / total_kn)
        fpr.append((unk & (mx > tau)).sum() / total_unk)
    return np.array(ccr), np.array(fpr)


rng = np.random.default_rng(0)
z = rng.normal(size=(N, C)) * 3
s = np.exp(z - z.max(1, keepdims=True)); s = (s / s.sum(1, keepdims=True)).astype(np.float32)
gt = rng.integers(0, C, size=N); gt[rng.random(N) < 0.4] = -1
sd, gd = torch.from_numpy(s).cuda(), torch.from_numpy(gt).cuda()
calculate_oscr(gd, sd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ccr, fpr = calculate_oscr(gd, sd)
gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
t0 = time.perf_counter()
rc, rf = reference_loop(gt, s)
cpu_s = time.perf_counter() - t0
assert np.array_equal(ccr, rc) and np.array_equal(fpr, rf)
print(f"OSCR N={N} C={C}: {len(ccr)} points, identical to the host loop; GPU (device-resident scores, incl. result copy) "
      f"{gpu_ms:.2f} ms, host loop {cpu_s:.2f} s -> x{cpu_s * 1e3 / gpu_ms:.0f}")
