"""Dev tool (GPU box): what HBM gives a pure read, a pure write and a copy stream on this device (torch kernels, 1 GiB tensors — four
times the Infinity Cache), next to the rates the step's own streaming kernels reach.   python tools/hbm_rates.py"""
import torch

n = 256 * 1024 * 1024          # fp32 elements = 1 GiB
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda"); z = torch.empty(n, device="cuda")
x.uniform_(); y.uniform_()


def t(fn, nbytes, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return nbytes / ms / 1e9, ms


for name, fn, nb in (("write only  (fill_)", lambda: z.fill_(1.0), 4 * n),
                     ("read only   (sum)", lambda: x.sum(), 4 * n),
                     ("copy        (1 read : 1 write)", lambda: z.copy_(x), 8 * n),
                     ("add         (2 reads : 1 write)", lambda: torch.add(x, y, out=z), 12 * n),
                     ("in-place mul (1 read : 1 write, same lines)", lambda: z.mul_(1.0001), 8 * n)):
    r, ms = t(fn, nb)
    print(f"{name:46s} {r:6.2f} TB/s  ({ms:.3f} ms)")
