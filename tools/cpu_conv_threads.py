"""Dev (GPU box): how torch-CPU fp64 convolutions (the references of the -m gpu parity tests) scale with the intra-op thread count on a GPU lease."""
import os, time, torch, torch.nn.functional as F
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
x = torch.randn(128, 64, 56, 56, dtype=torch.float64); w = torch.randn(64, 64, 3, 3, dtype=torch.float64)
x1 = torch.randn(128, 256, 56, 56, dtype=torch.float64); w1 = torch.randn(64, 256, 1, 1, dtype=torch.float64)
g = torch.randn(128, 64, 56, 56, dtype=torch.float64)
for n in (torch.get_num_threads(), 64, 32, 16, 8):
    torch.set_num_threads(n)
    t0 = time.time(); F.conv2d(x, w, None, 1, 1); t1 = time.time(); F.conv2d(x1, w1); t2 = time.time()
    torch.nn.grad.conv2d_input(x.shape, w, g, 1, 1); t3 = time.time()
    torch.nn.grad.conv2d_weight(x, w.shape, g, 1, 1); t4 = time.time()
    a = x1.permute(0, 2, 3, 1).contiguous(); t5 = time.time()
    print(f"threads {n:3d}: conv3x3 {t1-t0:.2f}s conv1x1 {t2-t1:.2f}s dgrad {t3-t2:.2f}s wgrad {t4-t3:.2f}s permute {t5-t4:.2f}s")
