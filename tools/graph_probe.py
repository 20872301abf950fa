"""Dev tool: does replaying the executor's launches from a captured HIP graph shorten the step? (GPU-side dependency gaps
between ~550 small launches are ~3 us each; a graph removes host launch cost but keeps the barriers.)
usage: python tools/graph_probe.py [B]"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import ResNet50, tools

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = tools.set_device_gpu(0)
torch.manual_seed(0)
model = tools.device(ResNet50(30, 30, False))
model.train()
x = torch.rand(B, 3, 224, 224, device=dev)
dl = torch.randn(B, 30, device=dev) * 1e-3


def fwd_bwd():
    lg, _ = model._run_forward(x, True)
    model._run_backward(dl, None)
    return lg


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    e_f = timeit(lambda: model._run_forward(x, False))
    e_fb = timeit(fwd_bwd)
    print(f"eager : forward {e_f:.3f} ms   forward+backward {e_fb:.3f} ms", flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        model._run_forward(x, False)
    g_f = timeit(g1.replay)
    print(f"graph : forward {g_f:.3f} ms", flush=True)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        fwd_bwd()
    g_fb = timeit(g2.replay)
    print(f"graph : forward+backward {g_fb:.3f} ms", flush=True)
