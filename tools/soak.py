"""Dev tool (GPU box): soak test of the overlapped step. Runs the bench workload for N steps twice from the same seeds (fresh model
objects) and compares every parameter, BN buffer, optimizer moment and loss bit for bit; a race between the executor's streams
(a scratch buffer recycled too early, a missing event) shows up as a difference.  usage: python tools/soak.py [steps] [B]"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import ResNet50, EntropicOpensetLoss, optim, tools

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
DP = len(sys.argv) > 3 and sys.argv[3] == "dp"
dev = tools.set_device_gpu(0)
C = 30


def run(dp=False):
    torch.manual_seed(1)
    model = tools.device(ResNet50(C, C, False)); opt = optim.Adam(model.parameters(), lr=1e-3); loss = EntropicOpensetLoss(C, 1.0)
    net = model
    if dp:
        from openset_imagenet.dp import DistributedDataParallel
        net = DistributedDataParallel(model)
        net.sync.world = 2          # force the collectives at world size 1
    g = torch.Generator(device=dev).manual_seed(2)
    xs = [torch.rand(B, 3, 224, 224, device=dev, generator=g) for _ in range(4)]
    ys = [torch.randint(-1, C, (B,), device=dev, generator=g) for _ in range(4)]
    losses = []
    t0 = time.perf_counter()
    for i in range(steps):
        model.train(); opt.zero_grad()
        lg, _ = net(xs[i % 4]); j = loss(lg, ys[i % 4]); j.backward(); opt.step()
        losses.append(j.detach())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return (model.flat_parameters().clone(), model._flat_buffers.clone(), opt._flat_state["exp_avg"].clone(), opt._flat_state["exp_avg_sq"].clone(),
            torch.stack(losses)), dt

if DP:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
a, ta = run()
b, tb = run(DP)
ok = all(torch.equal(u, v) and bool(torch.isfinite(u).all()) for u, v in zip(a, b))
print(f"steps {steps} B {B}: run A {steps * B / ta:.0f} img/s, run B {steps * B / tb:.0f} img/s; loss {float(a[4][0]):.4f} -> {float(a[4][-1]):.4f}; "
      f"bitwise identical and finite: {ok}")
sys.exit(0 if ok and float(a[4][-1]) < float(a[4][0]) else 1)
