"""Dev probe: how far do three SGD steps at B = 8 diverge between implementations that differ only in fp32 summation order?
Runs the loop of tests/test_model_gpu.py::test_sgd_training_steps_follow_oracle under several tail-split plans (each a different,
equally valid summation order) next to the fp32 / fp64 CPU oracle and prints the loss of every step."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import ResNet50, EntropicOpensetLoss, optim, _native as N
from oracle import resnet50_oracle as R, losses_oracle as L

C, lr = 10, 1e-3
cuda = torch.device("cuda:0")


def run(seed, cfg):
    for k, v in cfg.items():
        N.check(N.lib().osi_set_tuning(k.encode(), v))
    gen = torch.Generator().manual_seed(seed)
    sd = R.init_state(C, C, False, generator=gen)
    model = ResNet50(C, C, False); model.load_state_dict(sd); model = model.to(cuda)
    opt = optim.SGD(model.parameters(), lr=lr, momentum=0.9)
    out = []
    for step in range(3):
        x = torch.rand(8, 3, 96, 96, generator=gen); y = torch.randint(-1, C, (8,), generator=gen)
        model.train(); opt.zero_grad()
        logits, _ = model(x.to(cuda))
        j = EntropicOpensetLoss(C, 1.0)(logits, y.to(cuda)); j.backward(); opt.step()
        out.append(float(j.detach()))
    return out


def oracle(seed, dt):
    gen = torch.Generator().manual_seed(seed)
    sd = R.init_state(C, C, False, generator=gen)
    sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    st, out = {}, []
    for step in range(3):
        x = torch.rand(8, 3, 96, 96, generator=gen); y = torch.randint(-1, C, (8,), generator=gen)
        r = R.forward_backward(sd, x.to(dt), y, lambda lg, t, f: L.entropic_openset_loss(lg, t, 1.0))
        R.sgd_step(sd, r[3], st, lr=lr, momentum=0.9)
        out.append(float(r[2]))
    return out


for seed in (5, 6, 7):
    print(f"seed {seed}: fp64 oracle {['%.5f' % v for v in oracle(seed, torch.float64)]}  fp32 oracle {['%.5f' % v for v in oracle(seed, torch.float32)]}")
    for cfg in ({"tail_split": 0}, {"tail_split": 1, "tail_cus": 0}, {"tail_split": 1, "tail_cus": 128}, {"tail_split": 1, "tail_cus": 64},
                {"tail_split": 1, "tail_cus": 24}, {"tail_split": 0, "wgrad_blocks": 1024}, {"tail_split": 0, "wgrad_blocks": 4096}):
        print("   ", cfg, ["%.5f" % v for v in run(seed, cfg)])
    N.check(N.lib().osi_set_tuning(b"wgrad_blocks", 2048)); N.check(N.lib().osi_set_tuning(b"tail_cus", 0)); N.check(N.lib().osi_set_tuning(b"tail_split", 1))
