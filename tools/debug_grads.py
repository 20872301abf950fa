"""Dev diagnostic: per-parameter gradient error of the HIP path vs the fp64 oracle (and torch-cpu-fp32 for scale)."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import ResNet50, EntropicOpensetLoss
from oracle import resnet50_oracle as R, losses_oracle as L

B, HW, C, seed = (int(a) for a in (sys.argv[1:5] + ["4", "64", "8", "21"][len(sys.argv) - 1:]))
gen = torch.Generator().manual_seed(seed)
sd = R.init_state(C, C, False, generator=gen)
model = ResNet50(C, C, False); model.load_state_dict(sd); model = model.cuda()
x = torch.rand(B, 3, HW, HW, generator=gen); y = torch.randint(-1, C, (B,), generator=gen)
model.train(); lg, ft = model(x.cuda()); j = EntropicOpensetLoss(C)(lg, y.cuda()); j.backward()
fn = lambda lg, t, f: L.entropic_openset_loss(lg, t, 1.0)
r32 = R.forward_backward({k: v.clone() for k, v in sd.items()}, x, y, fn)
r64 = R.forward_backward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double(), y, fn)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
print("logit err", float((lg.cpu().double() - r64[0]).abs().max()), "torch32", float((r32[0].double() - r64[0]).abs().max()))
named = dict(model.named_parameters())
for k in R.param_keys(sd):
    a, b = rel(named[k].grad.cpu(), r64[3][k]), rel(r32[3][k], r64[3][k])
    flag = " <<<<" if a > max(10 * b, 2e-4) else ""
    print(f"{k:50s} mine {a:.2e}  torch32 {b:.2e}{flag}")
