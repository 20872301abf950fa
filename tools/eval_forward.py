#!/usr/bin/env python3
"""Dev (GPU box): the inference forward of validate() / get_arrays() (reference train.py:142-234) at the benchmarked batch, one mode per
process so that a kernel trace holds that mode only.
    python3 tools/eval_forward.py [fused|topology] [steps] [batch] [C]
    rocprofv3 --kernel-trace --stats -d gpurun_out/eval_fused -o s --output-format csv -- python3 tools/eval_forward.py fused 20
fused = the inference forms (default of the executor: every BatchNorm + shortcut + ReLU in its convolution's epilogue); topology = the
training topology on running statistics (executor option eval_fused = 0). Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch  # noqa: E402

from openset_imagenet import ResNet50, tools, _native as N  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
C = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = tools.set_device_gpu(0)
torch.manual_seed(42)
model = tools.device(ResNet50(C, C, False))
x = torch.rand(B, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
h = model._net(B, 224, 224).h
N.check(N.lib().osi_resnet50_set_option(h, b"eval_fused", 1 if mode == "fused" else 0))
model.eval()
with torch.no_grad():
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lg, _ = model(x)
    torch.cuda.synchronize()
t = (time.perf_counter() - t0) / steps
print(json.dumps({"mode": mode, "batch": B, "classes": C, "steps": steps, "ms_per_batch": round(t * 1e3, 3), "images_per_sec": round(B / t, 1),
                  "logit_checksum": float(lg.double().sum())}))
