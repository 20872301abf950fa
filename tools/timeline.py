"""Dev tool (GPU box): where the overlapped backward spends its time. Runs the bench workload with the executor's timeline
instrumentation (side-stream overlap kept; one completion event per op, tagged with its stream) and prints, per step, when the
critical-path ops (main stream) and the weight gradients (side stream) finish.   usage: python tools/timeline.py [workload] [B]"""
import ctypes, json, os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
import torch
from openset_imagenet import ResNet50, EntropicOpensetLoss, optim, tools, _native as N

C, B = 30, int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = tools.set_device_gpu(0)
torch.manual_seed(42)
model = tools.device(ResNet50(C, C, False)); opt = optim.Adam(model.parameters(), lr=1e-3); loss = EntropicOpensetLoss(C, 1.0)
x = torch.rand(B, 3, 224, 224, device=dev); y = torch.randint(-1, C, (B,), device=dev)
def step():
    model.train(); opt.zero_grad(); lg, _ = model(x); j = loss(lg, y); j.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
h = model._net(B, 224, 224).h
L = N.lib()
N.check(L.osi_resnet50_profile(h, 2))
step(); torch.cuda.synchronize()
cap = 4096
t = (ctypes.c_double * cap)(); cls = (ctypes.c_int * cap)(); side = (ctypes.c_int * cap)(); n = ctypes.c_int()
N.check(L.osi_resnet50_timeline_read(h, t, cls, side, cap, ctypes.byref(n)))
N.check(L.osi_resnet50_profile(h, 0))
names = ["start", "conv_fwd", "conv_dgrad", "conv_wgrad", "bn_fwd", "bn_bwd", "other"]
ev = [(t[i], names[cls[i]], side[i]) for i in range(n.value)]
starts = [i for i, e in enumerate(ev) if e[1] == "start"]
bwd0 = starts[1] if len(starts) > 1 else 0
print(f"events {n.value}; forward ends at {ev[bwd0][0]:.2f} ms; step (fwd+bwd) ends at {max(e[0] for e in ev):.2f} ms")
main_end = max(e[0] for e in ev[bwd0:] if not e[2]); side_end = max([e[0] for e in ev[bwd0:] if e[2]] or [0])
print(f"backward: main stream finishes at {main_end:.2f} ms, side stream (weight gradients) at {side_end:.2f} ms")
# lag of each wgrad behind the dgrad issued right after it
k = 0
for i in range(bwd0, n.value):
    tt, nm, sd = ev[i]
    if nm in ("conv_dgrad", "conv_wgrad"):
        k += 1
        if k % 8 == 0 or sd:
            pass
rows = [(round(tt, 2), nm, "side" if sd else "main") for tt, nm, sd in ev[bwd0:] if nm in ("conv_dgrad", "conv_wgrad")]
if os.environ.get("OSI_TIMELINE_FULL"):
    print(json.dumps(rows))
sname = {0: "main", 1: "side"}
print("first forward ops:", [(round(tt, 3), nm, sname[sd]) for tt, nm, sd in ev[:14]])
print("last ops of the step:", [(round(tt, 3), nm, sname[sd]) for tt, nm, sd in ev[-16:]])
