import os, sys, glob
os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_FILE="/tmp/osi_probe_rccl.log", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(1 << 20, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
dist.destroy_process_group()
for f in glob.glob("/tmp/osi_probe_rccl*"):
    txt = open(f, errors="replace").read()
    print(f, len(txt)); print("\n".join(l for l in txt.splitlines() if "hannel" in l or "nranks" in l)[:3000])
