"""Worker of tests/test_dp_two_ranks_gpu.py: one of two ranks that share GPU 0 and talk over gloo (RCCL refuses two ranks on
one device, and an 8-GPU node is not ours to launch on). Exercises the real data-parallel step: broadcast at wrap time, staged
backward of the HIP executor with one asynchronous bucket all-reduce per stage, averaged gradients, identical parameters after
an optimizer step. Exit code 0 = every check passed."""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openset_imagenet import ResNet50, EntropicOpensetLoss, optim, tools
        from openset_imagenet.dp import DistributedDataParallel
        dev = tools.set_device_gpu(0)
        C = 6
        torch.manual_seed(100 + rank)                       # different initial weights per rank: the wrapper must broadcast
        model = tools.device(ResNet50(C, C, False))
        ddp = DistributedDataParallel(model)

        def same_everywhere(t, what):
            ts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(ts, t.contiguous())
            assert all(torch.equal(a, ts[0]) for a in ts), f"{what} differ between ranks"

        same_everywhere(model.flat_parameters(), "parameters after broadcast")
        same_everywhere(model._flat_buffers, "BN buffers after broadcast")
        g = torch.Generator().manual_seed(7 + rank)         # per-rank batch
        x = torch.rand(4, 3, 64, 64, generator=g).to(dev)
        y = torch.randint(-1, C, (4,), generator=g).to(dev)
        loss = EntropicOpensetLoss(C, 1.0)

        def grads(sync):
            model._grad_sync = sync
            model.train()
            lg, _ = ddp(x)
            loss(lg, y).backward()
            torch.cuda.synchronize()
            return model.flat_gradients().clone()

        local = grads(None)                                  # this rank's own gradient (single-GPU path: one backward call)
        synced = grads(ddp.sync)                             # staged backward + one async all-reduce per stage
        every = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(every, local)
        assert not torch.equal(every[0], every[1]), "ranks saw different batches, their gradients must differ"
        expect = (every[0] + every[1]) / world
        assert torch.equal(synced, expect), f"bucketed average differs from the mean of the per-rank gradients: {float((synced - expect).abs().max()):.3e}"
        same_everywhere(synced, "averaged gradients")
        # an optimizer step keeps the replicas identical
        opt = optim.Adam(model.parameters(), lr=1e-3)
        opt.step()
        torch.cuda.synchronize()
        same_everywhere(model.flat_parameters(), "parameters after the optimizer step")
        print(f"rank {rank}: ok, |g| = {float(synced.norm()):.4e}", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
