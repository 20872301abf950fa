"""CPU, world_size 2 and 4 over gloo: the bucketed gradient averaging of dp.py equals the single-process mean, bucket by bucket, on
gradients computed by the oracle from different per-rank batches; parameters and BN buffers are broadcast from rank 0."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path[:0] = [root, os.path.join(root, "openset-imagenet_amd")]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import openset_imagenet as oi
        from openset_imagenet.dp import DistributedDataParallel, GradSync
        from oracle import resnet50_oracle as R, losses_oracle as L
        torch.manual_seed(100 + rank)          # different initial weights per rank: the wrapper must broadcast rank 0's
        model = oi.ResNet50(6, 6, False)
        ddp = DistributedDataParallel(model)
        assert ddp.module is model and model._grad_sync is ddp.sync
        flat0 = model.flat_parameters().clone()
        gathered = [torch.zeros_like(flat0) for _ in range(world)]
        dist.all_gather(gathered, flat0)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "parameters not broadcast"
        # per-rank gradients from the oracle on per-rank batches (plain DDP semantics: per-rank BN statistics)
        g = torch.Generator().manual_seed(7 + rank)
        x = torch.rand(2, 3, 32, 32, generator=g); y = torch.randint(-1, 6, (2,), generator=g)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        _, _, _, grads = R.forward_backward(sd, x, y, lambda lg, t, f: L.entropic_openset_loss(lg, t, 1.0))
        model.bind_gradients()
        named = dict(model.named_parameters())
        with torch.no_grad():
            for k, v in grads.items():
                named[k].grad.copy_(v)
        mine = model.flat_gradients().clone()
        allg = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allg, mine)
        expect = sum(allg) / world
        # world 2: a + b has one rounding whatever the order — the average matches to the last bit; more ranks: the collective's
        # summation order differs from Python's, each side makes (world - 1) roundings of at most 2^-24 of the sum of magnitudes
        bound = (world - 2) * 2.0 ** -22 * sum(a.abs() for a in allg) / world + 1e-7
        close = lambda got: bool(((got - expect).abs() <= bound).all())
        # the schedule the executor drives: one bucket per backward stage, head first
        buckets = model.gradient_buckets()
        for lo, hi in buckets:
            ddp.sync.bucket_ready(model.flat_gradients(), lo, hi)
        ddp.sync.finish()
        assert close(model.flat_gradients())
        assert sorted(buckets)[0][0] == 0 and max(h for _, h in buckets) == mine.numel()
        # instrumented form of the same schedule (bench.py's comm leg): same averages, and the timing record has its fields
        with torch.no_grad():
            model.flat_gradients().copy_(mine)
        ddp.sync.timing(True)
        for lo, hi in buckets:
            ddp.sync.bucket_ready(model.flat_gradients(), lo, hi)
        ddp.sync.finish()
        t = ddp.sync.read_timing()
        ddp.sync.timing(False)
        assert close(model.flat_gradients())
        assert t["steps"] == 1 and t["exposed_comm_ms"] is not None and t["exposed_comm_ms"] >= 0 and t["comm_ms_per_step"] is None
        out.put((rank, float(((model.flat_gradients() - expect).abs() - bound).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_gradient_average(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(540)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(out.get(timeout=5) for _ in range(world))
    assert [r for r, _ in res] == list(range(world)) and all(e <= 0 for _, e in res)
