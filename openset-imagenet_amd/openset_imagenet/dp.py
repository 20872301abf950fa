"""Data-parallel training: one process per GPU, gradients averaged with RCCL all-reduce over xGMI, overlapped with backward.

The reference has no working data-parallel path (DistributedDataParallel is only imported and unwrapped,
openset_imagenet/train.py:10,49-50,79-87; the `dist:` block of config/train.yaml:35-39 is unused). The intent recorded there
("the batch size is multiplied by the number of gpus", train.yaml:18; "Log only on first process", train.py:248) is what this
module implements, with plain DDP semantics: per-rank BatchNorm statistics, average of per-rank gradients.

Design for point-to-point xGMI (7 links x ~153 GB/s per GPU, no switch):
  * the gradient arena is ONE contiguous fp32 buffer laid out in forward order; backward finishes it back to front in
    4 stages (head+layer4 = 15.2 M floats, layer3 = 7.1 M, layer2 = 1.2 M, layer1+stem = 0.2 M), so each bucket is one
    contiguous slice — no gather/scatter copies, 4 collectives per step instead of 162;
  * `bucket_ready` issues the all-reduce of the slice as soon as its stage has been enqueued, from a communication stream that
    waits for exactly the kernels that wrote the slice (the compute stream's progress and the executor's weight-gradient side
    stream — osi_resnet50_grads_ready); the compute stream itself never waits between stages (round 4: the three per-stage joins
    of the side stream are gone) and runs the next stage while the ring is on the wire; `finish()` makes it wait for all four
    collectives before the optimizer step.
On gloo (CPU tests) the same code averages with SUM + divide; on nccl (= RCCL) it uses ReduceOp.AVG in place.
"""
import os

import torch
import torch.distributed as dist
from torch import nn


def reserved_cus_for_channels(channels, slots_per_cu=8, cap=32):
    """CUs' worth of wave slots the launch plans should leave to RCCL: a collective keeps one 256-thread workgroup per channel resident,
    and a CU holds eight such workgroups of the convolution kernels (8 waves per SIMD) — so `channels` workgroups displace
    ceil(channels / 8) CUs of them while a ring is on the wire. Fed to osi_set_tuning("dp_reserved_cus") before the executor is created."""
    if not channels or channels <= 0:
        return 0
    return min(cap, -(-int(channels) // slots_per_cu))


class GradSync:
    """Bucketed asynchronous gradient averaging over a flat arena.

    Device-resident buckets are reduced from a communication stream this object owns: the model hands each finished backward stage
    to it (`handoff`), the collective is issued there in its synchronous form (ProcessGroupNCCL runs the ring on its internal stream
    and makes the ISSUING stream — the communication stream — wait for it), and `finish()` makes the compute stream wait for the
    communication stream once, in front of the optimizer. `timing(True)` (bench.py's comm leg, never the timed region) brackets every
    collective with two HIP events on the communication stream — the first is recorded behind the hand-off, so the bracket is the
    collective's own duration — and measures how long the compute stream still waited in `finish()`: the exposed part.
    `read_timing()` returns the per-step means."""

    def __init__(self, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        self._work = []
        self._timing = False
        self._comm_stream = None
        self._ev = []            # per step: ([(start, end) per bucket], exposed_start, exposed_end)
        self._cur = None
        self._host = []          # gloo: host-side seconds per step (collectives complete on the host)
        self._on_comm = False    # collectives of this step were issued from the communication stream

    def timing(self, on):
        self._timing = bool(on)
        self._ev, self._host, self._cur = [], [], None

    def _comm(self, device):
        if self._comm_stream is None:
            # high priority: its own hardware queue set (the compute stream is normal, the weight-gradient side stream low priority). On a
            # queue shared with the compute stream, this stream's wait for the side stream would hold back the compute stream's next kernels
            prio = int(os.environ.get("OSI_DP_COMM_PRIO", "-1"))
            self._comm_stream = torch.cuda.Stream(device=device, priority=prio)
        return self._comm_stream

    def bucket_ready(self, flat, lo, hi, handoff=None):
        """Average flat[lo:hi] across the ranks, asynchronously to the calling (compute) stream.

        `handoff(comm_stream)` — given by the model for device-resident gradients — makes the communication stream wait for the
        bucket's producers (the compute stream's progress AND the executor's side stream, torch.ops.osi.resnet50_grads_ready) without
        making the compute stream wait for anything: the next backward stage is enqueued behind the previous one with no bubble, the
        collective is ordered behind exactly the kernels that wrote the bucket. Without it (CPU tensors, stand-alone use) the
        collective is ordered behind the current stream as torch.distributed does by default."""
        if self.world == 1 or hi <= lo:
            return
        bucket = flat[lo:hi]
        if bucket.is_cuda and (handoff is not None or self._timing):
            comm = self._comm(bucket.device)
            if handoff is not None:
                handoff(comm)
            else:
                comm.wait_stream(torch.cuda.current_stream(bucket.device))
            ev = None
            with torch.cuda.stream(comm):
                if self._timing:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(comm)
                # synchronous form ON THE COMMUNICATION STREAM: that stream (not the host, not the compute stream) waits for the ring
                if self.backend == "nccl":
                    dist.all_reduce(bucket, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
                    bucket.div_(self.world)
                if ev is not None:
                    ev[1].record(comm)
            if ev is not None:
                if self._cur is None:
                    self._cur = []
                self._cur.append(ev)
            self._on_comm = True
            return
        if self.backend == "nccl":
            w = dist.all_reduce(bucket, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self._work.append((w, None))
        else:
            w = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._work.append((w, bucket))

    def finish(self):
        """The calling stream waits for every collective issued since the last finish() (the optimizer step follows)."""
        if self._on_comm:
            cur = torch.cuda.current_stream(self._comm_stream.device)
            timed = self._timing and self._cur is not None
            if timed:
                x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x0.record(cur)
            cur.wait_stream(self._comm_stream)
            if timed:
                x1.record(cur)
                self._ev.append((self._cur, x0, x1))
                self._cur = None
            self._on_comm = False
        t0 = None
        if self._timing and self._work:
            import time
            t0 = time.perf_counter()
        for w, bucket in self._work:
            w.wait()  # nccl: the current stream waits for the process group's stream; gloo: host wait
            if bucket is not None:
                bucket.div_(self.world)
        if t0 is not None:
            import time
            self._host.append(time.perf_counter() - t0)
        self._work = []

    def read_timing(self):
        """{"steps", "comm_ms_per_step", "per_bucket_ms", "exposed_comm_ms"} over the instrumented steps since timing(True);
        synchronises on the recorded events. gloo: only the exposed (host wait) time exists."""
        if self._ev:
            self._ev[-1][2].synchronize()
            nb = len(self._ev[0][0])
            per = [sum(st[0][i][0].elapsed_time(st[0][i][1]) for st in self._ev) / len(self._ev) for i in range(nb)]
            exposed = sum(st[1].elapsed_time(st[2]) for st in self._ev) / len(self._ev)
            return {"steps": len(self._ev), "comm_ms_per_step": sum(per), "per_bucket_ms": per, "exposed_comm_ms": exposed}
        if self._host:
            return {"steps": len(self._host), "comm_ms_per_step": None, "per_bucket_ms": None,
                    "exposed_comm_ms": 1e3 * sum(self._host) / len(self._host)}
        return {"steps": 0, "comm_ms_per_step": None, "per_bucket_ms": None, "exposed_comm_ms": None}


class DistributedDataParallel(nn.Module):
    """Wrapper with the surface the reference expects from DDP (`.module`, train.py:49-50): forwards calls to the wrapped
    MI355X ResNet50 and hooks GradSync into its staged backward. Parameters and BN buffers are broadcast from rank 0."""

    def __init__(self, module, process_group=None, broadcast_buffers=True):
        super().__init__()
        self.module = module
        self.sync = GradSync(process_group)
        module._grad_sync = self.sync
        if self.sync.world > 1:
            dist.broadcast(module._flat_params, src=0, group=process_group)
            if broadcast_buffers:
                dist.broadcast(module._flat_buffers, src=0, group=process_group)
                dist.broadcast(module._nbt, src=0, group=process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def average_flat_gradients(flat, buckets, process_group=None):
    """Stand-alone use of the bucket schedule (used by the CPU/gloo tests): average `flat` across ranks bucket by bucket."""
    sync = GradSync(process_group)
    for lo, hi in buckets:
        sync.bucket_ready(flat, lo, hi)
    sync.finish()
    return flat
