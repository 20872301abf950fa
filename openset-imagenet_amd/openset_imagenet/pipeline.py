"""Input pipeline of the training loop, split between the host workers and the GPU.

The reference feeds `train()` from `DataLoader(ImagenetDataset(csv, imagenet_path, transform), batch_size, shuffle, num_workers=4,
pin_memory=True)` (openset_imagenet/train.py:299-311, dataset.py:33-54) with

    train: Compose([Resize(256), RandomCrop(224), RandomHorizontalFlip(0.5), ToTensor()])     train.py:259-263
    val:   Compose([Resize(256), CenterCrop(224), ToTensor()])                                train.py:265-268

and copies every fp32 batch with a blocking `device(images)` inside the step (train.py:128). At ~4 000 images/s per MI355X that
loader is the bottleneck, so the work is re-cut:

  host workers (CanvasDataset)    JPEG decode + Resize(256) only; the sample is handed over UNCROPPED as a fixed-size uint8
                                  canvas (a 256x256 window of the resized image that contains the crop), plus the crop corner
                                  inside the canvas and the flip flag: 196 KiB per image instead of 588 KiB of fp32, no PIL crop /
                                  transpose, no float work on the CPU
  copy stream (DevicePrefetcher)  pinned batch -> HBM with a non-blocking copy, then ONE kernel (osi_u8_crop_flip_to_nhwc4) does
                                  RandomCrop/CenterCrop + RandomHorizontalFlip + ToTensor (value / 255) + the NHWC4 layout the
                                  stem convolution reads; all of it one batch ahead of the compute stream, ordered by events
  compute stream (train())        `model(images)` binds the staged batch in place; no copy, no host synchronisation in the step

torchvision is not a dependency of this build (not installed offline): Resize / crop / flip semantics are restated below next to
the torchvision rule they follow, and tests/test_input_pipeline_gpu.py checks the device result bit for bit against PIL's crop /
transpose + the ToTensor division on the host.
"""
import pathlib
import random

import numpy as np
import torch

from . import _native as N

CANVAS = 256   # Resize(256): the short side of every resized image
CROP = 224


def resize_size(w, h, size=CANVAS):
    """Output (w, h) of torchvision Resize(size) for an int size: short side -> size, long side -> int(size * long / short)."""
    if w <= h:
        return size, int(size * h / w)
    return int(size * w / h), size


def center_crop_corner(w, h, crop=CROP):
    """Top-left corner of torchvision CenterCrop(crop): int(round((dim - crop) / 2.0)) per axis (Python rounding)."""
    return int(round((w - crop) / 2.0)), int(round((h - crop) / 2.0))


def canvas_window(arr, x0, y0, canvas=CANVAS, crop=CROP):
    """A canvas x canvas window of the resized image `arr` [h, w, 3] that contains the crop whose corner is (x0, y0); returns
    (window, cx, cy) with (cx, cy) the crop corner inside the window. The short side of `arr` equals `canvas`, so the window
    always fits; images smaller than the canvas (never produced by Resize(256)) are zero padded."""
    h, w = arr.shape[:2]
    wx = max(0, min(x0, w - canvas))
    wy = max(0, min(y0, h - canvas))
    win = arr[wy:wy + canvas, wx:wx + canvas]
    if win.shape[0] != canvas or win.shape[1] != canvas:
        pad = np.zeros((canvas, canvas, 3), dtype=np.uint8)
        pad[:win.shape[0], :win.shape[1]] = win
        win = pad
    return win, x0 - wx, y0 - wy


class CanvasDataset(torch.utils.data.Dataset):
    """ImagenetDataset of the reference (dataset.py:10-54: CSV with `path,label` rows, PIL RGB decode) with the transform cut
    after Resize(256). Samples: (canvas uint8 [256,256,3], crop_xy int32 [2], flip uint8 [], label int64 []); with
    `uint8=False` the reference's own sample is produced instead: (fp32 [3,224,224] in [0,1], label) — host crop, flip, ToTensor."""

    def __init__(self, csv_file, imagenet_path, train, loss_type, uint8=True):
        import pandas as pd
        from .dataset import LabelTable
        self.frame = pd.read_csv(csv_file, header=None)
        self.table = LabelTable(self.frame[1].to_numpy())
        if loss_type == "garbage":                         # train.py:287-290
            self.table.replace_negative_label()
            self.frame[1] = self.table.labels
        elif loss_type == "softmax" and train:             # train.py:291-293
            self.frame = self.frame[self.frame[1] >= 0].reset_index(drop=True)
            self.table.remove_negative_label()
        self.root = pathlib.Path(imagenet_path)
        self.train, self.uint8 = bool(train), bool(uint8)

    def __len__(self):
        return len(self.frame)

    def __getitem__(self, i):
        from PIL import Image
        path, label = self.frame.iloc[i]
        img = Image.open(self.root / path).convert("RGB")
        img = img.resize(resize_size(*img.size), Image.BILINEAR)                              # Resize(256)
        w, h = img.size
        if self.train:                                                                        # RandomCrop(224), flip(0.5)
            x0, y0 = random.randint(0, w - CROP), random.randint(0, h - CROP)
            flip = random.random() < 0.5
        else:                                                                                 # CenterCrop(224)
            (x0, y0), flip = center_crop_corner(w, h), False
        label = torch.as_tensor(int(label), dtype=torch.int64)
        arr = np.asarray(img, dtype=np.uint8)
        if self.uint8:
            win, cx, cy = canvas_window(arr, x0, y0)
            return (torch.from_numpy(np.array(win)), torch.tensor([cx, cy], dtype=torch.int32),   # np.array: a writable, contiguous copy
                    torch.tensor(1 if flip else 0, dtype=torch.uint8), label)
        x = arr[y0:y0 + CROP, x0:x0 + CROP]
        if flip:
            x = x[:, ::-1]
        return torch.from_numpy(np.array(x)).permute(2, 0, 1).float().div_(255.0), label   # ToTensor() on the host


def worker_init(worker_id=0, niceness=0):
    """DataLoader `worker_init_fn` of worker(): every decode worker keeps to ONE intra-op thread (a worker process otherwise inherits
    torch's default pool size — the host's CPU count — for its tensor ops, on a lease of 16 CPUs per GPU that it shares with 15 other
    workers and the thread that launches the GPU work). `niceness` > 0 also lowers the workers' scheduling priority; measured on a
    16-CPU lease it changes nothing (profiles/NOTES_r05.md: the launch thread is not starved by the workers), so the default leaves it."""
    import os
    if niceness > 0:
        try:
            os.nice(niceness)
        except OSError:
            pass
    torch.set_num_threads(1)


def stage_canvas_batch(canvas, crop_xy, flip, out=None, crop=CROP, stream=None):
    """uint8 canvases [B,Hc,Wc,3] + crop corners int32 [B,2] + flip flags uint8 [B] (all on the GPU) -> fp32 [B,crop,crop,4]
    (NHWC4, the executor's input layout) in one kernel on `stream` (default: the current stream)."""
    if not canvas.is_cuda:
        raise RuntimeError("openset_imagenet (MI355X build): the crop / flip / ToTensor staging runs on the GPU")
    canvas = canvas.contiguous()
    B, Hc, Wc, _ = canvas.shape
    if out is None:
        out = torch.empty(B, crop, crop, 4, device=canvas.device, dtype=torch.float32)
    crop_xy = None if crop_xy is None else crop_xy.to(device=canvas.device, dtype=torch.int32).contiguous()
    flip = None if flip is None else flip.to(device=canvas.device, dtype=torch.uint8).contiguous()
    st = stream.cuda_stream if stream is not None else N.stream_of(canvas)
    N.check(N.lib().osi_u8_crop_flip_to_nhwc4(N.ptr(canvas), N.ptr(crop_xy), N.ptr(flip), N.ptr(out), B, Hc, Wc, crop, crop, st),
            "osi_u8_crop_flip_to_nhwc4")
    return out


def device_batch(batch):
    """(images, labels) on the global device from whatever a loader of this package yields: the reference's (images, labels) pair
    (reference train.py:128-129: `images = device(images); labels = device(labels)`; already-resident tensors pass through) or the
    4-tuple of CanvasDataset, staged to NHWC4 on the current stream."""
    from . import tools
    if len(batch) == 4:
        canvas, crop_xy, flip, labels = batch
        dev = tools.get_device()
        images = stage_canvas_batch(canvas.to(dev, non_blocking=True), crop_xy.to(dev, non_blocking=True), flip.to(dev, non_blocking=True))
        return images, tools.device(labels)
    images, labels = batch
    return tools.device(images), tools.device(labels)


class DevicePrefetcher:
    """Iterates a host DataLoader one batch ahead of the GPU: pinned-memory -> HBM copies and the crop / flip / ToTensor staging
    run on a dedicated copy stream while the compute stream trains on the previous batch; an event per batch orders the two.

    Host batches may be (canvas, crop_xy, flip, labels) from CanvasDataset — yielded as (fp32 [B,224,224,4], labels) — or the
    reference's plain (images fp32 [B,3,224,224], labels), which are only copied asynchronously. Yields device tensors, so the
    `device(images)` calls of train() (train.py:128-129) become no-ops and the step contains no host synchronisation."""

    RING = 3   # staged-batch buffers: one being consumed, one being filled, one spare

    def __init__(self, loader, device=None, crop=CROP, group=1):
        """`group` > 1: the host loader yields SUB-batches (batch_size / group samples each) and `group` consecutive ones are staged
        into one device batch — the same samples in the same order as one loader batch of the full size (a sampler's index stream
        is cut at multiples of the sub-batch size, so is the ragged tail), but a worker has its first result after 1/group of the
        time: the bubble at the start of every epoch, when all workers build their first batch at once and the GPU waits for a
        whole one (128 images x 1.6 ms = 205 ms, profiles/r05_input_pipeline.json), shrinks by that factor."""
        from . import tools
        self.loader = loader
        self.device = torch.device(device) if device is not None else tools.get_device()
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher needs the GPU device (set_device_gpu(index) first)")
        if int(group) < 1:
            raise ValueError("DevicePrefetcher: group must be >= 1")
        self.crop = crop
        self.group = int(group)
        self.stream = torch.cuda.Stream(self.device)
        self._ring = [None] * self.RING
        self._free = [None] * self.RING      # event: the compute stream is done with ring slot i

    def __len__(self):
        return -(-len(self.loader) // self.group)

    def _load(self, it, k):
        hosts = []
        for _ in range(self.group):
            try:
                hosts.append(next(it))
            except StopIteration:
                break
        if not hosts:
            return None
        slot = k % self.RING
        B = sum(int(h[-1].shape[0]) for h in hosts)
        nb = dict(device=self.device, non_blocking=True)
        with torch.cuda.stream(self.stream):
            if self._free[slot] is not None:
                self.stream.wait_event(self._free[slot])          # the step that read this slot has been fully enqueued and run
            if len(hosts[0]) == 4:
                buf = self._ring[slot]
                if buf is None or buf.shape[0] < B:
                    buf = self._ring[slot] = torch.empty(B, self.crop, self.crop, 4, device=self.device, dtype=torch.float32)
                o = 0
                for canvas, crop_xy, flip, _ in hosts:
                    b = canvas.shape[0]
                    stage_canvas_batch(canvas.to(**nb), crop_xy.to(**nb), flip.to(**nb), out=buf[o:o + b], crop=self.crop, stream=self.stream)
                    o += b
                images = buf[:B]
            elif len(hosts) == 1:
                images = hosts[0][0].to(**nb)
            else:
                images = torch.cat([h[0].to(**nb) for h in hosts])
            labels = hosts[0][-1].to(**nb) if len(hosts) == 1 else torch.cat([h[-1].to(**nb) for h in hosts])
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return images, labels, ready, hosts, slot     # `hosts` keeps the pinned sources alive until the copies have been issued and consumed

    def __iter__(self):
        it = iter(self.loader)
        k = 0
        nxt = self._load(it, k)
        while nxt is not None:
            images, labels, ready, host, slot = nxt
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            for t in (images, labels):
                t.record_stream(cur)                  # allocator: these blocks were allocated on the copy stream
            k += 1
            nxt = self._load(it, k)                   # batch k+1 is copied and staged while the caller trains on batch k
            try:
                yield images, labels
            finally:                                  # also when the caller leaves the loop early (generator closed)
                done = torch.cuda.Event()             # the caller has enqueued everything that reads this slot
                done.record(torch.cuda.current_stream(self.device))
                self._free[slot] = done
            del host
