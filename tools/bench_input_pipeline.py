"""Dev tool (GPU box): the real-data training loop on JPEG files, with and without the device-side input pipeline.

    python tools/bench_input_pipeline.py [n_images] [batch] [workers]

Writes up to 1024 JPEG files (ImageNet-like sizes around 500x375 — decode cost is what matters, not the pixels), a protocol CSV of
n rows cycling over them (an epoch must be long enough to amortise the ~0.5 s the workers need to fill the pipeline at its start),
and times one epoch of openset_imagenet.train.train() three ways:

  reference-style   fp32 CHW samples produced on the host workers (decode, Resize(256), crop, flip, ToTensor — the reference's
                    transform, train.py:259-263), default-collated, copied inside the step (train.py:128)
  canvas            uint8 canvases from the host (decode + Resize(256) only); crop / flip / ToTensor on the GPU, copies inside the step
  canvas+prefetch   the same through pipeline.DevicePrefetcher (copy stream, one batch ahead)

and the loaders alone (no model) to show where the host limit sits. Prints one JSON line.
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else min(16, os.cpu_count() or 4)
    from PIL import Image
    from openset_imagenet import ResNet50, EntropicOpensetLoss, AverageMeter, optim, tools, pipeline as P
    from openset_imagenet.train import train
    from openset_imagenet.util import NameSpace
    tools.set_device_gpu(0)
    C = 30
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        rows = []
        base = rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
        n_files = min(n, 1024)
        for i in range(n_files):
            w, h = int(rng.integers(400, 600)), int(rng.integers(300, 450))
            img = Image.fromarray(base).resize((w, h), Image.BICUBIC)       # smooth content: realistic JPEG entropy
            img.save(os.path.join(d, f"{i}.jpg"), quality=90)
        for i in range(n):
            rows.append(f"{i % n_files}.jpg,{-1 if i % 3 == 0 else i % C}")
        csv = os.path.join(d, "p2_train.csv")
        open(csv, "w").write("\n".join(rows) + "\n")

        def loader(uint8):
            ds = P.CanvasDataset(csv, d, True, "entropic", uint8)
            return torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True, num_workers=workers, pin_memory=True, drop_last=True,
                                               persistent_workers=workers > 0, prefetch_factor=4 if workers > 0 else None)

        class Staged:      # canvas batches staged inside the step (no copy stream): isolates what the prefetch adds
            def __init__(self, ld): self.ld = ld
            def __len__(self): return len(self.ld)
            def __iter__(self):
                for c, xy, f, y in self.ld:
                    dev = tools.get_device()
                    yield P.stage_canvas_batch(c.to(dev, non_blocking=True), xy.to(dev, non_blocking=True), f.to(dev, non_blocking=True)), y

        torch.manual_seed(0)
        model = tools.device(ResNet50(C, C, False))
        opt = optim.Adam(model.parameters(), lr=1e-3)
        loss = EntropicOpensetLoss(C, 1.0)
        cfg = NameSpace({"parallel": True})

        def epoch(ld, with_model=True):
            it_n = 0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if with_model:
                train(model, ld, opt, loss, {"j": AverageMeter()}, cfg)
                it_n = len(ld)
            else:
                for batch in ld:
                    it_n += 1
            torch.cuda.synchronize()
            return it_n * B / (time.perf_counter() - t0)

        out = {"images": n, "batch": B, "workers": workers, "host_cpus": os.cpu_count()}
        ref, canv = loader(False), loader(True)
        epoch(canv)                                  # warm-up: worker start-up, kernels, allocator
        out["jpeg_files"] = n_files
        out["loader_only_reference_fp32_img_s"] = round(epoch(ref, False), 1)
        out["loader_only_canvas_u8_img_s"] = round(epoch(canv, False), 1)
        out["train_reference_style_img_s"] = round(epoch(ref), 1)
        out["train_canvas_img_s"] = round(epoch(Staged(canv)), 1)
        out["train_canvas_prefetch_img_s"] = round(epoch(P.DevicePrefetcher(canv)), 1)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
