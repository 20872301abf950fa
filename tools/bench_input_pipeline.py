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


def cpu_quota():
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if a == "max" else round(int(a) / int(b), 2)
    except (OSError, ValueError):
        return None


def photo_like(rng, w, h):
    """An image with the entropy of a photograph, not of a smooth gradient: multi-octave noise (1/f-like spectrum: every octave upsampled
    from a coarser grid), a few hard edges (rectangles / discs of flat colour) and sensor-like fine noise. At quality 90 a 500 x 375 image
    of this kind is ~105-115 KB, ImageNet's typical size (round 2-4 used a 48 x 64 patch upscaled bicubically: ~25 KB, far cheaper to
    entropy-decode)."""
    from PIL import Image
    acc = np.zeros((h, w, 3), dtype=np.float32)
    amp, total = 1.0, 0.0
    for octave in range(7):
        gh, gw = max(2, h >> (6 - octave)), max(2, w >> (6 - octave))
        grid = rng.random((gh, gw, 3), dtype=np.float32)
        up = np.asarray(Image.fromarray((grid * 255).astype(np.uint8)).resize((w, h), Image.BICUBIC), dtype=np.float32) / 255.0
        acc += amp * up
        total += amp
        amp *= 1.1
    acc /= total
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(6):      # hard edges
        colour = rng.random(3, dtype=np.float32)
        if rng.random() < 0.5:
            x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
            x1, y1 = x0 + int(rng.integers(30, w // 2)), y0 + int(rng.integers(30, h // 2))
            mask = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1)
        else:
            cx, cy, r = int(rng.integers(0, w)), int(rng.integers(0, h)), int(rng.integers(20, min(w, h) // 3))
            mask = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        acc[mask] = 0.6 * acc[mask] + 0.4 * colour
    acc += rng.normal(0.0, 0.07, size=acc.shape).astype(np.float32)
    return (np.clip(acc, 0, 1) * 255).astype(np.uint8)


def write_jpeg(arg):
    from PIL import Image
    d, i = arg
    rng = np.random.default_rng(1000 + i)
    w, h = int(rng.integers(400, 600)), int(rng.integers(300, 450))
    path = os.path.join(d, f"{i}.jpg")
    Image.fromarray(photo_like(rng, w, h)).save(path, quality=90)
    return os.path.getsize(path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else min(16, os.cpu_count() or 4)
    nice = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None          # niceness of the decode workers (pipeline.worker_init); default: none
    quick = len(sys.argv) > 5 and sys.argv[5] == "quick"            # only synthetic / loader-only / canvas + prefetch
    sub = int(sys.argv[6]) if len(sys.argv) > 6 else 4              # sub-batches per step (train.py's data.sub_batches)
    from PIL import Image
    from openset_imagenet import ResNet50, EntropicOpensetLoss, AverageMeter, optim, tools, pipeline as P
    from openset_imagenet.train import train
    from openset_imagenet.util import NameSpace
    tools.set_device_gpu(0)
    C = 30
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        rows = []
        n_files = min(n, 1024)
        import multiprocessing as mp
        with mp.Pool(min(16, os.cpu_count() or 1)) as pool:       # ~0.4 s per image on one core
            sizes = pool.map(write_jpeg, [(d, i) for i in range(n_files)])
        for i in range(n):
            rows.append(f"{i % n_files}.jpg,{-1 if i % 3 == 0 else i % C}")
        csv = os.path.join(d, "p2_train.csv")
        open(csv, "w").write("\n".join(rows) + "\n")

        def loader(uint8, nw=None, nice=nice, sub=1):
            nw = workers if nw is None else nw
            ds = P.CanvasDataset(csv, d, True, "entropic", uint8)
            import functools
            return torch.utils.data.DataLoader(ds, batch_size=B // sub, shuffle=True, num_workers=nw, pin_memory=True, drop_last=True,
                                               persistent_workers=nw > 0, prefetch_factor=4 * sub if nw > 0 else None,
                                               worker_init_fn=None if nice is None else functools.partial(P.worker_init, niceness=nice))

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

        out = {"images": n, "batch": B, "workers": workers, "worker_niceness": nice, "host_cpus": os.cpu_count(), "cgroup_cpu_quota": cpu_quota()}
        out["jpeg_bytes_mean"] = int(np.mean(sizes))
        # one core's cost per image, split: decode vs Resize(256) (what the host still does)
        t_dec = t_res = 0.0
        for i in range(64):
            t0 = time.perf_counter()
            im = Image.open(os.path.join(d, f"{i}.jpg")).convert("RGB")
            t1 = time.perf_counter()
            im = im.resize(P.resize_size(*im.size), Image.BILINEAR)
            t2 = time.perf_counter()
            t_dec += t1 - t0; t_res += t2 - t1
        out["host_ms_per_image_one_core"] = {"jpeg_decode": round(t_dec / 64 * 1e3, 3), "resize256": round(t_res / 64 * 1e3, 3)}

        class Resident:     # the synthetic headline inside the SAME loop: one device-resident batch yielded len(loader) times
            def __init__(self, n_batches):
                self.n = n_batches
                g = torch.Generator(device="cuda").manual_seed(1)
                self.x = torch.rand(B, 3, 224, 224, device="cuda", generator=g)
                self.y = torch.randint(-1, C, (B,), device="cuda", generator=g)
            def __len__(self): return self.n
            def __iter__(self):
                for _ in range(self.n):
                    yield self.x, self.y

        def note(msg):
            print(f"[bench_input_pipeline] {msg}", file=sys.stderr, flush=True)
        note(f"{n_files} JPEG files written, mean {out['jpeg_bytes_mean']} bytes")
        ref, canv = (None if quick else loader(False)), loader(True)
        epoch(canv)                                  # warm-up: worker start-up, kernels, allocator
        out["jpeg_files"] = n_files
        out["train_synthetic_resident_img_s"] = round(epoch(Resident(len(canv))), 1)
        note("synthetic done")
        if not quick:
            out["loader_only_reference_fp32_img_s"] = round(epoch(ref, False), 1)
        out["loader_only_canvas_u8_img_s"] = round(epoch(canv, False), 1)
        note("loader-only done")
        if not quick:
            out["train_reference_style_img_s"] = round(epoch(ref), 1)
            out["train_canvas_img_s"] = round(epoch(Staged(canv)), 1)
            note("reference-style / staged done")
        out["train_canvas_prefetch_img_s"] = round(epoch(P.DevicePrefetcher(canv)), 1)
        note("prefetch done")
        if sub > 1 and B % sub == 0:                 # what worker() builds by default: sub-batches grouped by the prefetcher
            fine = P.DevicePrefetcher(loader(True, sub=sub), group=sub)
            epoch(fine)                              # worker start-up
            out["sub_batches"] = sub
            out["train_canvas_prefetch_sub_img_s"] = round(epoch(fine), 1)
            out["train_canvas_prefetch_sub_img_s_2nd"] = round(epoch(fine), 1)
        out["train_synthetic_resident_img_s_after"] = round(epoch(Resident(len(canv))), 1)
        out["end_to_end_vs_synthetic"] = round(out["train_canvas_prefetch_img_s"] / out["train_synthetic_resident_img_s"], 4)
        if "train_canvas_prefetch_sub_img_s" in out:
            out["end_to_end_vs_synthetic_sub"] = round(max(out["train_canvas_prefetch_sub_img_s"], out["train_canvas_prefetch_sub_img_s_2nd"])
                                                       / out["train_synthetic_resident_img_s"], 4)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
