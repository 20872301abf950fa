"""One rank of tests/test_dp_worker_gpu.py: runs the drop-in's worker() (reference train.py:237-482 surface) under the environment
a launcher provides (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*), then reports what the test asserts on: a hash of the final
parameters and BN buffers, the number of checkpoint writes by this rank, and the best score."""
import hashlib
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    cfg_path, out_dir = sys.argv[1], sys.argv[2]
    from openset_imagenet import util
    from openset_imagenet import train as T
    cfg = util.load_yaml(cfg_path)
    cfg.protocol, cfg.output_directory = 2, out_dir
    cfg.data.synthetic = 24
    best = T.worker(cfg)
    st = T._last_worker_state
    import torch
    torch.cuda.synchronize()
    # validation on every rank (train.validate(shard=...)): this rank's trackers after the last epoch, and — same model, same process —
    # the single-process loop over the WHOLE validation set (the reference's "validate only on first process" computation)
    meters = lambda tr: {k: [m.val, m.avg, m.sum, m.count] for k, m in tr.items()}
    sharded = meters(st["v_metrics"])
    from openset_imagenet import losses as L
    ds = st["val_loader"].loader.dataset if hasattr(st["val_loader"], "loader") else st["val_loader"].dataset
    whole = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size)
    single = {"j": L.AverageMeter(), "conf_kn": L.AverageMeter(), "conf_unk": L.AverageMeter()}
    T.validate(st["model"], whole, st["loss_fn"], st["n_classes"], single, cfg)
    h = hashlib.sha1(st["model"].flat_parameters().cpu().numpy().tobytes()).hexdigest()
    hb = hashlib.sha1(st["model"]._flat_buffers.cpu().numpy().tobytes()).hexdigest()
    print("RESULT " + json.dumps({"rank": st["rank"], "world": st["world"], "params": h, "buffers": hb,
                                  "checkpoints_written": st["checkpoints_written"], "best": best,
                                  "sharded_validation": st["sharded_validation"], "v_sharded": sharded, "v_single": meters(single),
                                  "val_samples": len(ds)}), flush=True)


if __name__ == "__main__":
    main()
