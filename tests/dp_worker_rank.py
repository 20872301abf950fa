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
    h = hashlib.sha1(st["model"].flat_parameters().cpu().numpy().tobytes()).hexdigest()
    hb = hashlib.sha1(st["model"]._flat_buffers.cpu().numpy().tobytes()).hexdigest()
    print("RESULT " + json.dumps({"rank": st["rank"], "world": st["world"], "params": h, "buffers": hb,
                                  "checkpoints_written": st["checkpoints_written"], "best": best}), flush=True)


if __name__ == "__main__":
    main()
