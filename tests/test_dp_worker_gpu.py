"""Data parallel through the drop-in surface: two ranks run worker() — the function train_imagenet.py calls — for one synthetic
epoch. RCCL cannot place two ranks on the one GPU of the test box, so `dist.backend: gloo` carries the collectives here (GPU
tensors over gloo); everything else is the production path: RANK / WORLD_SIZE / LOCAL_RANK from the launcher, process group
initialised inside worker(), DistributedSampler shards, dp.DistributedDataParallel wrap (broadcast + bucketed all-reduce per
backward stage), validation on every rank (whole batches, trackers bit-identical to the single-process loop), rank-0-only logging /
checkpoints (reference intent: train.py:248, train.yaml:18,35-39)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(900)
def test_worker_two_ranks_one_epoch(cuda, tmp_path):
    from openset_imagenet import util
    cfg = util.load_yaml(os.path.join(os.path.dirname(__file__), "..", "config", "train.yaml"))
    cfg.epochs, cfg.batch_size, cfg.workers, cfg.parallel = 1, 4, 0, True
    cfg.loss.type = "entropic"
    cfg.dist.backend = "gloo"
    cfgp = tmp_path / "train.yaml"
    cfgp.write_text(cfg.dump())
    out = tmp_path / "out"
    port = _free_port()
    rank_script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_worker_rank.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, rank_script, str(cfgp), str(out)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=800)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    res = []
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o[-4000:]}"
        res.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    res.sort(key=lambda r: r["rank"])
    assert [r["rank"] for r in res] == [0, 1] and all(r["world"] == 2 for r in res)
    assert res[0]["params"] == res[1]["params"], "replicas diverged: parameters differ between the ranks after the epoch"
    assert res[0]["checkpoints_written"] >= 1 and res[1]["checkpoints_written"] == 0, "only the first process writes checkpoints"
    assert res[0]["best"] == res[1]["best"]
    # validation ran on BOTH ranks (each scored the batches rank, rank + 2, ... of the unshuffled validation set: 6 samples at batch 4 =
    # one whole batch + a ragged one) and both hold the trackers a single process computes on the whole set, bit for bit
    for r in res:
        assert r["sharded_validation"] is True and r["val_samples"] == 6
        assert r["v_sharded"] == r["v_single"], f"rank {r['rank']}: sharded {r['v_sharded']} vs single-process {r['v_single']}"
        assert r["v_sharded"]["j"][3] == 6
    assert res[0]["v_sharded"] == res[1]["v_sharded"]
    files = sorted(f.name for f in out.iterdir())
    assert "experiment_curr.pth" in files and "training.log" in files and "scalars-training.log.csv" in files
    ck = torch.load(out / "experiment_curr.pth", weights_only=False)
    assert ck["epoch"] == 1 and len(ck["model_state_dict"]) == 321 and not any(k.startswith("module.") for k in ck["model_state_dict"])
    # 24 synthetic samples, 2 ranks, batch 4 per GPU: every rank ran 3 steps on its own shard
    assert int(ck["model_state_dict"]["resnet_base.bn1.num_batches_tracked"]) == 3
