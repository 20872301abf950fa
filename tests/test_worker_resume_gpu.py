"""worker() with a learning-rate schedule (`opt.decay` > 0): the scheduler steps after validation and BEFORE the epoch's checkpoints
are written (reference train.py:435-437, then :463-471), so `_curr.pth` carries the scheduler state and learning rate of the epoch it
resumes into, and an interrupted + resumed run decays on the same epochs as an uninterrupted one."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(tmp, name, epochs, checkpoint=None):
    from openset_imagenet import util
    cfg = util.load_yaml(os.path.join(os.path.dirname(__file__), "..", "config", "train.yaml"))
    cfg.epochs, cfg.batch_size, cfg.workers, cfg.parallel, cfg.gpu, cfg.protocol = epochs, 8, 0, True, 0, 2
    cfg.loss.type = "entropic"
    cfg.name = name
    cfg.opt.type, cfg.opt.lr, cfg.opt.decay, cfg.opt.gamma = "adam", 1e-3, 1, 0.5
    cfg.data.synthetic = 16
    cfg.checkpoint = checkpoint
    cfg.output_directory = str(tmp / name)
    return cfg


def _state(path):
    ck = torch.load(path, weights_only=False)
    return ck["epoch"], ck["scheduler"]["last_epoch"], ck["opt_state_dict"]["param_groups"][0]["lr"]


def test_resume_with_lr_decay_matches_an_uninterrupted_run(cuda, tmp_path):
    from openset_imagenet.train import worker
    worker(_cfg(tmp_path, "whole", 3))
    assert _state(tmp_path / "whole" / "whole_curr.pth") == (3, 3, pytest.approx(1e-3 * 0.5 ** 3))
    worker(_cfg(tmp_path, "part", 2))
    # the checkpoint written after epoch index 1 resumes into epoch 2: scheduler and learning rate are already those of epoch 2
    assert _state(tmp_path / "part" / "part_curr.pth") == (2, 2, pytest.approx(1e-3 * 0.5 ** 2))
    lrs = []
    import openset_imagenet.train as T
    orig = T.train

    def spy(net, loader, opt, *a, **k):
        lrs.append(opt.param_groups[0]["lr"])
        return orig(net, loader, opt, *a, **k)
    T.train = spy
    try:
        worker(_cfg(tmp_path, "resumed", 3, checkpoint=str(tmp_path / "part" / "part_curr.pth")))
    finally:
        T.train = orig
    assert lrs == [pytest.approx(1e-3 * 0.5 ** 2)]          # one remaining epoch, trained at the rate an uninterrupted run uses there
    assert _state(tmp_path / "resumed" / "resumed_curr.pth") == (3, 3, pytest.approx(1e-3 * 0.5 ** 3))
