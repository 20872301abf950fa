"""Configuration container with the semantics of the reference's util.NameSpace / util.load_yaml
(openset_imagenet/util.py:16-34): attribute access over a nested YAML mapping, `dump()` back to YAML text.
`calculate_oscr` mirrors util.py:90-122 on the GPU (osi_oscr_*); the plotting helpers of the reference are out of scope.
"""
import yaml


class NameSpace:
    def __init__(self, config):
        self.update(config)

    def update(self, config):
        for key, value in config.items():
            setattr(self, key, NameSpace(value) if isinstance(value, dict) else value)

    def dict(self):
        return {k: (v.dict() if isinstance(v, NameSpace) else v) for k, v in vars(self).items()}

    def dump(self, indent=4):
        return yaml.dump(self.dict(), indent=indent)

    def __repr__(self):
        return "NameSpace(" + repr(self.dict()) + ")"


def load_yaml(yaml_file):
    """Load a YAML configuration file into a NameSpace."""
    with open(yaml_file, "r") as handle:
        return NameSpace(yaml.safe_load(handle))


def calculate_oscr(gt, scores, unk_label=-1):
    """OSCR curve with the signature and return values of the reference (util.py:90-122): two float64 arrays (ccr, fpr), one
    point per distinct target-class score of the known samples except the largest.

    The counting runs on the MI355X (osi_oscr_f32 / osi_oscr_f64, exact integer counts, so the quotients are bit-identical to
    the reference's); `gt` / `scores` may be numpy arrays or torch tensors, on the host or already on the device (what
    validate()/get_arrays() hold). Score dtype float32 or float64 is kept — thresholds compare in the dtype they were stored in."""
    import numpy as np
    import torch
    from . import _native as N
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise RuntimeError("openset_imagenet (MI355X build) has no CPU path: calculate_oscr needs the GPU")
    s = torch.as_tensor(scores)
    if s.dtype not in (torch.float32, torch.float64):
        s = s.double()
    if s.dim() != 2:
        raise ValueError("scores must be [N_samples, N_classes]")
    s = s.to(dev).contiguous()
    y = torch.as_tensor(np.asarray(gt).astype(int) if not isinstance(gt, torch.Tensor) else gt).to(torch.int64).to(dev).contiguous()
    n, c = s.shape
    if y.numel() != n:
        raise ValueError("gt and scores disagree on the number of samples")
    if n == 0:
        return np.zeros(0), np.zeros(0)
    lib = N.lib()
    nb = lib.osi_oscr_workspace(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    taus = torch.empty(n, dtype=s.dtype, device=dev)
    ccr_c = torch.zeros(n, dtype=torch.int64, device=dev)
    fpr_c = torch.zeros(n, dtype=torch.int64, device=dev)
    totals = torch.zeros(3, dtype=torch.int64, device=dev)
    fn = lib.osi_oscr_f32 if s.dtype == torch.float32 else lib.osi_oscr_f64
    N.check(fn(N.ptr(s), N.ptr(y), n, c, int(unk_label), N.ptr(ws), nb, N.ptr(taus), N.ptr(ccr_c), N.ptr(fpr_c), N.ptr(totals),
               N.stream_of(s)), "osi_oscr")
    n_unique, total_kn, total_unk = (int(v) for v in totals.cpu())
    pts = max(0, n_unique - 1)
    with np.errstate(divide="ignore", invalid="ignore"):   # 0/0 -> nan like the reference when a class of samples is absent
        ccr = ccr_c[:pts].cpu().numpy() / np.int64(total_kn)
        fpr = fpr_c[:pts].cpu().numpy() / np.int64(total_unk)
    return np.asarray(ccr, dtype=np.float64), np.asarray(fpr, dtype=np.float64)
