"""Global-device helpers with the call contract of `vast.tools` as the reference uses it
(openset_imagenet/train.py:13,78,128-129,315,318; openset_imagenet/losses.py:11,13,17):
`set_device_gpu(index)`, `set_device_cpu()`, `device(x)` and the module global `_device`.
`vast` is an un-vendored third-party dependency of the reference; only these symbols are on the hot path.
"""
import torch

_device = torch.device("cpu")


def set_device_gpu(index=0):
    """Select GPU `index` as the global device (HIP devices appear as 'cuda' in PyTorch-ROCm)."""
    global _device
    if not torch.cuda.is_available():
        raise RuntimeError("set_device_gpu: no GPU visible to PyTorch-ROCm")
    _device = torch.device(f"cuda:{int(index)}")
    torch.cuda.set_device(_device)
    return _device


def set_device_cpu():
    global _device
    _device = torch.device("cpu")
    return _device


def get_device():
    return _device


def device(x):
    """Move a tensor or module to the global device (identity when already there). Tensors in pinned host memory (DataLoader
    pin_memory=True, reference train.py:304) are copied asynchronously on the current stream: no host synchronisation."""
    if isinstance(x, torch.Tensor):
        return x.to(_device, non_blocking=True)
    return x.to(_device)
