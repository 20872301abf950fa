"""openset_imagenet — MI355X-native training hot path with the reference package's interface.

Same import surface as the reference package (openset_imagenet/__init__.py:1-7): `ResNet50`, `ImagenetDataset` (the reference's
constructor `(csv_file, imagenet_path, transform=None)` and members; `CanvasDataset` is the uint8 hand-over worker() uses), and the
sub-modules `util`, `train`, `metrics`, `losses`; plus `tools` (the three `vast.tools` symbols the path uses), `dataset`,
`optim`, `dp`, `pipeline`. `OpenSetProtocol` (protocol.py, offline CSV generation from WordNet / robustness metadata) is out of
scope for this build and is not exported — see INTEGRATION.md.
"""
from . import tools, util
from . import dataset, losses, metrics, train
from .losses import (AverageMeter, EarlyStopping, EntropicOpensetLoss, GarbageLoss, ObjectosphereLoss, SoftmaxLoss)
from .model import ResNet50
from .dataset import ImagenetDataset
from .pipeline import CanvasDataset

__all__ = ["ResNet50", "ImagenetDataset", "CanvasDataset", "EntropicOpensetLoss", "SoftmaxLoss", "GarbageLoss", "ObjectosphereLoss", "AverageMeter",
           "EarlyStopping", "tools", "util", "train", "metrics", "losses", "dataset"]
