"""openset_imagenet — MI355X-native training hot path with the reference package's interface.

Importable pieces (mirroring the reference package layout): `model.ResNet50`, `losses.EntropicOpensetLoss` / `AverageMeter` /
`EarlyStopping`, `train.train` / `save_checkpoint` / `load_checkpoint`, `util.NameSpace` / `load_yaml`, `tools.device` ...
"""
from . import tools, util
from .losses import (AverageMeter, EarlyStopping, EntropicOpensetLoss, GarbageLoss, ObjectosphereLoss, SoftmaxLoss)
from .model import ResNet50

__all__ = ["ResNet50", "EntropicOpensetLoss", "SoftmaxLoss", "GarbageLoss", "ObjectosphereLoss", "AverageMeter",
           "EarlyStopping", "tools", "util"]
