"""Label bookkeeping of the reference's ImagenetDataset that feeds the hot path (reference openset_imagenet/dataset.py).

Only the label arithmetic is mirrored — `replace_negative_label` (dataset.py:60-68), `remove_negative_label` (dataset.py:70-74)
and `calculate_class_weights` (dataset.py:77-86) — because their results parameterise the garbage / softmax losses
(train.py:287-293, 344-347). JPEG decoding and the DataLoader are outside the hot path (the benchmark feeds device-resident
synthetic batches); `SyntheticImagenet` provides batches of the reference's shape and dtype for loops and smoke runs.
"""
import numpy as np
import torch


class LabelTable:
    """The label column of a protocol CSV (`p{N}_{train,val}.csv`, second column) with the reference's rewrites."""

    def __init__(self, labels):
        self.labels = np.asarray(labels, dtype=np.int64).copy()
        self.unique_classes = np.sort(np.unique(self.labels))
        self.label_count = len(self.unique_classes)          # counts the -1 class, like dataset.py:26

    @classmethod
    def from_csv(cls, csv_file):
        import pandas as pd
        return cls(pd.read_csv(csv_file, header=None)[1].to_numpy())

    def has_negatives(self):
        return -1 in self.unique_classes

    def replace_negative_label(self):
        """-1 -> label_count - 1 (the background class of the garbage loss), dataset.py:60-68."""
        biggest = self.label_count - 1
        self.labels[self.labels == -1] = biggest
        self.unique_classes[self.unique_classes == -1] = biggest
        self.unique_classes.sort()

    def remove_negative_label(self):
        """Drop every negative label (plain softmax training), dataset.py:70-74."""
        self.labels = self.labels[self.labels >= 0]
        self.unique_classes = np.sort(np.unique(self.labels))
        self.label_count = len(self.unique_classes)

    def calculate_class_weights(self):
        """w_c = N / (count_c * label_count), ordered by ascending label (dataset.py:77-86)."""
        _, counts = np.unique(self.labels, return_counts=True)
        return torch.from_numpy(len(self.labels) / (counts * self.label_count)).float()


class ImagenetDataset(torch.utils.data.Dataset):
    """The reference's dataset class with its own constructor and members (reference dataset.py:10-86):
    `ImagenetDataset(csv_file, imagenet_path, transform=None)`, samples `(image, label)` with `transform` applied to the decoded
    RGB PIL image, attributes `dataset` (the CSV frame), `imagenet_path`, `transform`, `label_count`, `unique_classes`, and the
    label rewrites `has_negatives / replace_negative_label / remove_negative_label / calculate_class_weights`.

    This is the compatibility surface for reference-style callers; worker() itself feeds the GPU through pipeline.CanvasDataset
    (uint8 canvases, crop / flip / ToTensor on the device). Both share LabelTable, so the label arithmetic is one implementation."""

    def __init__(self, csv_file, imagenet_path, transform=None):
        import pathlib
        import pandas as pd
        self.dataset = pd.read_csv(csv_file, header=None)
        self.imagenet_path = pathlib.Path(imagenet_path)
        self.transform = transform
        self._sync(LabelTable(self.dataset[1].to_numpy()))

    def _sync(self, table):
        self._table = table
        self.label_count, self.unique_classes = table.label_count, table.unique_classes

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        from PIL import Image
        if torch.is_tensor(index):
            index = index.tolist()
        rel_path, label = self.dataset.iloc[index]
        image = Image.open(self.imagenet_path / rel_path).convert("RGB")
        if self.transform is not None:
            image = self.transform(image)
        return image, torch.as_tensor(int(label), dtype=torch.int64)

    def has_negatives(self):
        return self._table.has_negatives()

    def replace_negative_label(self):
        self._table.replace_negative_label()
        self.dataset[1] = self._table.labels
        self._sync(self._table)

    def remove_negative_label(self):
        self.dataset = self.dataset[self.dataset[1] >= 0].reset_index(drop=True)
        self._table.remove_negative_label()
        self._sync(self._table)

    def calculate_class_weights(self):
        return self._table.calculate_class_weights()


class SyntheticImagenet(torch.utils.data.Dataset):
    """Deterministic stand-in for ImagenetDataset: fp32 images in [0,1) of shape [3,224,224] (ToTensor() without mean/std
    normalisation, reference train.py:259-263) and int64 labels drawn from a given label table."""

    def __init__(self, labels, image_size=224, seed=0):
        self.labels = torch.as_tensor(labels, dtype=torch.int64)
        self.image_size, self.seed = image_size, seed

    def __len__(self):
        return self.labels.numel()

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(index))
        return torch.rand(3, self.image_size, self.image_size, generator=g), self.labels[index]
