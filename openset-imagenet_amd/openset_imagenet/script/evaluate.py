"""Evaluation entry point with the surface of the reference's script/evaluate.py (evaluate.py:15-149): same positional arguments
and options, same checkpoint naming (`{loss}_best.pth` / `{loss}_curr.pth` in the output directory), same output files
`{loss}_{val,test}_arr{suffix}.npz` with `gt`, `logits`, `features`, `scores`. The forward passes run on the MI355X executor in
eval mode (train.get_arrays); there is no CPU evaluation path. `--oscr` (this build's addition) also prints the area-free summary
of the OSCR curve computed on the GPU (util.calculate_oscr)."""
import argparse
import pathlib

import numpy as np
import torch

from .. import tools, util
from ..model import ResNet50
from ..train import _image_loader, get_arrays, load_checkpoint


def get_args(command_line_options=None):
    p = argparse.ArgumentParser("Get parameters for evaluation", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("loss", choices=["entropic", "softmax", "garbage"], help="Which loss function to evaluate")
    p.add_argument("protocol", type=int, choices=(1, 2, 3), help="Which protocol to evaluate")
    p.add_argument("--use-best", "-b", action="store_true", help="Take the best model of the validation set, otherwise the last")
    p.add_argument("--gpu", "-g", type=int, nargs="?", default=None, const=0, help="GPU index (bare -g = 0); required")
    p.add_argument("--imagenet-directory", type=pathlib.Path, default=pathlib.Path("/local/scratch/datasets/ImageNet/ILSVRC2012/"))
    p.add_argument("--protocol-directory", type=pathlib.Path, default="protocols", help="Where are the protocol files stored")
    p.add_argument("--output-directory", default="experiments/Protocol_{}", help="Where to find the results of the experiments")
    p.add_argument("--batch-size", type=int, default=64)
    p.add_argument("--workers", type=int, default=4)
    p.add_argument("--oscr", action="store_true", help="also compute the OSCR curve of each split on the GPU")
    args = p.parse_args(command_line_options)
    try:
        args.output_directory = str(args.output_directory).format(args.protocol)
    except Exception:
        pass
    args.output_directory = pathlib.Path(args.output_directory)
    return args


def main(command_line_options=None):
    args = get_args(command_line_options)
    if args.gpu is None:
        raise RuntimeError("No GPU device selected: the MI355X build has no CPU evaluation path (pass -g [index])")
    tools.set_device_gpu(index=args.gpu)
    splits = {}
    for split in ("val", "test"):
        ds = _image_loader(args.protocol_directory / f"p{args.protocol}_{split}.csv", args.imagenet_directory, False, "eval")
        splits[split] = ds
        print(f"{split} dataset len:{len(ds)}, labels:{ds.table.label_count}")
    n_labels = splits["val"].table.label_count
    n_classes = n_labels if args.loss == "garbage" else n_labels - 1     # evaluate.py:121-124
    suffix = "_best" if args.use_best else "_curr"
    model = ResNet50(fc_layer_dim=n_classes, out_features=n_classes, logit_bias=False)
    start_epoch, best_score = load_checkpoint(model, args.output_directory / (args.loss + suffix + ".pth"))
    print(f"Taking model from epoch {start_epoch} that achieved best score {best_score}")
    tools.device(model)
    written = {}
    for split, ds in splits.items():
        loader = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, num_workers=args.workers)
        gt, logits, features, scores = get_arrays(model=model, loader=loader)
        file_path = args.output_directory / f"{args.loss}_{split}_arr{suffix}.npz"
        np.savez(file_path, gt=gt, logits=logits, features=features, scores=scores)
        print(f"Target labels, logits, features and scores saved in: {file_path}")
        written[split] = file_path
        if args.oscr:
            s = scores[:, :-1] if args.loss == "garbage" else scores           # the background column is not a known class
            for unk in (-1, -2):
                if (gt == unk).any():
                    ccr, fpr = util.calculate_oscr(gt, s, unk_label=unk)
                    at = ccr[np.searchsorted(-fpr, -0.1)] if len(ccr) and np.isfinite(fpr).all() and (fpr <= 0.1).any() else float("nan")
                    print(f"{split}: OSCR vs label {unk}: {len(ccr)} points, CCR@FPR<=0.1 = {at:.4f}")
    return written


if __name__ == "__main__":
    main()
