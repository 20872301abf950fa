"""train_imagenet.py — command line of the reference (openset_imagenet/script/train.py:8-63): `configuration protocol
[-o DIR] [-g [IDX]] [--nice N]`. One fix over the reference: `-g 0` / bare `-g` selects GPU 0 (the reference tests `if args.gpu:`,
which is false for index 0, script/train.py:58). `--synthetic N` (new) trains on N synthetic samples instead of the protocol CSVs."""
import argparse
import os
import pathlib

from .. import train as _train
from .. import util


def get_args(command_line_options=None):
    p = argparse.ArgumentParser("Imagenet Training Parameters", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("configuration", type=pathlib.Path, help="The configuration file that defines the experiment")
    p.add_argument("protocol", type=int, choices=(1, 2, 3), help="Open set protocol: 1, 2 or 3")
    p.add_argument("--output-directory", "-o", type=pathlib.Path, default=".", help="Directory to store the trained models into")
    p.add_argument("--gpu", "-g", type=int, nargs="?", default=None, const=0, help="GPU index (bare -g = 0)")
    p.add_argument("--nice", type=int, default=20, help="Select Priority Level")
    p.add_argument("--synthetic", type=int, default=0, help="train on this many synthetic samples instead of the protocol CSV files")
    args = p.parse_args(command_line_options)
    os.nice(args.nice)
    return args


def main(command_line_options=None):
    args = get_args(command_line_options)
    config = util.load_yaml(args.configuration)
    if args.gpu is not None:
        config.gpu = args.gpu
    config.protocol = args.protocol
    config.output_directory = args.output_directory
    if args.synthetic:
        config.data.synthetic = args.synthetic
    return _train.worker(config)


if __name__ == "__main__":
    main()
