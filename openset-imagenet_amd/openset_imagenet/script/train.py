"""train_imagenet.py — command line of the reference (openset_imagenet/script/train.py:8-63): `configuration protocol
[-o DIR] [-g [IDX]] [--nice N]`. One fix over the reference: `-g 0` / bare `-g` selects GPU 0 (the reference tests `if args.gpu:`,
which is false for index 0, script/train.py:58). `--synthetic N` (new) trains on N synthetic samples instead of the protocol CSVs.

Data parallel: started under `python -m torch.distributed.run` (RANK / WORLD_SIZE / LOCAL_RANK in the environment) every rank
simply runs worker(); with `dist.distributed: on` in the configuration (config/train.yaml, the reference's unused `dist:` block)
and no such environment, this process becomes the launcher: it starts `dist.gpus` child ranks of the same command line on
127.0.0.1:`dist.port` BEFORE anything touches the GPU, waits for them and returns rank 0's exit status."""
import argparse
import os
import pathlib
import subprocess
import sys

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


def _launch_ranks(config, argv):
    """Start config.dist.gpus ranks of this command line (one per GPU) and wait; no GPU call has happened in this process."""
    n = int(config.dist.gpus)
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(config.dist.port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "openset_imagenet.script.train", *argv], env=env))
    return _wait_ranks(procs)


def _wait_ranks(procs, poll_seconds=0.2, grace_seconds=10.0):
    """Poll the child ranks; on the first non-zero exit terminate the survivors (they would otherwise sit in an RCCL collective
    until the process-group timeout) and report every exit code."""
    import time
    codes = [None] * len(procs)
    failed = False
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if not failed and any(c not in (None, 0) for c in codes):
            failed = True
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()                     # the exact children this function was given, never a pattern
            deadline = time.time() + grace_seconds
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=max(0.0, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
        if any(c is None for c in codes):
            time.sleep(poll_seconds)
    if any(codes):
        raise SystemExit(f"data-parallel ranks exited with {codes}")
    return 0


def main(command_line_options=None):
    args = get_args(command_line_options)
    config = util.load_yaml(args.configuration)
    dcfg = getattr(config, "dist", None)
    if dcfg is not None and getattr(dcfg, "distributed", False) and int(getattr(dcfg, "gpus", 1) or 1) > 1 and "RANK" not in os.environ:
        return _launch_ranks(config, list(sys.argv[1:] if command_line_options is None else command_line_options))
    if args.gpu is not None:
        config.gpu = args.gpu
    config.protocol = args.protocol
    config.output_directory = args.output_directory
    if args.synthetic:
        config.data.synthetic = args.synthetic
    return _train.worker(config)


if __name__ == "__main__":
    main()
