"""Generates tests/golden/oscr_reference.npz — DEV CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Executes the reference's OWN `calculate_oscr` (/root/reference/openset_imagenet/util.py:90-122). The module cannot be imported
as a whole (its top imports matplotlib, which is not installed and cannot be fetched), so the function definition is taken from
the file's syntax tree at run time and compiled in memory with numpy — the function body that runs is the reference's, and no
source text is written anywhere. Only arrays (inputs / expected outputs) go into the fixture.
"""
import ast
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/openset_imagenet/util.py"


def load_reference_function(name):
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF_FILE, "exec"), ns)
    return ns[name]


def cases():
    rng = np.random.default_rng(2024)
    out = []

    def mk(name, N, C, p_neg, dtype, unk_label=-1, quant=None, p_other=0.0):
        z = rng.normal(size=(N, C)) * 2
        s = np.exp(z - z.max(1, keepdims=True)); s /= s.sum(1, keepdims=True)
        if quant:                                   # ties between target scores and between target / max scores
            s = np.round(s * quant) / quant
        gt = rng.integers(0, C, size=N)
        r = rng.random(N)
        gt[r < p_neg] = -1
        gt[(r >= p_neg) & (r < p_neg + p_other)] = -2
        out.append((name, gt.astype(np.int64), s.astype(dtype), unk_label))

    mk("mixed_f32", 257, 30, 0.4, np.float32)
    mk("mixed_f64", 200, 12, 0.3, np.float64)
    mk("ties_f32", 300, 8, 0.35, np.float32, quant=16)
    mk("ties_f64", 150, 5, 0.5, np.float64, quant=8)
    mk("unk_is_minus2", 180, 10, 0.2, np.float32, unk_label=-2, p_other=0.25)
    mk("no_unknown", 64, 6, 0.0, np.float32)        # fpr = 0/0
    mk("all_unknown", 40, 6, 1.1, np.float32)       # no target scores at all
    mk("single_known", 30, 7, 0.0, np.float32)
    out[-1] = (out[-1][0], np.where(np.arange(30) == 11, out[-1][1], -1), out[-1][2], -1)
    mk("one_row", 1, 4, 0.0, np.float64)
    mk("large", 2000, 30, 0.37, np.float32)
    return out


def main():
    fn = load_reference_function("calculate_oscr")
    out, names = {}, []
    for name, gt, scores, unk in cases():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ccr, fpr = fn(gt, scores, unk_label=unk)
        out[f"{name}.gt"], out[f"{name}.scores"], out[f"{name}.unk"] = gt, scores, np.int64(unk)
        out[f"{name}.ccr"], out[f"{name}.fpr"] = np.asarray(ccr, dtype=np.float64), np.asarray(fpr, dtype=np.float64)
        names.append(name)
        print(f"{name:16s} N={len(gt):5d} C={scores.shape[1]:4d} {scores.dtype}  points={len(ccr)}")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "oscr_reference.npz"), **out)


if __name__ == "__main__":
    main()
