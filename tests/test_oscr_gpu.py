"""GPU parity of the OSCR curve (util.calculate_oscr, reference util.py:90-122) through the C ABI: bit-identical to the
reference-generated vectors and to the CPU oracle on larger random inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_oscr_matches_reference_vectors(cuda, golden_dir):
    from openset_imagenet.util import calculate_oscr
    G = np.load(os.path.join(golden_dir, "oscr_reference.npz"))
    for n in G["names"]:
        gt, scores, unk = G[f"{n}.gt"], G[f"{n}.scores"], int(G[f"{n}.unk"])
        ccr, fpr = calculate_oscr(gt, scores, unk_label=unk)
        assert ccr.dtype == np.float64 and fpr.dtype == np.float64
        assert np.array_equal(ccr, G[f"{n}.ccr"], equal_nan=True), f"{n}: ccr"
        assert np.array_equal(fpr, G[f"{n}.fpr"], equal_nan=True), f"{n}: fpr"
        # device-resident inputs (what validate()/get_arrays() hold) give the same answer
        c2, f2 = calculate_oscr(torch.from_numpy(gt).to(cuda), torch.from_numpy(scores).to(cuda), unk_label=unk)
        assert np.array_equal(c2, ccr, equal_nan=True) and np.array_equal(f2, fpr, equal_nan=True)


@pytest.mark.parametrize("N,C,dtype", [(20000, 116, np.float32), (7001, 151, np.float64)])
def test_oscr_test_set_sizes_vs_oracle(cuda, N, C, dtype):
    """Sizes of the protocols' test splits: same bits as the CPU oracle; curve properties: ccr and fpr are non-increasing in the
    threshold, bounded by the closed-set accuracy / 1, and the first point counts everything above the smallest target score."""
    from openset_imagenet.util import calculate_oscr
    from oracle.oscr_oracle import calculate_oscr as oracle_oscr
    rng = np.random.default_rng(N)
    z = rng.normal(size=(N, C)) * 3
    s = np.exp(z - z.max(1, keepdims=True)); s = (s / s.sum(1, keepdims=True)).astype(dtype)
    s[::7] = np.round(s[::7] * 64) / 64            # some exact ties
    gt = rng.integers(0, C, size=N); gt[rng.random(N) < 0.4] = -1
    ccr, fpr = calculate_oscr(gt, s)
    occr, ofpr = oracle_oscr(gt, s)
    assert np.array_equal(ccr, occr) and np.array_equal(fpr, ofpr)
    assert np.all(np.diff(ccr) <= 0) and np.all(np.diff(fpr) <= 0)
    kn = gt >= 0
    acc = float((s[kn].argmax(1) == gt[kn]).mean())
    assert ccr[0] <= acc + 1e-12 and 0 <= fpr[-1] <= fpr[0] <= 1
