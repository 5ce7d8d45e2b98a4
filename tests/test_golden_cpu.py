"""CPU suite: the oracle against its frozen snapshot (tests/golden/*.npz, written by tests/golden/make_golden.py — SURVEY.md 8(c)
"Fixtures to commit").  Index / integer / selection outputs must match bit for bit, floating-point outputs to 1e-6 (relative to the
output's scale).  The fixtures pin the ORACLE over time (no reference output exists offline: parity stays unpinned against TF itself)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG      # noqa: E402


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_oracle_matches_its_frozen_fixture(name):
    torch.set_num_threads(1)
    path = os.path.join(HERE, "golden", name + ".npz")
    assert os.path.getsize(path) <= 100 * 1024
    z = np.load(path)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    want = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    exact = set(z["exact"].tolist())
    gen_inp, compute, meta = MG.CASES[name]()
    assert sorted(gen_inp) == sorted(inp) and set(meta["exact"]) == exact
    for k in inp:                                   # the committed inputs are what the seeded generator still produces
        assert np.array_equal(np.asarray(gen_inp[k]), inp[k]), "input %s of %s changed" % (k, name)
    got = compute(inp)
    assert sorted(got) == sorted(want)
    for k, w in want.items():
        g = np.asarray(got[k])
        assert g.shape == w.shape and g.dtype == w.dtype, (name, k, g.shape, w.shape, g.dtype, w.dtype)
        if k in exact or not np.issubdtype(w.dtype, np.floating):
            assert np.array_equal(g, w), "%s.%s differs from the frozen oracle output" % (name, k)
        else:
            scale = max(1.0, float(np.abs(w).max()))
            assert float(np.abs(g.astype(np.float64) - w.astype(np.float64)).max()) <= 1e-6 * scale, (name, k)
