#!/usr/bin/env python3
"""Locate rows where a chunked eigenval run differs from single-k-point runs (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

n_r = int(os.environ.get("NR", "64"))
r_vec, hop, pos = syn.dense_model_arrays(64, n_r, syn.MODEL_SEED + 2)
k = syn.random_kpoints(100_000)
model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
model.pin_staging()
traces = np.einsum("rii->r", hop)
tr = np.empty(len(k))
for lo in range(0, len(k), 8192):
    tr[lo:lo + 8192] = 2.0 * (np.exp(2j * np.pi * (k[lo:lo + 8192] @ r_vec.T.astype(float))) @ traces).real
for rep in range(3):
    out = np.array(model.eigenval(k))
    err = np.abs(out.sum(axis=1) - tr)
    bad = np.flatnonzero(err > 1e-9)
    print("rep", rep, "bad rows:", len(bad), bad[:12], "..." if len(bad) > 12 else "", "max err %.3g" % err.max())
