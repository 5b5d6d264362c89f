#!/usr/bin/env python3
"""Repeat-run bit identity through Model.eigenval_array (H(k) + reduction + bisection): n n_r nk reps"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

n, n_r, nk, reps = (int(x) for x in sys.argv[1:5])
r_vec, hop, pos = syn.dense_model_arrays(n, n_r, syn.MODEL_SEED + n)
k = syn.random_kpoints(nk, seed=n)
model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
first = model.eigenval_array(k).copy()
ham0 = model.hamilton(k[:256]).copy()
bad_total = 0
for rep in range(reps):
    again = model.eigenval_array(k)
    bad = np.flatnonzero(np.any(again != first, axis=1))
    bad_total += len(bad)
    if len(bad):
        i = bad[0]
        print("rep %d: %d rows differ; first %d: max |dE| %.2e, columns %s" % (
            rep, len(bad), i, np.abs(again[i] - first[i]).max(), np.flatnonzero(again[i] != first[i])[:8]))
    if rep % 8 == 0 and not np.array_equal(model.hamilton(k[:256]), ham0):
        print("rep %d: hamilton differs" % rep)
print("n=%d n_r=%d nk=%d reps=%d FUSE=%s: %d differing rows" % (n, n_r, nk, reps, os.environ.get("TBK_BAND_FUSE"), bad_total))
