#!/usr/bin/env python3
"""One-k-point eigenval / hamilton calls at the headline shape, for `rocprofv3 --kernel-trace` (timeline of the last calls)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

n_orb, n_r = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 4096)
r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 2)
model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
k = syn.random_kpoints(64)
for q in range(8):
    model.eigenval(k[q])
t0 = time.perf_counter()
for q in range(8, 40):
    model.eigenval(k[q])
t1 = time.perf_counter()
for q in range(8, 40):
    model.hamilton(k[q])
t2 = time.perf_counter()
print("eigenval %.1f us per call, hamilton %.1f us per call" % ((t1 - t0) / 32 * 1e6, (t2 - t1) / 32 * 1e6))
