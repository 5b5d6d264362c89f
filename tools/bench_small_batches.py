#!/usr/bin/env python3
"""eigenval latency of small batches above 128 orbitals: two-stage (default) vs one-stage (TBK_BAND=0) reduction."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

for n in (160, 256, 384, 512):
    r_vec, hop, pos = syn.dense_model_arrays(n, 16, syn.MODEL_SEED + n)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    for nk in (1, 8, 64, 256, 512, 1024):
        k = syn.random_kpoints(nk)
        model.eigenval_array(k)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            model.eigenval_array(k)
        dt = (time.perf_counter() - t0) / reps
        print("n=%3d nk=%5d  %9.1f us per call  %7.2f us per k-point  (TBK_BAND=%s)" % (n, nk, dt * 1e6, dt / nk * 1e6, os.environ.get("TBK_BAND", "1")), flush=True)
