#!/usr/bin/env python3
"""Host-buffer entry points (what Model.hamilton / Model.eigenval cost end to end, PCIe legs included)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

for n_orb, n_r, n_k in ((64, 256, 20000), (8, 95, 100000), (64, 4096, 20000)):
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 9)
    k = syn.random_kpoints(n_k)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.eigenval(k[:256])  # stage + warm up
    model.hamilton(k[:256])
    for name, fn in (("eigenval", model.eigenval), ("hamilton", model.hamilton), ("hamilton conv1", lambda kk: model.hamilton(kk, convention=1))):
        out = None
        dt = 1e9
        for _ in range(3):  # best of three: the first call also pays the page faults of a fresh output array
            del out
            t0 = time.perf_counter()
            out = fn(k)
            dt = min(dt, time.perf_counter() - t0)
        nbytes = out.nbytes if isinstance(out, np.ndarray) else sum(o.nbytes for o in out)
        print("N=%d N_R=%d NK=%d %-15s %8.1f ms  %9.0f k-points/s  out %.2f GB  %.1f GB/s" % (
            n_orb, n_r, n_k, name, dt * 1e3, n_k / dt, nbytes / 1e9, nbytes / dt / 1e9))
    model.pin_staging()
    t0 = time.perf_counter()
    model.eigenval(k)
    print("   eigenval with pinned staging: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
