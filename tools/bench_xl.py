#!/usr/bin/env python3
"""Above 1024 orbitals: the own launch chain (csrc/tbk_eig_band_xl.hip, band_xl_*) against rocsolver_zheevd_strided_batched,
whole `eigenval` (H(k) of a dense N_R = 4 model + eigenvalues) of nk k-points, and the stage times of the own path.

    python tools/bench_xl.py [--own] [sizes ...]        (GPU box; default 1030 1536 2048; --own: without the rocSOLVER runs,
                                                          e.g. under rocprofv3 -- rocSOLVER is tens of thousands of launches)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn, _lib  # noqa: E402

own_only = "--own" in sys.argv[1:]
batches = (64,) if "--nk64" in sys.argv[1:] else (1, 8, 64, 256)  # (--nk64: one batch size, for profiler runs)
sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [1030, 1536, 2048]
for n in sizes:
    r_vec, hop, pos = syn.dense_model_arrays(n, 4, syn.MODEL_SEED + n)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    for nk in batches:
        k = syn.random_kpoints(nk)
        row = "N=%4d nk=%3d " % (n, nk)
        results = {}
        for name, code in (("own", _lib.TBK_EIG_AUTO), ("rocsolver", _lib.TBK_EIG_ROCSOLVER)):
            if name == "rocsolver" and (nk > 64 or own_only):
                continue
            model.set_option(_lib.TBK_OPT_EIGENSOLVER, code)
            model.eigenval_array(k)
            model.set_option(_lib.TBK_OPT_TIMING, 1)
            model.timing()
            t0 = time.perf_counter()
            results[name] = model.eigenval_array(k)
            dt = time.perf_counter() - t0
            stages = {key: round(ms, 1) for key, (ms, _) in model.timing().items()}
            model.set_option(_lib.TBK_OPT_TIMING, 0)
            row += " %s %9.2f ms (%7.3f ms per k) %s" % (name, dt * 1e3, dt * 1e3 / nk, stages if name == "own" else "")
        if len(results) == 2:
            row += "  max|dE| %.1e" % np.abs(results["own"] - results["rocsolver"]).max()
        print(row, flush=True)
