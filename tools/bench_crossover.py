#!/usr/bin/env python3
"""QL vs bisection for the tridiagonal stage at n_orb <= 64 (the `small_call` rule of csrc/tbk_api.hip): run once with
TBK_SMALL_CALL_PER_ORBITAL=0 (QL from 4097 k-points on) and once with 100000 (bisection throughout)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

n, n_r = int(sys.argv[1]), int(sys.argv[2])
r_vec, hop, pos = syn.dense_model_arrays(n, n_r, syn.MODEL_SEED + 2)
model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
model.pin_staging()
for nk in [int(x) for x in sys.argv[3:]] or (6144, 8192, 12288, 16384, 24576, 32768, 40960, 49152):
    k = syn.random_kpoints(nk)
    model.eigenval_array(k)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        model.eigenval_array(k)
    dt = (time.perf_counter() - t0) / reps
    print("n=%d N_R=%d nk=%6d  %8.2f ms  (per-orbital rule %s)" % (n, n_r, nk, dt * 1e3, os.environ.get("TBK_SMALL_CALL_PER_ORBITAL", "640")), flush=True)
