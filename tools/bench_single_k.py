#!/usr/bin/env python3
"""One-k-point calls above 64 orbitals (the Z2Pack-style call shape, _tb_model.py:1103-1108): wall-clock of
Model.eigenval(k) / Model.hamilton(k) and the GPU stage times of one call, dense synthetic models with N_R = 16
(the eigensolver is the point here; bench.py's configs.*.single_k_us has the BASELINE shapes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402
from tbmodels_amd import _lib  # noqa: E402

rocsolver = "--rocsolver" in sys.argv  # the library path (rocsolver_zheevd) instead of the own kernels, for comparison
sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [96, 128, 200, 256, 384, 512, 768, 1024]
for n in sizes:
    r_vec, hop, pos = syn.dense_model_arrays(n, 16, syn.MODEL_SEED + n)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    k = syn.random_kpoints(64)
    if rocsolver:
        model.set_option(_lib.TBK_OPT_EIGENSOLVER, _lib.TBK_EIG_ROCSOLVER)
    for nk in (1, 7, 64):
        arg = k[0] if nk == 1 else k[:nk]
        reps = 20 if n <= 512 else 6
        model.eigenval(arg)
        model.set_option(_lib.TBK_OPT_TIMING, 1)
        model.eigenval(arg)
        model.timing()
        model.eigenval(arg)
        stages = {name: round(ms * 1e3, 1) for name, (ms, _) in model.timing().items()}
        model.set_option(_lib.TBK_OPT_TIMING, 0)
        t0 = time.perf_counter()
        for _ in range(reps):
            model.eigenval(arg)
        t_e = (time.perf_counter() - t0) / reps * 1e6
        print("N=%4d nk=%3d  eigenval %9.1f us per call   stages(us) %s" % (n, nk, t_e, stages), flush=True)
