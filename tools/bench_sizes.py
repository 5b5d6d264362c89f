#!/usr/bin/env python3
"""eigenval throughput of dense synthetic models over the orbital count (N_R = 64), device-resident timing."""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [16, 32, 48, 64, 80, 96, 128, 192, 256, 384, 512]
for n in sizes:
    r_vec, hop, pos = syn.dense_model_arrays(n, 64, syn.MODEL_SEED + n)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    nk = max(2048, min(65536, int(4e9 / (16 * n * n))) // 4096 * 4096)
    k = syn.random_kpoints(nk)
    model.eigenval(k[:512])
    model.set_option(tbmodels_amd._lib.TBK_OPT_TIMING, 1)
    model.eigenval(k)
    model.timing()
    t0 = time.perf_counter()
    model.eigenval(k)
    dt = time.perf_counter() - t0
    stages = {name: round(ms, 2) for name, (ms, _) in model.timing().items()}
    print("N=%4d nk=%6d  %8.2f ms  %10.0f k-points/s  %7.3f us/k   stages(ms) %s" % (n, nk, dt * 1e3, nk / dt, dt / nk * 1e6, stages))
