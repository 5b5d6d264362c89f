import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, tbmodels_amd
from tbmodels_amd import synthetic as syn, _lib
for n, n_r in ((8, 95), (14, 140), (20, 300), (32, 500)):
    r_vec, hop, pos = syn.dense_model_arrays(n, n_r, syn.MODEL_SEED + n)
    m = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    k = syn.random_kpoints(500_000)
    m.eigenval_array(k[:1000]); m.set_option(_lib.TBK_OPT_TIMING, 1); m.eigenval_array(k); m.timing()
    t0 = time.perf_counter(); m.eigenval_array(k); dt = time.perf_counter() - t0
    print("N=%2d N_R=%3d  500k k-points: %7.2f ms  stages %s" % (n, n_r, dt * 1e3, {a: round(b[0], 2) for a, b in m.timing().items()}))
