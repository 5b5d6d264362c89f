import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tbmodels_amd
from tbmodels_amd import synthetic as syn
r_vec, hop, pos = syn.dense_model_arrays(64, 4096, 1)
m = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
k = syn.random_kpoints(64)
for i in range(30):
    t0 = time.perf_counter(); e = m.eigenval(k[i]); t1 = time.perf_counter()
print("last eigenval(1 k): %.1f us" % ((t1 - t0) * 1e6))
