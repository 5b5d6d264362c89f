#!/usr/bin/env python3
"""One-k hamilton calls at the headline shape (N = 64, N_R = 4096) for a kernel trace:
   rocprofv3 --kernel-trace --stats --output-format csv -d out -o t -- python3 tools/trace_onek_hamilton.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

r_vec, hop, pos = syn.dense_model_arrays(64, 4096, syn.MODEL_SEED + 2)
model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
k = syn.random_kpoints(64)
for i in range(8):
    model.hamilton(k[i])
t0 = time.perf_counter()
for i in range(48):
    model.hamilton(k[8 + i], convention=1 if i % 2 else 2)
print("one-k hamilton %.1f us per call" % ((time.perf_counter() - t0) / 48 * 1e6))
