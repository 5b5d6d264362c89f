#!/usr/bin/env python3
"""One-k eigenval calls at a large orbital count, for a kernel trace of the launch chain:
   rocprofv3 --kernel-trace --stats --output-format csv -d out -o t -- python3 tools/trace_single_big.py 512"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
r_vec, hop, pos = syn.dense_model_arrays(n, 16, syn.MODEL_SEED + n)
model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
k = syn.random_kpoints(reps)
for i in range(reps):
    model.eigenval(k[i])
