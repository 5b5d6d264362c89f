#!/usr/bin/env python3
"""One-k-point calls (the Z2Pack call shape, _tb_model.py:1103-1108) at a BASELINE config's model shape, for a kernel trace:

   rocprofv3 --kernel-trace --stats --output-format csv -d out -o t -- python3 tools/trace_single_k_cfg.py cfg2

32 one-k `hamilton` calls, then 32 one-k `eigenval` calls through the host-buffer entry points (after a warm-up of 4 each);
prints the wall-clock per call and the HIP-event time of the H(k) stage of one call (gemv + finish)."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tbmodels_amd import _lib  # noqa: E402
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = _lib.lib()
arrays = bench.build_model_arrays(cfg)
n_orb = arrays["n_orb"]
model = bench.stage(lib, 0, arrays)
k = bench.config_kpoints(cfg, 64, arrays["R"].shape[1])
one_h = np.empty((1, n_orb, n_orb), dtype=np.complex128)
one_e = np.empty((1, n_orb))
ms = (ctypes.c_double * _lib.TBK_T_COUNT)()
launches = (ctypes.c_int64 * _lib.TBK_T_COUNT)()
for name, call in (("hamilton", lambda q: lib.tbk_hamilton(model, _lib.ptr(k[q:q + 1]), 1, 2, None, _lib.ptr(one_h))),
                   ("eigenval", lambda q: lib.tbk_eigenval(model, _lib.ptr(k[q:q + 1]), 1, _lib.ptr(one_e)))):
    for q in range(4):
        _lib.check(call(q))
    t0 = time.perf_counter()
    for q in range(calls):
        _lib.check(call(4 + q % 60))
    wall = (time.perf_counter() - t0) / calls * 1e6
    _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 1))
    _lib.check(lib.tbk_get_timing(model, None, None, 1))
    for q in range(8):
        _lib.check(call(q))
    _lib.check(lib.tbk_get_timing(model, ms, launches, 1))
    _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 0))
    stages = {s: round(ms[i] / 8 * 1e3, 1) for i, s in enumerate(_lib.STAGE_NAMES)}
    print("%s one-k %s: %.1f us per call; stage us per call (HIP events) %s" % (cfg, name, wall, stages), flush=True)
lib.tbk_model_destroy(model)
