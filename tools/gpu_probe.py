#!/usr/bin/env python3
"""Small on-GPU probes used while tuning: sustained f64 MFMA rate, device properties."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbmodels_amd import _lib  # noqa: E402

lib = _lib.lib()
print("devices:", _lib.device_count())
tf = ctypes.c_double(0)
_lib.check(lib.tbk_mfma_f64_peak(0, ctypes.byref(tf)))
print("sustained v_mfma_f64_16x16x4_f64: %.2f TFLOP/s" % tf.value)
free_b, total_b = ctypes.c_int64(0), ctypes.c_int64(0)
_lib.check(lib.tbk_device_mem_info(0, ctypes.byref(free_b), ctypes.byref(total_b)))
print("HBM free/total GiB: %.1f / %.1f" % (free_b.value / 2**30, total_b.value / 2**30))
