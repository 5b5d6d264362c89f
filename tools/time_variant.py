#!/usr/bin/env python3
"""Time the reduction stage alone (tbk_reduce_standalone) with a VARIANT build of the library -- the timing-only ablation
builds of tools/build_variants.sh, whose results are wrong by construction:

    python3 tools/time_variant.py tools/exp/libtbk_<name>.so  n:matrices [n:matrices ...]

prints us per matrix (whole reduction, stage 1, stage 2) -- mean of 5 after a warm-up, three times."""
import ctypes
import sys

lib = ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)
lib.tbk_reduce_standalone.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
lib.tbk_last_error.restype = ctypes.c_char_p
for spec in sys.argv[2:]:
    n, nk = (int(x) for x in spec.split(":"))
    us = (ctypes.c_double * 3)()
    out = []
    for _ in range(3):
        rc = lib.tbk_reduce_standalone(0, n, nk, 5, us)
        if rc != 0:
            raise SystemExit("tbk_reduce_standalone: %s" % lib.tbk_last_error().decode())
        out.append("%.4f (%.4f + %.4f)" % (us[0], us[1], us[2]))
    print("%s n=%d x %d: us per matrix %s" % (sys.argv[1].split("/")[-1], n, nk, "  ".join(out)), flush=True)
