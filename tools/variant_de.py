#!/usr/bin/env python3
"""(d, e) of seeded random Hermitian matrices through tbk_tridiagonal_reduce of a VARIANT build of the library, saved for
comparison with another build's (bit identity of a re-scheduled kernel):  python3 tools/variant_de.py lib.so n nk out.npy"""
import ctypes
import sys

import numpy as np

lib = ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)
n, nk, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
vp = ctypes.c_void_p
lib.tbk_tridiagonal_reduce.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, vp, ctypes.c_int, vp, vp, vp]
lib.tbk_last_error.restype = ctypes.c_char_p
rng = np.random.default_rng(1000 + n)
h = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
h = (h + h.conj().transpose(0, 2, 1)) / 2
if nk > 2:
    h[1] *= 1e-20
    h[2, : n // 2, n // 2:] = 0.0
    h[2, n // 2:, : n // 2] = 0.0
d, e = np.empty((nk, n)), np.empty((nk, n))
rc = lib.tbk_tridiagonal_reduce(0, n, nk, h.ctypes.data, 0, d.ctypes.data, e.ctypes.data, None)
if rc != 0:
    raise SystemExit(lib.tbk_last_error().decode())
np.save(out, np.stack([d, e]))
import scipy.linalg as la
err = max(np.abs(la.eigvalsh_tridiagonal(d[i], np.abs(e[i, : n - 1])) - np.linalg.eigvalsh(h[i])).max() / max(1e-300, np.abs(h[i]).max()) for i in range(min(nk, 4)))
print("%s n=%d nk=%d: scaled eigenvalue error of the first matrices %.2e" % (sys.argv[1].split("/")[-1], n, nk, err))
